set -u
mkdir -p gpurun_out/r06k3
for cfg in "0 0" "1 0" "1 16" "0 16"; do
  set -- $cfg
  RBG_K3_ALIGN=$1 RBG_K3_CHUNK=$2 python tools/pangenome_stream.py --preset driver --check-reads 500 --property-reads 20000 --total-reads 30000000 --out-json gpurun_out/r06k3/pg_align$1_chunk$2.json > /dev/null 2> gpurun_out/r06k3/pg_align$1_chunk$2.log || echo "FAILED $cfg"
  grep "one batch, per kernel" gpurun_out/r06k3/pg_align$1_chunk$2.log
done
