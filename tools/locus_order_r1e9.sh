# K3's chain order at r = 1.07e9 (n = 3.0e11, H = 520): by locus (document table attached) against absolute position; default load; no oracle (properties only)
set -u
mkdir -p gpurun_out/r06locus
for ord in abs locus; do
  [ $ord = abs ] && export RBG_LOCATE_ORDER=abs || unset RBG_LOCATE_ORDER
  timeout -k 10 900 python tools/pangenome_stream.py --L 580000000 --H 520 --total-reads 30000000 --check-reads 0 --property-reads 100000 --hbm-reserve-gb 0 --out-json gpurun_out/r06locus/r1e9_$ord.json > /dev/null 2> gpurun_out/r06locus/r1e9_$ord.log || echo "FAILED r1e9 $ord"
  grep "one batch, per kernel\|index replica" gpurun_out/r06locus/r1e9_$ord.log
done
