# K3's chain order by locus against the order by absolute position: the pangenome preset and the bench workload, one box
set -u
mkdir -p gpurun_out/r06locus
for ord in abs locus; do
  env=""; [ $ord = abs ] && export RBG_LOCATE_ORDER=abs || unset RBG_LOCATE_ORDER
  python tools/pangenome_stream.py --preset driver --check-reads 1000 --property-reads 50000 --total-reads 30000000 --out-json gpurun_out/r06locus/pg_$ord.json > /dev/null 2> gpurun_out/r06locus/pg_$ord.log || echo "FAILED pangenome $ord"
  grep "one batch, per kernel" gpurun_out/r06locus/pg_$ord.log
  python bench.py --docs on --steps 10 --no-space-speed --no-markers --no-cpu-baseline --no-pangenome-shape --check-reads 5000 > gpurun_out/r06locus/bench_$ord.json 2> gpurun_out/r06locus/bench_$ord.log || echo "FAILED bench $ord"
  python - <<P
import json
d=json.loads(open("gpurun_out/r06locus/bench_$ord.json").read().strip().splitlines()[-1])
print("bench $ord", "%.4e" % d["value"], {k: round(v["ms"],3) for k,v in d["kernels"].items()}, d["parity"]["bit_exact_vs_oracle"])
P
done
