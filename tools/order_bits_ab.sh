mkdir -p gpurun_out/r06ob
for b in 0 16 19; do
  RBG_LOCATE_ORDER_BITS=$b python bench.py --steps 10 --warmup 2 --no-pangenome-shape --no-cpu-baseline --no-space-speed --no-markers --check-reads 2000 --property-reads 0 > gpurun_out/r06ob/b$b.json 2> gpurun_out/r06ob/b$b.log
  python - <<P
import json
d=json.loads(open("gpurun_out/r06ob/b$b.json").read().strip().splitlines()[-1])
print("order bits $b", "%.4e" % d["value"], {k: round(v["ms"],3) for k,v in d["kernels"].items()})
P
done
python bench.py --two-stream --steps 10 --warmup 2 --no-pangenome-shape --no-cpu-baseline --no-space-speed --no-markers --check-reads 2000 --property-reads 0 > gpurun_out/r06ob/two.json 2> gpurun_out/r06ob/two.log
python - <<P
import json
d=json.loads(open("gpurun_out/r06ob/two.json").read().strip().splitlines()[-1])
print("two-stream", d["value"], d["two_stream_pipeline"])
P
