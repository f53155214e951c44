set -u
out=gpurun_out/r06uni
mkdir -p $out
for rep in 1 2 3; do
for v in 0 1; do
  RBG_RUN_UNIFORM=$v python bench.py --steps 10 --warmup 2 --no-pangenome-shape --no-cpu-baseline --no-space-speed --no-markers --check-reads 2000 --property-reads 0 > $out/b_$v.json 2> $out/b_$v.log || { echo "FAILED $v"; tail -5 $out/b_$v.log; exit 1; }
  python - <<P
import json
d=json.loads(open("$out/b_$v.json").read().strip().splitlines()[-1])
li=d["config"]["index"]["layout_info"]
print("uniform=$v rep=$rep", "%.4e" % d["value"], {k.split("(")[0]: round(x["ms"],3) for k,x in d["kernels"].items()}, "parity", d.get("parity",{}).get("bit_exact_vs_oracle"), "rec_bytes", [round(x/1e9,2) for x in li["rec_bytes"]], "overflow", li["rec_overflow"], "per_read", {k: round(v,2) for k,v in d["per_read"].items()})
P
done
done
