#!/bin/bash
# A/B of library builds on ONE box (box-to-box spread is larger than most kernel changes): for every rowbowt_amd/librbg_<v>.so
# named on the command line, put it in place of librbg.so and run the run-indexed bench at both position widths.
# usage: tools/lib_ab.sh cur flat cond ...   (writes gpurun_out/ab_<v>_{4,8}.json)
set -u
mkdir -p gpurun_out
# RBG_AB_ARGS: other bench.py arguments (default: the run-indexed layout), e.g. "--no-space-speed --no-markers --no-cpu-baseline" for the slot tables
A=${RBG_AB_ARGS:-"--layout runs --no-space-speed --no-markers --no-cpu-baseline --steps 3 --warmup 1"}
for v in "$@"; do
  cp rowbowt_amd/librbg_$v.so rowbowt_amd/librbg.so || exit 1
  timeout -k 10 200 python bench.py $A > gpurun_out/ab_${v}_4.json 2> gpurun_out/ab_${v}_4.err || { echo "$v 4: failed"; exit 1; }
  timeout -k 10 200 python bench.py $A --pos-bytes 8 > gpurun_out/ab_${v}_8.json 2> gpurun_out/ab_${v}_8.err || { echo "$v 8: failed"; exit 1; }
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
for w in (4, 8):
    d = json.loads(open(f"gpurun_out/ab_{v}_{w}.json").read())
    print(v, w, {k.split("(")[0]: round(x["ms"], 2) for k, x in d["kernels"].items() if "plan" not in k and "order" not in k}, flush=True)
PY
done
