#!/bin/bash
# A/B of library builds on ONE box (box-to-box spread is larger than most kernel changes): for every rowbowt_amd/librbg_<v>.so
# named on the command line, put it in place of librbg.so and run the run-indexed bench at both position widths.
# usage: tools/lib_ab.sh cur flat cond ...   (writes gpurun_out/ab_<v>_{4,8}.json)
set -u
mkdir -p gpurun_out
# RBG_AB_ARGS: other bench.py arguments (default: the run-indexed layout), e.g. "--no-space-speed --no-markers --no-cpu-baseline" for the slot tables
A=${RBG_AB_ARGS:-"--layout runs --no-space-speed --no-markers --no-cpu-baseline --steps 3 --warmup 1"}
# RBG_AB_WIDTHS: position widths to run ("4 8" by default; "4" halves the time of an A/B that needs many samples)
W=${RBG_AB_WIDTHS:-"4 8"}
for v in "$@"; do
  cp rowbowt_amd/librbg_$v.so rowbowt_amd/librbg.so || exit 1
  for w in $W; do
    extra=""; [ "$w" = 8 ] && extra="--pos-bytes 8"
    timeout -k 10 200 python bench.py $A $extra > gpurun_out/ab_${v}_$w.json 2> gpurun_out/ab_${v}_$w.err || { echo "$v $w: failed"; exit 1; }
  done
  python - "$v" $W <<'PY'
import json, sys
v = sys.argv[1]
for w in sys.argv[2:]:
    d = json.loads(open(f"gpurun_out/ab_{v}_{w}.json").read())
    print(v, w, {k.split("(")[0]: round(x["ms"], 2) for k, x in d["kernels"].items() if "plan" not in k and "order" not in k}, "value %.3e" % d["value"], flush=True)
PY
done
