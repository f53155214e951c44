#!/bin/bash
# Load times from the native cache file, every load in a FRESH process (tools/load_time.py), with what the platform
# charges for large allocations beside them (tools/alloc_probe.py).   usage (through gpurun): bash tools/run_load_time.sh <tag>
#   -> gpurun_out/<tag>/load_time.txt
set -u
tag=${1:-r03}
out=gpurun_out/$tag
mkdir -p $out
{
  echo "# tools/run_load_time.sh: rbg_load_cache in a fresh process, RBG_VERBOSE=1 stage lines of the library + the tool's JSON line"
  echo "## hipMalloc on this box (tools/alloc_probe.py serial)"
  timeout 120 python3 tools/alloc_probe.py serial 2>&1 | grep -v amdgpu.ids
  echo "## bench index (synthetic chr22-scale pangenome, n = 2.0e9, r = 3.7e7): cache written by rbg_convert_runs"
  timeout 300 python3 tools/load_time.py --make /dev/shm/bench.rbgpu 2>&1 | grep -v amdgpu.ids
  sleep 20   # (the memory the synthesis freed is cleared in the background)
  for i in 1 2; do
    echo "### load $i, no option set (RBG_LAYOUT_AUTO, budget = a quarter of the free HBM: the run-indexed layout)"
    RBG_VERBOSE=1 timeout 300 python3 tools/load_time.py --load /dev/shm/bench.rbgpu 2>&1 | grep -v amdgpu.ids
    sleep 10
  done
  echo "### load, slot tables, 5 symbols per gather (RBG_HBM_BUDGET_MB=230000: the 221 GB replica of the bench headline)"
  RBG_HBM_BUDGET_MB=230000 RBG_VERBOSE=1 timeout 300 python3 tools/load_time.py --load /dev/shm/bench.rbgpu --layout slots 2>&1 | grep -v amdgpu.ids
  sleep 20
  echo "### load, run-indexed layout, minimal form (RBG_RUN_REC=1 RBG_RUN_PHI=1)"
  RBG_RUN_REC=1 RBG_RUN_PHI=1 RBG_VERBOSE=1 timeout 300 python3 tools/load_time.py --load /dev/shm/bench.rbgpu --layout runs 2>&1 | grep -v amdgpu.ids
  rm -f /dev/shm/bench.rbgpu
  echo "## n = 5.0e10 (true BWT of a 200-haplotype pangenome, tools/pangenome_bwt.py; r = 3.1e8)"
  timeout 600 python3 tools/load_time.py --make /dev/shm/pg.rbgpu --pangenome --L 250000000 --H 200 2>&1 | grep -v amdgpu.ids
  sleep 30
  echo "### load, no option set (RBG_LAYOUT_AUTO under the default budget: the run-indexed layout in its lean form)"
  RBG_VERBOSE=1 timeout 600 python3 tools/load_time.py --load /dev/shm/pg.rbgpu 2>&1 | grep -v amdgpu.ids
  sleep 20
  echo "### load, run-indexed layout under a 250 GB budget (bucket records, phi slots)"
  RBG_HBM_BUDGET_MB=250000 RBG_VERBOSE=1 timeout 600 python3 tools/load_time.py --load /dev/shm/pg.rbgpu --layout runs 2>&1 | grep -v amdgpu.ids
  sleep 20
  echo "### load, slot tables under a 250 GB budget (4 symbols per gather)"
  RBG_HBM_BUDGET_MB=250000 RBG_VERBOSE=1 timeout 600 python3 tools/load_time.py --load /dev/shm/pg.rbgpu --layout slots 2>&1 | grep -v amdgpu.ids
  rm -f /dev/shm/pg.rbgpu
} > $out/load_time.txt 2>&1
grep -c . $out/load_time.txt; grep "load_s\|hipMalloc(200" $out/load_time.txt
