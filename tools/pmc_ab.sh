#!/bin/bash
# rocprofv3 --pmc passes (separate runs per counter group, never combined with tracing) over the run-indexed K1/K2 at
# 8-byte positions for library builds rowbowt_amd/librbg_<v>.so, on one box.
#   usage (through gpurun): bash tools/pmc_ab.sh <tag> <v> [<v> ...]   ->  gpurun_out/<tag>/pmc_ab.txt
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag/pmc_ab
mkdir -p $out
for v in "$@"; do
  cp rowbowt_amd/librbg_$v.so rowbowt_amd/librbg.so || exit 1
  i=0
  for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum" "SQ_BUSY_CYCLES SQ_WAVES SQ_INST_CYCLES_VMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $grp --kernel-include-regex "k_find_range_runs" --output-format csv -d $out/$v/p$i -- python3 bench.py --layout runs --pos-bytes 8 --steps 2 --warmup 0 --no-cpu-baseline --check-reads 0 --no-space-speed --no-markers --property-reads 0 > $out/$v.p$i.json 2> $out/$v.p$i.err || echo "$v pass $i ($grp) failed"
  done
done
{
  echo "# rocprofv3 --pmc (separate passes, tools/pmc_ab.sh) of bench.py --layout runs --pos-bytes 8 on the bench index, 10 M x 100 bp per launch; per-dispatch averages"
  for v in "$@"; do echo "### librbg_$v.so"; python3 tools/summarize_pmc.py $out/$v; done
} > gpurun_out/$tag/pmc_ab.txt 2>&1
rm -rf $out/*/*/*/*.db 2>/dev/null
head -90 gpurun_out/$tag/pmc_ab.txt
