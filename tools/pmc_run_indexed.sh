#!/bin/bash
# rocprofv3 --pmc passes (separate runs per counter group, never combined with tracing) over the run-indexed kernels
# on the bench index: tools/tune.py with RBG_TUNE_LAYOUT=runs, one replica, 10 M x 100 bp reads per launch.
#   usage (through gpurun): bash tools/pmc_run_indexed.sh <tag> [tune config = -1:-1:256:5:0:48]
#   -> gpurun_out/<tag>/run_indexed_pmc.txt
set -u
tag=${1:-r02}
cfg=${2:--1:-1:256:5:0:48}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag/run_indexed_pmc
mkdir -p $out
export RBG_TUNE_LAYOUT=runs
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE TCC_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $grp --kernel-include-regex "_runs" --output-format csv -d $out/p$i -- python3 tools/tune.py --configs=$cfg > $out/p$i.txt 2> $out/p$i.err || echo "pass $i ($grp) failed"
done
{
  echo "# rocprofv3 --pmc (separate passes, tools/pmc_run_indexed.sh) of tools/tune.py with RBG_TUNE_LAYOUT=runs on the bench index (config $cfg; 10 M x 100 bp per launch; FETCH_SIZE / WRITE_SIZE in KiB; SQ_*_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* in quad-cycles summed over waves or SIMDs, SQ_BUSY_CYCLES summed over the 32 shader engines, SQ_LDS_IDX_ACTIVE summed over the 256 CUs); the k_locate_fill_runs dispatches mix the unordered and the ordered walk"
  python3 tools/summarize_pmc.py $out
} > gpurun_out/$tag/run_indexed_pmc.txt 2>&1
rm -rf $out/*/*/*.db 2>/dev/null
head -40 gpurun_out/$tag/run_indexed_pmc.txt
