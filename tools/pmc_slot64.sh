#!/bin/bash
# rocprofv3 --pmc passes (separate runs per counter group, never combined with tracing) over K1/K2 with 16-byte and with
# 64-byte rank slots on the bench index: what the 64-byte experiment costs in instructions and what it saves in misses.
#   usage (through gpurun): bash tools/pmc_slot64.sh <tag>   ->  gpurun_out/<tag>/slot64_pmc.txt
set -u
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag/slot64_pmc
mkdir -p $out
for sb in 16 64; do
  i=0
  for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
             "TCC_HIT_sum TCC_MISS_sum" "SQ_BUSY_CYCLES SQ_WAVES SQ_INST_CYCLES_VMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $grp --kernel-include-regex "k_find_range" --output-format csv -d $out/s$sb/p$i -- python3 bench.py --slot-bytes $sb --steps 2 --warmup 0 --no-cpu-baseline --check-reads 0 --no-space-speed --no-markers --property-reads 0 > $out/s$sb.p$i.json 2> $out/s$sb.p$i.err || echo "slot bytes $sb pass $i ($grp) failed"
  done
done
{
  echo "# rocprofv3 --pmc (separate passes, tools/pmc_slot64.sh) of bench.py --slot-bytes {16, 64} on the bench index, 10 M x 100 bp per launch; per-dispatch averages (SQ_*_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* in quad-cycles summed over waves or SIMDs)"
  for sb in 16 64; do echo "### slot bytes $sb"; python3 tools/summarize_pmc.py $out/s$sb; done
} > gpurun_out/$tag/slot64_pmc.txt 2>&1
rm -rf $out/*/*/*/*.db 2>/dev/null
cat gpurun_out/$tag/slot64_pmc.txt | head -80
