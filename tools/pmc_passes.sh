#!/bin/bash
# Separate rocprofv3 --pmc passes (never combined with --stats/--sys-trace), one counter group per
# run, as /opt/skills/guides/MI355X_MICROARCH.md prescribes (TCC: FETCH_SIZE costs 3 slots,
# WRITE_SIZE 2 -> separate passes).  Output CSVs land in gpurun_out/pmc_<tag>/.
# usage: tools/pmc_passes.sh <tag> [bench args...]
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc_${tag}
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE TCC_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
           "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-include-regex "rbg" --output-format csv -d gpurun_out/pmc_${tag}/p$i -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --check-reads 0 "$@" > gpurun_out/pmc_${tag}/p$i.json 2> gpurun_out/pmc_${tag}/p$i.err || echo "pass $i ($grp) failed"
done
ls -R gpurun_out/pmc_${tag} | head -40
