#!/bin/bash
# rocprofv3 --pmc passes (one counter group per run, never combined with tracing) over the run-indexed search kernels of the bench workload.
#   usage (through gpurun): bash tools/pmc_k2.sh <tag> [environment assignments, e.g. RBG_KMER_STEPS=5]
#   -> gpurun_out/<tag>/pmc_k2.txt
set -u
tag=${1:-r05}; shift || true
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag/pmc_k2
mkdir -p $out
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE TCC_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-include-regex "k_find_range_runs" --output-format csv -d $out/p$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-space-speed --no-markers --check-reads 0 --property-reads 0 > $out/p$i.json 2> $out/p$i.err || echo "pass $i ($grp) failed"
done
{
  echo "# rocprofv3 --pmc (separate passes, tools/pmc_k2.sh $*) of bench.py's search kernels on the bench index, 10 M x 100 bp per launch; FETCH_SIZE / WRITE_SIZE in KiB; SQ_*_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* in quad-cycles summed over waves or SIMDs, SQ_BUSY_CYCLES over the 32 shader engines"
  python3 tools/summarize_pmc.py $out
} > gpurun_out/$tag/pmc_k2.txt 2>&1
rm -rf $out/*/*/*.db 2>/dev/null
grep -c . gpurun_out/$tag/pmc_k2.txt
