#!/bin/bash
# rocprofv3 --pmc passes (separate runs per counter group, never combined with tracing) over the search and locate kernels of the
# streamed workload on the n = 5.0e10 pangenome, run-indexed layout: two batches of 10 M x 150 bp reads, no oracle.
#   usage (through gpurun): bash tools/pmc_stream.sh <tag> <name> [extra pangenome_stream.py arguments]
#   -> gpurun_out/<tag>/pmc_stream_<name>.txt
set -u
tag=${1:-r04}
name=${2:-default}
shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag/pmc_stream_$name
mkdir -p $out
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE TCC_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout -k 10 280 rocprofv3 --pmc $grp --kernel-include-regex "k_find_range_runs|k_locate_fill" --output-format csv -d $out/p$i -- python3 tools/pangenome_stream.py --L 250000000 --H 200 --layout runs --reads 10000000 --total-reads 20000000 --check-reads 0 --property-reads 0 --implicit-text on "$@" > $out/p$i.txt 2> $out/p$i.err || { echo "pass $i ($grp) failed"; tail -3 $out/p$i.err; exit 1; }
done
{
  echo "# rocprofv3 --pmc (separate passes, tools/pmc_stream.sh) of tools/pangenome_stream.py --L 250000000 --H 200 --layout runs $* (n = 5.0e10, r = 3.1e8; 10 M x 150 bp per launch; FETCH_SIZE / WRITE_SIZE in KiB; SQ_* in quad-cycles summed over waves or SIMDs, SQ_BUSY_CYCLES over the 32 shader engines)"
  grep -h "one batch, per kernel" $out/p1.err $out/p1.txt | head -1
  python3 tools/summarize_pmc.py $out
} > gpurun_out/$tag/pmc_stream_$name.txt 2>&1
rm -rf $out/*/*/*.db 2>/dev/null
grep -c . gpurun_out/$tag/pmc_stream_$name.txt
