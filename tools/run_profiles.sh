#!/bin/bash
# One measurement set on the GPU box: bench.py line, rocprofv3 kernel stats of the same command, PMC passes.
# usage (through gpurun): bash tools/run_profiles.sh   -> gpurun_out/v16/
set -u
mkdir -p gpurun_out/v16
timeout 900 python3 bench.py > gpurun_out/v16/bench.json 2> gpurun_out/v16/bench.err
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/v16/stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --check-reads 0 --property-reads 0 > gpurun_out/v16/stats_bench.json 2> gpurun_out/v16/stats.err
python3 tools/summarize_rocprof.py gpurun_out/v16/stats/*/*_kernel_stats.csv gpurun_out/v16/stats/*/*_kernel_trace.csv > gpurun_out/v16/kernel_stats.md 2>&1
bash tools/pmc_passes.sh v16 --property-reads 0 > gpurun_out/v16/pmc.log 2>&1
python3 tools/summarize_pmc.py gpurun_out/pmc_v16 > gpurun_out/v16/pmc.txt 2>&1
rm -rf gpurun_out/v16/stats/*/*.db 2>/dev/null
tail -3 gpurun_out/v16/bench.err; head -c 600 gpurun_out/v16/bench.json; echo; head -20 gpurun_out/v16/kernel_stats.md; grep -c . gpurun_out/v16/pmc.txt
