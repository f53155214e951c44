#!/bin/bash
# Second half of a measurement set: the bench.py line (reads profiles/pmc_traffic.json + gather_ceiling.json of the first
# half), rocprofv3 kernel stats of the same command, host-pointer API and CLI rates.
#   -> gpurun_out/<tag>/{bench.json,kernel_stats.md,host_api.txt,cli_rate.txt}
set -u
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python3 bench.py > $out/bench.json 2> $out/bench.err
# the slot tables at five symbols per gather as the headline replica (221 GB: the headline of rounds 1-3), for the record
timeout 600 python3 bench.py --hbm-budget-gb -1 --no-space-speed > $out/bench_slots_221GB.json 2> $out/bench_slots_221GB.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --check-reads 0 --property-reads 0 --no-space-speed > $out/stats_bench.json 2> $out/stats.err
python3 tools/summarize_rocprof.py $out/stats/*/*_kernel_stats.csv $out/stats/*/*_kernel_trace.csv > $out/kernel_stats.md 2>&1
rm -rf $out/stats
timeout 400 python3 tools/host_api_rate.py 2>&1 | grep -v amdgpu.ids > $out/host_api.txt
timeout 400 python3 tools/cli_rate.py 2>&1 | grep -v amdgpu.ids > $out/cli_rate.txt
tail -3 $out/bench.err; head -c 1200 $out/bench.json; echo; head -12 $out/kernel_stats.md; cat $out/host_api.txt; cat $out/cli_rate.txt
