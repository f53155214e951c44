#!/bin/bash
# Round 6's measurement set, part 2 (after profiles/pmc_traffic.json + gather_ceiling.json of part 1 are in place): the bench.py line, rocprofv3 kernel
# stats of the same command, host-pointer API and CLI rates, rb_markers under both layouts.
#   -> gpurun_out/<tag>/{bench.json, kernel_stats.md, host_api.txt, cli_rate.txt, rb_markers_layout.txt}
set -u
tag=${1:-r06}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python3 bench.py --steps 20 --warmup 3 > $out/bench.json 2> $out/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --check-reads 0 --property-reads 0 --no-space-speed --no-pangenome-shape > $out/stats_bench.json 2> $out/stats.err
python3 tools/summarize_rocprof.py $out/stats/*/*_kernel_stats.csv $out/stats/*/*_kernel_trace.csv > $out/kernel_stats.md 2>&1
rm -rf $out/stats
timeout 400 python3 tools/host_api_rate.py 2>&1 | grep -v amdgpu.ids > $out/host_api.txt
timeout 400 python3 tools/cli_rate.py 2>&1 | grep -v amdgpu.ids > $out/cli_rate.txt
timeout 900 python3 tools/cli_rate_bench.py --only-markers 2>&1 | grep -v amdgpu.ids > $out/rb_markers_layout.txt
tail -3 $out/bench.err; head -c 800 $out/bench.json; echo; head -14 $out/kernel_stats.md; cat $out/host_api.txt; cat $out/cli_rate.txt; tail -14 $out/rb_markers_layout.txt
