#!/bin/bash
# One measurement set on the GPU box: the gather ceiling, the bench.py line, rocprofv3 kernel stats of the same
# command, PMC passes + profiles-ready summaries.   usage (through gpurun): bash tools/run_profiles.sh <tag>
#   -> gpurun_out/<tag>/{gather_ceiling.json,bench.json,kernel_stats.md,pmc.txt,pmc_traffic.json}
set -u
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
[ -x tools/gather_ceiling ] || /opt/rocm/bin/hipcc -O3 -Wno-unused-value --offload-arch=gfx950 tools/gather_ceiling.hip -o tools/gather_ceiling
timeout 600 tools/gather_ceiling 16 256 $out/gather_ceiling.json > $out/gather_ceiling.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/pmc_passes.sh $tag --property-reads 0 --no-space-speed --no-markers > $out/pmc.log 2>&1
python3 tools/summarize_pmc.py gpurun_out/pmc_$tag > $out/pmc.txt 2>&1
python3 tools/make_pmc_traffic.py gpurun_out/pmc_$tag "profiles/${tag}_pmc.txt (rocprofv3 --pmc, separate passes per counter group, tools/pmc_passes.sh; default bench.py workload, one launch = 10M x 100 bp reads)" > $out/pmc_traffic.json 2> $out/pmc_traffic.err
# the bench line reads the two files from profiles/: put this run's in place first
cp $out/pmc_traffic.json profiles/pmc_traffic.json
cp $out/gather_ceiling.json profiles/gather_ceiling.json
timeout 1500 python3 bench.py > $out/bench.json 2> $out/bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --check-reads 0 --property-reads 0 --no-space-speed > $out/stats_bench.json 2> $out/stats.err
python3 tools/summarize_rocprof.py $out/stats/*/*_kernel_stats.csv $out/stats/*/*_kernel_trace.csv > $out/kernel_stats.md 2>&1
rm -rf $out/stats/*/*.db 2>/dev/null
tail -3 $out/bench.err; head -c 1500 $out/bench.json; echo; head -20 $out/kernel_stats.md; grep -c . $out/pmc.txt; tail -3 $out/gather_ceiling.txt
