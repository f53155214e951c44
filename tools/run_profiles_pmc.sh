#!/bin/bash
# First half of a measurement set (run_profiles.sh in two gpurun calls): gather ceiling + PMC passes.
#   -> gpurun_out/<tag>/{gather_ceiling.json,gather_ceiling.txt,pmc.txt,pmc_traffic.json}; copy the two JSON files to profiles/ before the second half
set -u
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
[ -x tools/gather_ceiling ] || /opt/rocm/bin/hipcc -O3 -Wno-unused-value --offload-arch=gfx950 tools/gather_ceiling.hip -o tools/gather_ceiling
timeout 600 tools/gather_ceiling 16 256 $out/gather_ceiling.json > $out/gather_ceiling.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# the default bench.py (the headline replica is what a default rbg_load builds: the run-indexed layout since round 4) ...
bash tools/pmc_passes.sh $tag --property-reads 0 --no-space-speed --no-markers > $out/pmc.log 2>&1
python3 tools/summarize_pmc.py gpurun_out/pmc_$tag > $out/pmc.txt 2>&1
# ... and the slot tables at five symbols per gather (--hbm-budget-gb -1: the 221 GB replica, the headline of rounds 1-3)
bash tools/pmc_passes.sh ${tag}slots --hbm-budget-gb -1 --property-reads 0 --no-space-speed --no-markers > $out/pmc_slots.log 2>&1
python3 tools/summarize_pmc.py gpurun_out/pmc_${tag}slots > $out/pmc_slots.txt 2>&1
python3 tools/make_pmc_traffic.py gpurun_out/pmc_$tag "profiles/${tag}_pmc.txt, ${tag}_pmc_slots.txt (rocprofv3 --pmc, separate passes per counter group, tools/pmc_passes.sh; default bench.py workload, one launch = 10M x 100 bp reads)" gpurun_out/pmc_${tag}slots > $out/pmc_traffic.json 2> $out/pmc_traffic.err
rm -rf gpurun_out/pmc_$tag/*/*/*.db gpurun_out/pmc_${tag}slots/*/*/*.db 2>/dev/null
tail -3 $out/gather_ceiling.txt; grep -c . $out/pmc.txt; head -c 600 $out/pmc_traffic.json
