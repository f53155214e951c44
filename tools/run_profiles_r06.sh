#!/bin/bash
# Round 6's measurement set, part 1 (one gpurun call): the gather ceiling, PMC passes over the default bench.py (K1/K2/K3), over bench.py --markers (the
# seeding kernels) and over the pangenome preset (K2/K3 at r = 1.2e8) -> gpurun_out/<tag>/{gather_ceiling.json, pmc.txt, pmc_markers.txt, pmc_pangenome.txt,
# pmc_traffic.json}.  Copy pmc_traffic.json + gather_ceiling.json to profiles/ before part 2 (tools/run_profiles_r06b.sh): the bench line reads them.
set -u
tag=${1:-r06}
out=gpurun_out/$tag
mkdir -p $out
[ -x tools/gather_ceiling ] || /opt/rocm/bin/hipcc -O3 -Wno-unused-value --offload-arch=gfx950 tools/gather_ceiling.hip -o tools/gather_ceiling
timeout 600 tools/gather_ceiling 16 256 $out/gather_ceiling.json > $out/gather_ceiling.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/pmc_passes.sh $tag --property-reads 0 --no-space-speed --no-markers --no-pangenome-shape > $out/pmc.log 2>&1
python3 tools/summarize_pmc.py gpurun_out/pmc_$tag > $out/pmc.txt 2>&1
python3 tools/make_pmc_traffic.py gpurun_out/pmc_$tag "profiles/${tag}_pmc.txt (rocprofv3 --pmc, separate passes per counter group, tools/pmc_passes.sh; default bench.py workload, one launch = 10M x 100 bp reads)" > $out/pmc_traffic.json 2> $out/pmc_traffic.err
# the seeding kernels (bench.py --markers): memory-side counter groups only
mkdir -p gpurun_out/pmc_${tag}mk
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE TCC_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-include-regex "seed|k_markers|k_find_range_markers" --output-format csv -d gpurun_out/pmc_${tag}mk/p$i -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --check-reads 0 --property-reads 0 --no-space-speed --no-pangenome-shape --markers > gpurun_out/pmc_${tag}mk/p$i.json 2> gpurun_out/pmc_${tag}mk/p$i.err || echo "markers pass $i ($grp) failed"
done
python3 tools/summarize_pmc.py gpurun_out/pmc_${tag}mk > $out/pmc_markers.txt 2>&1
python3 tools/make_pmc_traffic.py --merge-seeds $out/pmc_traffic.json gpurun_out/pmc_${tag}mk
# the pangenome preset (BASELINE.json configs[3]'s index shape on one GPU): K2 / K3 at 8-byte positions
mkdir -p gpurun_out/pmc_${tag}pg
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE TCC_REQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $grp --kernel-include-regex "k_find_range_runs|k_locate_fill" --output-format csv -d gpurun_out/pmc_${tag}pg/p$i -- python3 tools/pangenome_stream.py --preset driver --check-reads 0 --property-reads 0 --total-reads 20000000 > gpurun_out/pmc_${tag}pg/p$i.json 2> gpurun_out/pmc_${tag}pg/p$i.err || echo "pangenome pass $i ($grp) failed"
done
python3 tools/summarize_pmc.py gpurun_out/pmc_${tag}pg > $out/pmc_pangenome.txt 2>&1
python3 tools/make_pmc_traffic.py --merge-pangenome $out/pmc_traffic.json gpurun_out/pmc_${tag}pg "pangenome_shape L=100000000 H=200 m=150"
rm -rf gpurun_out/pmc_$tag/*/*/*.db gpurun_out/pmc_${tag}mk/*/*/*.db gpurun_out/pmc_${tag}pg/*/*/*.db 2>/dev/null
tail -3 $out/gather_ceiling.txt; grep -c . $out/pmc.txt $out/pmc_markers.txt $out/pmc_pangenome.txt; head -c 400 $out/pmc_traffic.json
