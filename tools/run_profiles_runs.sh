#!/bin/bash
# Third part of a measurement set: the run-indexed layout on the bench index.
#   bench.py --layout runs (both position widths), rocprofv3 kernel stats of the same command, the format A/B, PMC passes.
#   -> gpurun_out/<tag>/{bench_runs_pos4.json,bench_runs_pos8.json,kernel_stats_run_indexed.md,fmt_ab.txt,run_indexed_pmc.txt}
set -u
tag=${1:-r04}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# PMC passes first (8- and 4-byte positions): their misses / bytes go into profiles/pmc_traffic.json beside the slot kernels' for the bench lines below
bash tools/pmc_run_indexed.sh $tag -1:-1:256:5:8:48 > $out/pmc_run_indexed.log 2>&1
python3 tools/make_pmc_traffic.py --merge-runs profiles/pmc_traffic.json $out/run_indexed_pmc
mv $out/run_indexed_pmc.txt $out/run_indexed_pmc_pos8.txt
rm -rf $out/run_indexed_pmc
bash tools/pmc_run_indexed.sh $tag -1:-1:256:5:4:48 >> $out/pmc_run_indexed.log 2>&1
python3 tools/make_pmc_traffic.py --merge-runs profiles/pmc_traffic.json $out/run_indexed_pmc
mv $out/run_indexed_pmc.txt $out/run_indexed_pmc_pos4.txt
cp profiles/pmc_traffic.json $out/pmc_traffic_with_runs.json
for pb in 4 8; do
  timeout -k 10 600 python3 bench.py --layout runs --pos-bytes $pb --no-cpu-baseline --no-space-speed > $out/bench_runs_pos$pb.json 2> $out/bench_runs_pos$pb.err || { echo "bench --layout runs --pos-bytes $pb failed"; tail -5 $out/bench_runs_pos$pb.err; exit 1; }
done
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_runs -- python3 bench.py --layout runs --pos-bytes 8 --steps 3 --warmup 1 --no-cpu-baseline --check-reads 0 --property-reads 0 --no-space-speed > $out/stats_runs_bench.json 2> $out/stats_runs.err || { echo "rocprofv3 stats failed"; exit 1; }
python3 tools/summarize_rocprof.py $out/stats_runs/*/*_kernel_stats.csv $out/stats_runs/*/*_kernel_trace.csv > $out/kernel_stats_run_indexed.md 2>&1
rm -rf $out/stats_runs
bash tools/fmt_ab.sh > $out/fmt_ab.txt 2>&1 || { echo "fmt_ab failed"; tail -5 $out/fmt_ab.txt; exit 1; }
head -14 $out/kernel_stats_run_indexed.md; cat $out/fmt_ab.txt; grep -c . $out/run_indexed_pmc_pos8.txt $out/run_indexed_pmc_pos4.txt
