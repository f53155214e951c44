#!/bin/bash
# Entries per bucket record (RBG_RUN_REC_PER) swept on the bench index (bench.py --layout runs) and on the n = 5.0e10 stream.
# usage (on the GPU box): bash tools/rec_per_sweep.sh > gpurun_out/rec_per_sweep.txt
set -o pipefail
cd "$(dirname "$0")/.."
for per in 2.5 4 6; do
  for pb in 4 8; do
    echo "## bench.py --layout runs --pos-bytes $pb RBG_RUN_REC_PER=$per"
    RBG_RUN_REC_PER=$per timeout -k 10 300 python3 bench.py --layout runs --pos-bytes $pb --steps 10 --warmup 2 --no-cpu-baseline --no-space-speed --check-reads 5000 --property-reads 100000 2> gpurun_out/rps.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); m = d['markers']
print(json.dumps({'value': round(d['value'] / 1e9, 4), 'hbm_GB': round(d['config']['index']['hbm_bytes'] / 1e9, 2), 'ms': {k.split('(')[0]: round(v['ms'], 3) for k, v in d['kernels'].items()},
                  'greedy_ms': round(m['greedy_seed']['ms_per_step'], 2), 'seeds_two_walk_ms': round(m['marker_seeds']['two_walks_ms_per_step'], 2), 'seeds_logged_ms': round(m['marker_seeds']['logged']['ms_per_step'], 2)}))" || { echo FAILED; tail -5 gpurun_out/rps.err; exit 1; }
  done
done
for per in 2.5 5; do
  echo "## pangenome_stream.py n = 5.0e10 --layout runs RBG_RUN_REC_PER=$per"
  RBG_RUN_REC_PER=$per timeout -k 10 400 python3 tools/pangenome_stream.py --L 250000000 --H 200 --layout runs --reads 10000000 --total-reads 100000000 --implicit-text on --out-json gpurun_out/rps_s.json > gpurun_out/rps_s.log 2>&1 || { echo FAILED; tail -5 gpurun_out/rps_s.log; exit 1; }
  python3 -c "
import json
d = json.load(open('gpurun_out/rps_s.json')); li = d['config']['index']['layout_info']
print(json.dumps({'value': d['value'], 'hbm_GB': round(d['config']['index']['hbm_bytes'] / 1e9, 1), 'ms': {k: round(v, 2) for k, v in d['kernel_ms_one_batch'].items()}, 'bit_exact': d['parity']['bit_exact_vs_oracle'],
                  'rec_GB': [round(x / 1e9, 1) for x in li['rec_bytes']], 'overflow_share': [round(o / max(1, b // 64), 4) for o, b in zip(li['rec_overflow'], li['rec_bytes'])]}))"
done
