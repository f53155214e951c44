#!/bin/bash
# bucket records of the run-indexed layout (RBG_OPT_RUN_REC): entries per bucket (RBG_RUN_REC_PER) against K1/K2 time and replica size, bench index.
# usage (GPU box): bash tools/rec_ab.sh > gpurun_out/r04_rec_ab.txt
set -o pipefail
cd "$(dirname "$0")/.."
for pb in 4 8; do
  for per in off 1.5 2.5 4; do
    if [ "$per" = off ]; then export RBG_RUN_REC=1; unset RBG_RUN_REC_PER; else export RBG_RUN_REC=2; export RBG_RUN_REC_PER=$per; fi
    echo "## --layout runs --pos-bytes $pb records: $per entries per bucket"
    timeout -k 10 400 python bench.py --layout runs --pos-bytes $pb --steps 10 --warmup 2 --no-cpu-baseline --no-markers --no-space-speed --check-reads 5000 --property-reads 100000 2> gpurun_out/rec_ab.err | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
k = d.get('kernels') or {}
print(json.dumps({'value': round(d['value']), 'ms_per_step': round(d['ms_per_step'], 3), 'hbm_GB': round(d['config']['index']['hbm_bytes'] / 1e9, 2), 'count_only_ms': round(d['count_only']['ms_per_step'], 3),
                  'kernels_ms': {n: round(v['ms'], 3) for n, v in k.items()}, 'touched': (k.get('k_find_range<toehold>') or {}).get('touched')}))
" || { echo FAILED; tail -5 gpurun_out/rec_ab.err; exit 1; }
  done
done
