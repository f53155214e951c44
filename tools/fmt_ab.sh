#!/bin/bash
# A/B of the run-indexed layout's two formats on the bench index (bench.py --layout runs), both position widths.
# usage (on the GPU box): bash tools/fmt_ab.sh > gpurun_out/r04_fmt_ab.txt
set -o pipefail
cd "$(dirname "$0")/.."
for pb in 4 8; do
  for fmt in 1 2; do
    echo "## --layout runs --pos-bytes $pb RBG_RUN_FMT=$fmt"
    RBG_RUN_FMT=$fmt timeout -k 10 500 python bench.py --layout runs --pos-bytes $pb --steps 10 --warmup 2 --no-cpu-baseline --no-markers --no-space-speed --check-reads 5000 --property-reads 100000 2> gpurun_out/fmt_ab_${pb}_${fmt}.err | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
r = d.get('roofline', {})
print(json.dumps({'value': d['value'], 'ms_per_step': d['ms_per_step'], 'kernels_ms': r.get('kernels_ms') or d.get('kernel_ms'), 'hbm_bytes': d.get('config', {}).get('index', {}).get('hbm_bytes')}))
open('gpurun_out/fmt_ab_%s_%s.json' % ('$pb', '$fmt'), 'w').write(json.dumps(d))
print(json.dumps({k: round(v['ms'], 3) for k, v in (r.get('kernels') or {}).items()}))
" || { echo FAILED; tail -5 gpurun_out/fmt_ab_${pb}_${fmt}.err; exit 1; }
  done
done
