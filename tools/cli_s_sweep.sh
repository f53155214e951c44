#!/bin/bash
# rb_align -s on the bench index over RB_ALIGN_TEXT_STREAMS (half-shards per replica) and the host formatter, one box.
mkdir -p gpurun_out
for s in 1 2 3 4; do echo "## RB_ALIGN_TEXT_STREAMS=$s"; RB_ALIGN_TEXT_STREAMS=$s timeout -k 10 300 python tools/cli_rate_bench.py --only-s 2>/dev/null | grep "rb_align"; done
echo "## RB_ALIGN_HOST_TEXT=1"; RB_ALIGN_HOST_TEXT=1 timeout -k 10 300 python tools/cli_rate_bench.py --only-s 2>/dev/null | grep "rb_align"
