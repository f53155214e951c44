#!/bin/bash
# Start-up of rb_align with one and with three replicas of the index (three no slower than 1.5 x one).
# One GPU holds one 221 GB slot replica, so the three are of the run-indexed layout (RBG_LAYOUT=runs: 2.8 GB each) on the
# same device: the peer-copy path of rbg_replicate_many with every relocation, not the xGMI links.
# usage (through gpurun): bash tools/replica_start.sh   -> stdout
d=/tmp/cli_big
python3 tools/cli_rate_bench.py --only-sm 1 --dir $d > /dev/null 2>&1   # (prepares $d/idx.rbgpu and $d/reads.fq)
head -c 400000000 $d/reads.fq | head -n 1600000 > $d/reads_s.fq
for dev in 0 0,0,0 0 0,0,0; do
  RBG_LAYOUT=runs timeout -k 10 200 rowbowt_amd/rb_align -s --devices $dev $d/idx $d/reads_s.fq > /dev/null 2> $d/err.txt
  echo "RBG_LAYOUT=runs rb_align -s --devices $dev: <load s> <query s> = $(tail -1 $d/err.txt)"
done
