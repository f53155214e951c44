#!/bin/bash
# The run-indexed search (find_range_w_toehold, 10 M x 150 bp per launch) over pangenomes of growing length and haplotype count: is it the SIZE of the
# index or its SHAPE (the clusters of short runs at variant sites grow with the haplotypes) that takes the kernel off the sector ceiling?
# usage (on the GPU box): bash tools/scale_probe.sh > gpurun_out/scale_probe.txt
cd "$(dirname "$0")/.."
for cfg in "40000000 50" "100000000 100" "100000000 200" "200000000 100" "250000000 200"; do
  set -- $cfg
  echo "## L = $1, H = $2"
  timeout -k 10 300 python3 tools/pangenome_stream.py --L $1 --H $2 --layout runs --reads 10000000 --total-reads 20000000 --check-reads 2000 --property-reads 20000 --implicit-text on --run-rec 2 --run-phi 2 2>&1 | grep "pangenome: L=\|index replica\|one batch" | cut -c1-330 || exit 1
done
