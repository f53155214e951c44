#!/bin/bash
# Fourth part of a measurement set: the streamed workload (BASELINE.json configs[3] shape) on the n = 5.0e10 pangenome,
# slot tables and the run-indexed layout (phi over the list of sampled positions; phi slots).
#   -> gpurun_out/<tag>/pangenome_stream_n5e10{,_run_indexed_minimal,_run_indexed}.{json,log}   (minimal: directories + run lists, phi over the list)
set -u
tag=${1:-r04}
out=gpurun_out/$tag
mkdir -p $out
run() {  # name, args...
  local name=$1; shift
  RBG_VERBOSE=1 timeout -k 10 560 python3 tools/pangenome_stream.py --L 250000000 --H 200 --reads 10000000 "$@" --out-json $out/$name.json > $out/$name.log 2>&1 || { echo "$name failed"; tail -5 $out/$name.log; exit 1; }
  python3 -c "
import json, sys
d = json.load(open('$out/$name.json')); ix = d['config']['index']
print('$name', '%.3e reads/s' % d['value'], {k: round(v, 2) for k, v in d['kernel_ms_one_batch'].items()}, '%.1f GB replica' % (ix['hbm_bytes'] / 1e9), 'depths', ix.get('kmer_depths_with_tables'), 'bit-exact', d['parity']['bit_exact_vs_oracle'])"
}
run pangenome_stream_n5e10_run_indexed_minimal --layout runs --run-phi 1 --run-rec 1 --total-reads 200000000 --implicit-text on
run pangenome_stream_n5e10_run_indexed --layout runs --total-reads 200000000 --implicit-text on
run pangenome_stream_n5e10 --layout slots --total-reads 1000000000
