for k in 5 6 7 8; do
  RBG_KMER_STEPS=$k timeout -k 10 150 python bench.py --no-space-speed --no-cpu-baseline --no-markers --check-reads 2000 --property-reads 0 > gpurun_out/ks_$k.json 2> gpurun_out/ks_$k.err || echo fail $k
done
