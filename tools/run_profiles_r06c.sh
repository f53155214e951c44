bash tools/run_profiles_r06b.sh r06 > gpurun_out/r06/part2.log 2>&1
RBG_VERBOSE=1 timeout -k 10 500 python3 tools/pangenome_stream.py --L 250000000 --H 200 --reads 10000000 --total-reads 200000000 --hbm-reserve-gb 0 --implicit-text on --check-reads 2000 --property-reads 100000 --out-json gpurun_out/r06/pangenome_stream_n5e10_default.json > gpurun_out/r06/pangenome_stream_n5e10_default.log 2>&1 || echo "n5e10 failed"
tail -30 gpurun_out/r06/part2.log
grep "one batch, per kernel" gpurun_out/r06/pangenome_stream_n5e10_default.log
