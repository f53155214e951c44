mkdir -p gpurun_out/r06mg
# (1) the driver's launch form with one rank: torch.distributed.run, nccl = RCCL, world 1 (full default sizes are not needed for the plumbing)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 3 --warmup 1 --L 1500000 --H 8 --reads 150000 --no-cpu-baseline --check-reads 2000 --property-reads 20000 > gpurun_out/r06mg/tdr1.json 2> gpurun_out/r06mg/tdr1.log; echo "tdr1 rc=$?"
python - <<P
import json
d=json.loads([l for l in open("gpurun_out/r06mg/tdr1.json") if l.startswith("{")][-1])
print("tdr1", d["n_gpus"], "%.3e" % d["value"], d["per_rank"], d["rccl_ranks_seen"], d["collective_backend"], d["cache_write_s"], d["counters"], d["parity"]["bit_exact_vs_oracle"])
P
# (2) four rehearsal ranks on the one GPU (gloo): launcher, one cache file, barriers, max-over-ranks, counters, per-rank block
RBG_BENCH_TRACE=1 python bench.py --gpus 4 --rehearse-ranks --steps 2 --warmup 1 --L 1500000 --H 8 --reads 150000 --no-cpu-baseline --check-reads 2000 --property-reads 20000 > gpurun_out/r06mg/reh4.json 2> gpurun_out/r06mg/reh4.log; echo "reh4 rc=$?"
python - <<P
import json
d=json.loads([l for l in open("gpurun_out/r06mg/reh4.json") if l.startswith("{")][-1])
print("reh4", d["n_gpus"], "%.3e" % d["value"], [(x["rank"], x["device"], round(x["load_s"],2), round(x["wait_for_rank0_s"],2), round(x["ms_per_step"],3)) for x in d["per_rank"]], d["rccl_ranks_seen"], d["cache_write_s"], d["counters"]["reads"], d["parity"]["bit_exact_vs_oracle"])
P
grep -c "rank" gpurun_out/r06mg/reh4.log
# (3) replicas of one process, three on device 0: peer copy stats
python bench.py --replicas 3 --replica-devices 0,0,0 --steps 2 --warmup 1 --L 1500000 --H 8 --reads 150000 --no-cpu-baseline --no-markers --check-reads 2000 --property-reads 20000 2> gpurun_out/r06mg/rep3.log | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['replicas_one_process']; print('rep3', d['n_gpus'], r['replicated'], r['peer_copies'], r['fan_out_GBps'], r['every_copy_identical_to_the_primary_on_its_batch'])"
