#!/usr/bin/env python3
"""BASELINE.json configs[3] on one GPU (and, under torch.distributed.run, on N): a pangenome-scale r-index with
positions beyond 32 bits (with --gpus N this process starts the N ranks itself), a stream of synthetic 150 bp reads generated ON THE DEVICE batch by batch from a
counter-based RNG (rbg_sample_reads_dev: the host cannot feed 150 GB), count+locate per batch, global counters
reduced over RCCL.  Prints one JSON line (rank 0).

The index is a TRUE BWT: rowbowt_amd/tools/pangenome_bwt.py derives the run-length BWT and the run-boundary SA
samples of the H-haplotype text from its structure (no suffix array of the text; checked against prefix doubling
in tests/test_pangenome_bwt.py and, with --verify-sa, here).  Parity: ranges, toeholds (k-mer steps included) and
locations of --check-reads reads against oracle/rb_oracle.c; size-independent properties on --property-reads reads
(every location is an occurrence of the read in the text, locations distinct, as many as the range is wide).

  python tools/pangenome_stream.py                      # n = 5.0e10 (L = 2.5e8, H = 200), 1e9 reads x 150 bp
  python tools/pangenome_stream.py --L 44000000 --H 100 --total-reads 100000000   # n = 4.4e9: just beyond 2^32
  python tools/pangenome_stream.py --replicas 8         # ONE process, one index build: rbg_replicate_many copies the replica to devices 1..7
                                                        # (peer copies over xGMI), one host thread + HIP stream + read generator per replica,
                                                        # counters through rbg_counters_allreduce_local; host memory independent of the count
  python tools/pangenome_stream.py --replicas 3 --replica-devices 0,0,0    # the same path on one GPU (tests, the builder's box)

The reference's dispatcher for comparison: rb_align.cpp:176-178 (one process, one index, a loop over the reads).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MAXU = 2**64 - 1
# --preset driver: one definition for bench.py and the test (an option given explicitly on the same command line wins)
PRESET_DRIVER = dict(L=100_000_000, H=200, site_rate=0.01, read_len=150, reads=10_000_000, total_reads=50_000_000, check_reads=2000,
                     property_reads=100_000, hbm_reserve_gb=0.0, implicit_text="on", layout="auto", count_only=False)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--L", type=int, default=250_000_000)
    ap.add_argument("--H", type=int, default=200)
    ap.add_argument("--site-rate", type=float, default=0.01)
    ap.add_argument("--seed", type=int, default=20240229)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per batch per GPU")
    ap.add_argument("--total-reads", type=int, default=1_000_000_000, help="reads streamed over all GPUs")
    ap.add_argument("--sub-ppm", type=int, default=100_000, help="reads with one substitution, per million")
    ap.add_argument("--max-hits", type=int, default=-1)
    ap.add_argument("--check-reads", type=int, default=20000)
    ap.add_argument("--property-reads", type=int, default=200_000)
    ap.add_argument("--verify-sa", action="store_true", help="also build the suffix array by prefix doubling and compare (n < 2.5e9 only)")
    ap.add_argument("--count-only", action="store_true")
    ap.add_argument("--layout", choices=("auto", "slots", "runs"), default="auto")
    ap.add_argument("--run-depths", type=lambda v: int(v, 0), default=0, help="RBG_OPT_RUN_DEPTHS: bit d - 1 = keep run lists of the k-mer depth d (run-indexed layout; 0 = all)")
    ap.add_argument("--kmer-steps", type=int, default=0, choices=range(0, 9), help="RBG_OPT_KMER_STEPS (0 = the library's default, 8)")
    ap.add_argument("--run-phi", type=int, default=0, choices=(0, 1, 2), help="RBG_OPT_RUN_PHI: 1 = phi over the list of sampled positions, 2 = phi slots, 0 = the library's choice")
    ap.add_argument("--run-rec", type=int, default=0, choices=(0, 1, 2), help="RBG_OPT_RUN_REC: 1 = directories over the run lists, 2 = bucket records, 0 = the library's choice")
    ap.add_argument("--run-rec-depths", type=lambda v: int(v, 0), default=0, help="RBG_OPT_RUN_REC_DEPTHS: with --run-rec 2, the depths (bit d - 1) that get bucket records (0 = all kept)")
    ap.add_argument("--replica-probe", action="store_true",
                    help="after the per-kernel times: copy the device index once more onto the same device (rbg_replicate: fresh allocations, no composition "
                         "going on around them) and time the search kernel on the copy -- does the PLACEMENT of the arrays matter?")
    ap.add_argument("--ftab-k", type=int, default=-1, help="word length of the device ftab (-1 = the library's choice)")
    ap.add_argument("--hbm-reserve-gb", type=float, default=45.0,
                    help="HBM left to this tool's own buffers (reads, ranges, locations, sort workspace): the index replica gets the rest of "
                         "what is free once the text is resident (0 = the library's default budget, a quarter of the free HBM)")
    ap.add_argument("--implicit-text", choices=("auto", "on", "off"), default="auto",
                    help="sample the reads (and check the properties) from the pangenome's STRUCTURE instead of its text (rbg_sample_reads_pangenome_dev, "
                         "pangenome_bwt.TextView): the same reads byte for byte; auto = when the text would not fit (n > 1e11)")
    ap.add_argument("--out-json", default="", help="also write the JSON line to this file")
    ap.add_argument("--gpus", type=int, default=1,
                    help="GPUs of this node = ranks; N > 1 without RANK in the environment starts the N ranks itself (rowbowt_amd/launch.py)")
    ap.add_argument("--launch-check", action="store_true", help="start the ranks, print what each was given, touch no GPU")
    ap.add_argument("--replicas", type=int, default=0,
                    help="ONE process: the index is built once, rbg_replicate_many copies it to the other devices, one host thread per replica streams "
                         "its rbg_shard_bounds block on its own stream with its own read generator; host memory does not depend on the count")
    ap.add_argument("--replica-devices", default="", help="devices of --replicas, comma separated (default: 0, 1, ...; may repeat: tests put several on one GPU)")
    ap.add_argument("--docs", choices=("on", "off"), default="on",
                    help="attach the document table of the pangenome (one document per haplotype, like the .docs file a pangenome index ships with: "
                         "doclist.hpp:62-65) through rbg_set_docs: K3 then orders its chains by LOCUS (offset inside the document, then document) instead of "
                         "absolute text position -- result-neutral; off = no document table, the absolute order")
    ap.add_argument("--preset", choices=("driver",), default=None,
                    help="driver: BASELINE.json configs[3]'s index shape at the size the driver's own runs carry (bench.py's pangenome_shape block and "
                         "tests/test_gpu_scale.py): a true BWT of r >= 1e8 runs (L = 1e8, H = 200: n = 2.0e10), a DEFAULT rbg_load (no option, no budget), "
                         "5 batches of 10 M x 150 bp device-generated reads, 2 000 reads against the oracle, the properties on 100 000")
    args = ap.parse_args()
    if args.preset == "driver":
        for k_, v_ in PRESET_DRIVER.items():
            if getattr(args, k_) == ap.get_default(k_):   # (an option given explicitly beside the preset wins: A/B runs with fewer check reads)
                setattr(args, k_, v_)
    if args.replicas and args.gpus > 1:
        raise SystemExit("--replicas (one process) and --gpus N (one process per GPU) are two ways to use several GPUs: pick one")
    import importlib.util
    spec = importlib.util.spec_from_file_location("rbg_launch", os.path.join(ROOT, "rowbowt_amd", "launch.py"))
    launch = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(launch)
    if not launch.under_launcher():
        if args.gpus > 1:   # the parent: no torch, no HIP -- N fresh children, one rank each
            raise SystemExit(launch.run_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus, check_devices=not args.launch_check))
        if args.gpus < 1:
            raise SystemExit("--gpus must be at least 1")
        rank, local_rank, world = 0, 0, 1
    else:
        rank, local_rank, world = launch.check_world(args.gpus)
    if args.launch_check:
        launch.echo_rank()
        return
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("needs an MI355X: the hot path has no CPU fallback")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"RCCL group has {dist.get_world_size()} ranks, --gpus asked for {args.gpus}")
        world = dist.get_world_size()

    def log(*a):
        if rank == 0:
            print("[pangenome]", *a, file=sys.stderr, flush=True)

    # ---- a guard for the machine: this tool builds indexes of r = 1e9 runs on a host whose container has a memory limit (cgroup v2
    # memory.max; 300 GiB on the GPU boxes of this pool), and a process that runs into it takes more than itself down.  A thread
    # samples memory.current (and the device's used HBM) once a second, keeps the peaks for the JSON line, and ends the process
    # before the limit is reached.
    import threading
    peaks = {"host_bytes": 0, "hbm_bytes": 0, "host_limit": None}

    def _read_int(path):
        try:
            with open(path) as f:
                t = f.read().strip()
            return None if t == "max" else int(t)
        except Exception:
            return None
    peaks["host_limit"] = _read_int("/sys/fs/cgroup/memory.max")

    def _watch():
        while True:
            cur = _read_int("/sys/fs/cgroup/memory.current")
            if cur is not None:
                peaks["host_bytes"] = max(peaks["host_bytes"], cur)
                if peaks["host_limit"] and cur > 0.88 * peaks["host_limit"]:
                    print(f"[pangenome] host memory {cur / 1e9:.0f} GB of the container's {peaks['host_limit'] / 1e9:.0f} GB: stopping before the limit does",
                          file=sys.stderr, flush=True)
                    os._exit(3)
            try:
                free_b, total_b = torch.cuda.mem_get_info(dev)
                peaks["hbm_bytes"] = max(peaks["hbm_bytes"], total_b - free_b)
            except Exception:
                pass
            time.sleep(1.0)
    threading.Thread(target=_watch, daemon=True).start()

    def mem_line(what):
        log(f"{what}: host {(_read_int('/sys/fs/cgroup/memory.current') or 0) / 1e9:.0f} GB (peak {peaks['host_bytes'] / 1e9:.0f}), "
            f"HBM peak {peaks['hbm_bytes'] / 1e9:.0f} GB")

    import rowbowt_amd as ra
    from rowbowt_amd import capi, shard
    from rowbowt_amd.tools import pangenome_bwt as pb

    Lb = ra.lib()
    m = args.read_len
    max_hits = MAXU if args.max_hits < 0 else args.max_hits

    # ---- synthesis (outside every timed region) ---------------------------------------------------
    t0 = time.time()
    pg = pb.make_pangenome(args.L, args.H, args.site_rate, args.seed, dev)
    inp = pb.build_runs(pg, log=log)
    t_build = time.time() - t0
    import gc
    gc.collect()
    torch.cuda.empty_cache()   # (before anything small is allocated: a live tensor carved out of a cached 60 GB block keeps the whole block)
    log(f"torch after the build: {torch.cuda.memory_allocated(dev) / 1e9:.1f} GB allocated, {torch.cuda.memory_reserved(dev) / 1e9:.1f} GB reserved, "
        f"{torch.cuda.mem_get_info(dev)[0] / 1e9:.1f} GB free on the device")
    log(f"pangenome: L={args.L} H={args.H} sites={pg['n_sites']} n={inp['n']} r={inp['r']} n/r={inp['n'] / inp['r']:.1f} (runs in {t_build:.1f}s)")
    mem_line("after the run-length BWT")
    implicit = args.implicit_text == "on" or (args.implicit_text == "auto" and pg["n"] > 100_000_000_000)
    tv = pb.TextView(pg)
    text = None if implicit else pb.materialize_text(pg)
    if implicit and args.verify_sa:
        raise SystemExit("--verify-sa needs the text")

    def text_at(pos):
        return tv.at(pos) if implicit else text[pos]
    if args.verify_sa:
        from rowbowt_amd.tools import synth_pangenome as sp
        want = sp.index_inputs(text, sp.suffix_array(text))
        same = want["r"] == inp["r"] and all(np.array_equal(want[k], inp[k]) for k in ("heads", "lens", "ssa", "esa"))
        log(f"prefix-doubling suffix array gives the same runs and samples: {same}")
        if not same:
            raise SystemExit("structured BWT builder disagrees with the suffix array")
    unit, H, L, n = pg["unit"], pg["H"], pg["L"], pg["n"]
    del pg
    if not implicit:
        tv = None
    gc.collect()
    torch.cuda.empty_cache()
    log(f"torch before the load: {torch.cuda.memory_allocated(dev) / 1e9:.1f} GB allocated, {torch.cuda.memory_reserved(dev) / 1e9:.1f} GB reserved, "
        f"{torch.cuda.mem_get_info(dev)[0] / 1e9:.1f} GB free on the device")
    if args.ftab_k >= 0:
        capi.set_default_option(capi.OPT_FTAB_K, args.ftab_k)
    if args.layout != "auto":
        capi.set_default_option(capi.OPT_RANK_LAYOUT, {"slots": 1, "runs": 2}[args.layout])
    if args.run_depths:
        capi.set_default_option(capi.OPT_RUN_DEPTHS, args.run_depths)
    if args.run_phi:
        capi.set_default_option(capi.OPT_RUN_PHI, args.run_phi)
    if args.run_rec:
        capi.set_default_option(capi.OPT_RUN_REC, args.run_rec)
    if args.run_rec_depths:
        capi.set_default_option(capi.OPT_RUN_REC_DEPTHS, args.run_rec_depths)
    if args.kmer_steps:
        capi.set_default_option(capi.OPT_KMER_STEPS, args.kmer_steps)
    # a DEFAULT load: no option of the library set by this tool and none through the environment (what a drop-in caller's rbg_load gets)
    default_load = (args.hbm_reserve_gb <= 0 and args.ftab_k < 0 and args.layout == "auto" and not (args.run_depths or args.run_phi or args.run_rec or args.run_rec_depths or args.kmer_steps)
                    and not any(k.startswith("RBG_") for k in os.environ))
    if args.hbm_reserve_gb > 0:
        free_b, _total = torch.cuda.mem_get_info(dev)
        capi.set_default_option(capi.OPT_HBM_BUDGET_MB, max(1024, int((free_b - args.hbm_reserve_gb * 1e9) / 2**20)))
    t0 = time.time()
    rb = ra.RowBowt.from_runs(inp["heads"], inp["lens"], inp["ssa"], inp["esa"], device=local_rank)
    if args.docs == "on":
        rb.set_docs([f"hap{h_}" for h_ in range(H)], [h_ * unit for h_ in range(H)])
    ix = rb.info()
    t_load = time.time() - t0
    mem_line("after the load")
    log(f"index replica: {ix.hbm_bytes / 1e9:.1f} GB HBM, pos_bytes={ix.pos_bytes}, {ix.kmer_steps} symbol(s) per gather "
        f"(asked {ix.kmer_steps_requested}; {ix.hbm_free_at_load / 1e9:.0f} GB free at load, budget {ix.hbm_budget / 1e9:.0f} GB), "
        f"ftab_k={ix.ftab_k}, flatten+upload {t_load:.1f}s")

    # ---- replicas of this process (--replicas G): G - 1 peer copies of the replica just built; the primary is replica 0
    G = max(1, args.replicas)
    rep_devices = [local_rank]
    replicas = [rb]
    t_replicate = 0.0
    if args.replicas:
        rep_devices = [int(x) for x in args.replica_devices.split(",")] if args.replica_devices else list(range(G))
        if len(rep_devices) != G or rep_devices[0] != local_rank or any(d < 0 or d >= torch.cuda.device_count() for d in rep_devices):
            raise SystemExit(f"--replica-devices must name {G} visible devices, the first of them {local_rank} (where the index was built): {rep_devices}")
        t0 = time.time()
        if G > 1:
            replicas += rb.replicate_many(rep_devices[1:])
        torch.cuda.synchronize()
        t_replicate = time.time() - t0
        mem_line(f"after {G - 1} peer cop{'y' if G == 2 else 'ies'} of the replica ({t_replicate:.2f} s)")

    def chk(rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed: {Lb.rbg_strerror(rc).decode()}")

    class Lane:
        """one replica's stream: its block of the global read indices, its buffers, its read generator, its HIP stream"""

        def __init__(self, g, handle, device_index, lanes):
            self.g, self.rb, self.dev = g, handle, torch.device("cuda", device_index)
            self.gb, self.ge = shard.shard_bounds(args.total_reads, rank * lanes + g if lanes > 1 else rank, world * lanes if lanes > 1 else world)
            self.N = min(args.reads, max(1, self.ge - self.gb))
            self.nbatch = (self.ge - self.gb + self.N - 1) // self.N
            dev_l, N = self.dev, self.N
            with torch.cuda.device(dev_l):
                # (the primary's lane runs on the stream the synthesis used; the others get their own)
                self.stream = torch.cuda.current_stream(dev_l) if g == 0 else torch.cuda.Stream(dev_l)
                self.st = self.stream.cuda_stream
                self.d_seqs = torch.zeros(N * m + 32, dtype=torch.uint8, device=dev_l)
                self.d_off = torch.empty(N + 1, dtype=torch.int64, device=dev_l)
                self.d_lo, self.d_hi, self.d_k = (torch.empty(N, dtype=torch.int64, device=dev_l) for _ in range(3))
                self.d_loc_off = torch.empty(N + 1, dtype=torch.int64, device=dev_l)
                self.tmp_bytes = Lb.rbg_locate_plan_tmp_bytes(N)
                self.d_tmp = torch.empty(self.tmp_bytes, dtype=torch.uint8, device=dev_l)
                self.ws_bytes = Lb.rbg_locate_order_ws_bytes(N)
                self.d_ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev_l)
                self.d_locs = None
                # the text (or the structure it is sampled from) on this lane's device: the same tensors where it is the primary's
                if dev_l == dev:
                    self.tv, self.text = tv, text
                elif implicit:
                    import copy
                    self.tv, self.text = copy.copy(tv), None
                    for name in ("base_b", "alt_b", "sites", "G", "site_dir"):
                        t = getattr(tv, name)
                        setattr(self.tv, name, None if t is None else t.to(dev_l))
                else:
                    self.tv, self.text = None, text.to(dev_l)
                torch.cuda.synchronize(dev_l)   # (made on the device's default stream; the lane works on its own)
            self.t_gen = 0.0
            self.t_begin = self.t_end = 0.0
            self.done = 0

        def text_at(self, pos):
            return self.tv.at(pos) if implicit else self.text[pos]

        def gen(self, first, cnt, start_out=None):
            so = start_out.data_ptr() if start_out is not None else None
            t = self.tv
            if implicit:
                chk(Lb.rbg_sample_reads_pangenome_dev(t.base_b.data_ptr(), t.sites.data_ptr(), t.alt_b.data_ptr(), t.G.data_ptr(), t.S,
                                                      t.site_dir.data_ptr() if t.site_dir is not None else None, t.site_dir_shift, unit, H, L, m,
                                                      args.seed + 2, first, cnt, args.sub_ppm, self.d_seqs.data_ptr(), self.d_off.data_ptr(), so, self.st), "sample_reads_pangenome")
            else:
                chk(Lb.rbg_sample_reads_dev(self.text.data_ptr(), unit, H, L, m, args.seed + 2, first, cnt, args.sub_ppm, self.d_seqs.data_ptr(),
                                            self.d_off.data_ptr(), so, self.st), "sample_reads")

        def search(self, cnt):
            if args.count_only:
                chk(Lb.rbg_find_range_dev(self.rb.h, self.d_seqs.data_ptr(), self.d_off.data_ptr(), cnt, self.d_lo.data_ptr(), self.d_hi.data_ptr(), self.st), "find_range")
            else:
                chk(Lb.rbg_find_range_w_toehold_dev(self.rb.h, self.d_seqs.data_ptr(), self.d_off.data_ptr(), cnt, self.d_lo.data_ptr(), self.d_hi.data_ptr(),
                                                    self.d_k.data_ptr(), self.st), "find_range_w_toehold")

        def plan(self, cnt):
            chk(Lb.rbg_locate_plan_dev(self.rb.h, self.d_lo.data_ptr(), self.d_hi.data_ptr(), cnt, max_hits, self.d_loc_off.data_ptr(), self.d_tmp.data_ptr(),
                                       self.tmp_bytes, self.st), "plan")

        def order(self, cnt):
            chk(Lb.rbg_locate_order_dev(self.rb.h, self.d_k.data_ptr(), cnt, self.d_ws.data_ptr(), self.ws_bytes, self.st), "order")

        def fill(self, cnt):
            chk(Lb.rbg_locate_fill_dev(self.rb.h, self.d_lo.data_ptr(), self.d_hi.data_ptr(), self.d_k.data_ptr(), cnt, max_hits, self.d_loc_off.data_ptr(),
                                       self.d_locs.data_ptr(), self.d_ws.data_ptr(), self.st), "fill")

        def size_locs(self):
            """the location buffer sized on the lane's first batch (+25 %); a batch that needs more re-allocates"""
            with torch.cuda.device(self.dev), torch.cuda.stream(self.stream):
                self.gen(self.gb, self.N)
                self.search(self.N)
                if not args.count_only:
                    self.plan(self.N)
                    total = int(self.d_loc_off[-1].item())
                    self.d_locs = torch.empty(int(total * 1.25) + 1024, dtype=torch.int64, device=self.dev)
                    return total
            return 0

        def locate(self, cnt):
            self.plan(cnt)
            total = int(self.d_loc_off[cnt].item())            # (the one host round trip of a batch: the ragged output has to be sized)
            if total > self.d_locs.numel():
                self.d_locs = torch.empty(int(total * 1.25), dtype=torch.int64, device=self.dev)
            self.order(cnt)
            self.fill(cnt)
            return total

        def run(self, gate):
            """the timed stream of this lane (a host thread per lane when there are several): between the two gates"""
            with torch.cuda.device(self.dev), torch.cuda.stream(self.stream):
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                gate()
                self.t_begin = time.perf_counter()
                for b in range(self.nbatch):
                    first = self.gb + b * self.N
                    cnt = min(self.N, self.ge - first)
                    ev[0].record(self.stream)
                    self.gen(first, cnt)
                    ev[1].record(self.stream)
                    self.search(cnt)
                    if not args.count_only:
                        self.locate(cnt)
                    self.stream.synchronize()
                    self.t_gen += ev[0].elapsed_time(ev[1]) * 1e-3
                    self.done += cnt
                self.t_end = time.perf_counter()
                gate()

    lanes = [Lane(g, replicas[g], rep_devices[g], G) for g in range(G)]
    ln0 = lanes[0]
    gb, ge, N, nbatch = ln0.gb, ln0.ge, ln0.N, ln0.nbatch
    stream, st = ln0.stream, ln0.st
    d_seqs, d_off, d_lo, d_hi, d_k, d_loc_off = ln0.d_seqs, ln0.d_off, ln0.d_lo, ln0.d_hi, ln0.d_k, ln0.d_loc_off
    text_at, gen, search, locate = ln0.text_at, ln0.gen, ln0.search, ln0.locate
    for ln in lanes:
        total0 = ln.size_locs()
        if ln.g == 0 and not args.count_only:
            log(f"batch 0: {total0} locations for {N} reads; location buffer {ln.d_locs.numel() * 8 / 1e9:.1f} GB")
    mem_line(f"with the buffers of {G} lane(s)")

    def barrier():
        if use_dist:
            dist.barrier()
        for d in sorted(set(rep_devices)):
            torch.cuda.synchronize(d)

    # per-kernel times of one batch (outside the timed region; HIP events on the launch stream)
    kernel_ms, touched, roof = {}, None, None
    if rank == 0:
        def timed(fn):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            fn()
            e1.record(stream)
            torch.cuda.synchronize()
            return e0.elapsed_time(e1)
        gen(gb, N)
        kernel_ms["sample_reads"] = timed(lambda: gen(gb, N))
        kernel_ms["find_range" if args.count_only else "find_range_w_toehold"] = timed(lambda: search(N))
        if not args.count_only:
            kernel_ms["locate_plan"] = timed(lambda: ln0.plan(N))
            kernel_ms["locate_order"] = timed(lambda: ln0.order(N))
            kernel_ms["locate_fill"] = timed(lambda: ln0.fill(N))
        log("one batch, per kernel (ms): " + ", ".join(f"{k} {v:.2f}" for k, v in kernel_ms.items()))
        if args.replica_probe:
            rep = rb.replicate(local_rank)
            def search_rep():
                chk(Lb.rbg_find_range_w_toehold_dev(rep.h, d_seqs.data_ptr(), d_off.data_ptr(), N, d_lo.data_ptr(), d_hi.data_ptr(), d_k.data_ptr(), st), "find_range_w_toehold")
            search_rep()
            t_rep = min(timed(search_rep) for _ in range(3))
            t_own = min(timed(lambda: search(N)) for _ in range(3))
            log(f"replica probe: find_range_w_toehold {t_own:.2f} ms on the index as loaded, {t_rep:.2f} ms on a copy of it in fresh allocations")
            rep.close()
        # what the search of one batch touched (the instrumented instantiation of the same kernel; include/rbg.h SearchStat): per read
        d_stats = torch.zeros(16, dtype=torch.int64, device=dev)
        chk(Lb.rbg_find_range_stats_dev(rb.h, d_seqs.data_ptr(), d_off.data_ptr(), N, d_lo.data_ptr(), d_hi.data_ptr(),
                                        None if args.count_only else d_k.data_ptr(), d_stats.data_ptr(), st), "find_range_stats")
        torch.cuda.synchronize()
        touched = dict(zip(("steps", "slots", "dense", "searched_ranks", "ftab", "resamples", "read_chunks", "symbols"), (d_stats.cpu().numpy()[:8] / N).round(3).tolist()))
        log("one batch, per read: " + ", ".join(f"{k} {v}" for k, v in touched.items()) +
            "   (run-indexed layout: slots = bucket records or directory gathers, dense = run-list entries scanned, searched_ranks = narrowing rounds)")
        # ---- roofline of this index's own kernels (bench.py's byte model): bytes of the algorithm AS RUN, counted by the
        # instrumented instantiations on this batch, over the HIP-event durations above, against the 8 TB/s HBM peak
        sv = dict(zip(("steps", "slots", "dense", "searched_ranks", "ftab", "resamples", "read_chunks", "symbols"), d_stats.cpu().numpy().tolist()[:8]))
        P_ = int(ix.pos_bytes)
        li_ = rb.layout_info() if int(ix.rank_layout) == 2 else None
        rec_on_ = bool(li_ and any(int(x) for x in li_.rec_bytes))
        ftab_b = 16 if P_ == 4 else 32
        if li_ is not None:
            per_slot = 64 if rec_on_ else (16 if P_ == 8 else 8)
            k2_bytes = (N * (16 + (16 if args.count_only else 24)) + 16 * sv["read_chunks"] + ftab_b * sv["ftab"] + per_slot * sv["slots"]
                        + 8 * sv["dense"] + 28 * sv["searched_ranks"] + (4 if P_ == 4 else 6) * sv["resamples"])
        else:
            k2_bytes = (N * (16 + (16 if args.count_only else 24)) + 16 * sv["read_chunks"] + ftab_b * sv["ftab"] + 16 * sv["slots"]
                        + 2 * sv["dense"] + (8 + 3 * 2 * P_) * sv["searched_ranks"] + (4 + P_) * sv["resamples"])
        k2_name = "find_range" if args.count_only else "find_range_w_toehold"
        roof = {"kernels": {k2_name: {"ms": kernel_ms[k2_name], "alg_bytes": int(k2_bytes)}}}
        if not args.count_only:
            d_stats.zero_()
            chk(Lb.rbg_locate_fill_stats_dev(rb.h, d_lo.data_ptr(), d_hi.data_ptr(), d_k.data_ptr(), N, max_hits, d_loc_off.data_ptr(), ln0.d_locs.data_ptr(),
                                             ln0.d_ws.data_ptr(), d_stats.data_ptr(), st), "locate_fill_stats")
            torch.cuda.synchronize()
            phi_slots_ = li_ is None or int(li_.phi_slots) > 0
            lv = dict(zip(("phi_steps", "phi_searched" if phi_slots_ else "probe_entries", "chains", "locs"), d_stats.cpu().numpy().tolist()[:4]))
            if phi_slots_:   # one PhiSlot per step (16 bytes packed while n < 2^38, else 4 x P), searched steps as on the slot layout
                slot_b_ = 16 if (P_ == 4 or (int(ix.n) >> 38) == 0) else 32
                k3_bytes = N * 28 + slot_b_ * lv["phi_steps"] + (8 + 3 * 2 * P_) * lv["phi_searched"] + 8 * lv["locs"]
            else:
                k3_bytes = N * 28 + (8 if P_ == 4 else 16) * lv["phi_steps"] + (8 if P_ == 4 else 12) * lv["probe_entries"] + 8 * lv["locs"]
            roof["kernels"]["locate_fill"] = {"ms": kernel_ms["locate_fill"], "alg_bytes": int(k3_bytes), "touched": lv}
        for kk, v in roof["kernels"].items():
            v["alg_GBps"] = v["alg_bytes"] / (v["ms"] * 1e-3) / 1e9
            v["frac_of_hbm_peak"] = v["alg_GBps"] / 8000.0
        dom_ = max(roof["kernels"], key=lambda kk: roof["kernels"][kk]["ms"])
        roof.update({"bound": "hbm", "kernel": dom_, "achieved": roof["kernels"][dom_]["alg_GBps"], "peak": 8000.0, "unit": "GB/s",
                     "frac": roof["kernels"][dom_]["frac_of_hbm_peak"], "traffic": None,
                     "note": "achieved = bytes of the algorithm as run (instrumented instantiation of the same kernel on one batch of this stream) / the "
                             "kernel's duration by HIP events on the launch stream; traffic = counter bytes when profiles/pmc_traffic.json holds a pass of "
                             "this workload taken with this librbg.so"})
        try:   # counter traffic: only a committed pass of THIS library on THIS workload counts
            import hashlib
            so_hash = hashlib.sha256(open(os.path.join(ROOT, "rowbowt_amd", "librbg.so"), "rb").read()).hexdigest()
            pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            ent = pj.get(f"pangenome_shape L={args.L} H={args.H} m={m}", {})
            if pj.get("_librbg_sha256") == so_hash and dom_ in ent:
                roof["traffic"] = ent[dom_].get("hbm_bytes_per_launch")
                roof["tcc_miss_per_launch"] = ent[dom_].get("tcc_miss_per_launch")
                roof["traffic_frac_of_hbm_peak"] = roof["traffic"] / (roof["kernels"][dom_]["ms"] * 1e-3) / 1e9 / 8000.0
            roof["librbg_sha256"] = so_hash
        except Exception:
            pass

    # ---- identical per-read outputs on every replica (outside the timed region): the primary's first reads through each copy
    same_on_replicas = None
    if G > 1:
        ncmp = min(N, 100_000)
        with torch.cuda.device(dev), torch.cuda.stream(ln0.stream):
            gen(gb, ncmp)
            search(ncmp)
            tot = 0 if args.count_only else locate(ncmp)
            ln0.stream.synchronize()
            want = [t[:ncmp].cpu() for t in ((d_lo, d_hi) if args.count_only else (d_lo, d_hi, d_k))]
            if not args.count_only:
                want += [ln0.d_loc_off[:ncmp + 1].cpu(), ln0.d_locs[:tot].cpu()]
        same_on_replicas = True
        for ln in lanes[1:]:
            with torch.cuda.device(ln.dev), torch.cuda.stream(ln.stream):
                ln.gen(gb, ncmp)
                ln.search(ncmp)
                tot_g = 0 if args.count_only else ln.locate(ncmp)
                ln.stream.synchronize()
                got = [t[:ncmp].cpu() for t in ((ln.d_lo, ln.d_hi) if args.count_only else (ln.d_lo, ln.d_hi, ln.d_k))]
                if not args.count_only:
                    got += [ln.d_loc_off[:ncmp + 1].cpu(), ln.d_locs[:tot_g].cpu()]
            same_on_replicas = same_on_replicas and len(got) == len(want) and all(a.shape == b.shape and bool((a == b).all()) for a, b in zip(got, want))
        log(f"the first {ncmp} reads give identical ranges, toeholds and locations on all {G} replicas: {same_on_replicas}")

    for r_ in replicas:
        r_.counters_reset()
    barrier()
    if G == 1:
        ln0.run(barrier)
    else:
        gate_obj = threading.Barrier(G)
        errs = []

        def lane_thread(ln):
            try:
                ln.run(gate_obj.wait)
            except BaseException as e:   # noqa: BLE001  (a lane that fails must not leave the others at the gate)
                errs.append(e)
                gate_obj.abort()
        ths = [threading.Thread(target=lane_thread, args=(ln,)) for ln in lanes]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if errs:
            raise errs[0]
        barrier()
    el = max(ln.t_end for ln in lanes) - min(ln.t_begin for ln in lanes)    # max over the replicas, between the gates
    t_gen = max(ln.t_gen for ln in lanes)
    done, ge_all = sum(ln.done for ln in lanes), sum(ln.ge - ln.gb for ln in lanes)
    t_el = torch.tensor([el, t_gen], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
    el, t_gen = float(t_el[0].item()), float(t_el[1].item())
    if G > 1:
        if len(set(rep_devices)) == G:
            g_counters = [int(x) for x in capi.counters_allreduce_local(replicas)]
            reduced_over = f"rbg_counters_allreduce_local: one grouped RCCL all-reduce over the {G} replicas of this process"
        else:   # (several replicas on one device -- tests, the one-GPU box: RCCL wants distinct devices)
            g_counters = [int(x) for x in np.sum([r_.counters().astype(np.int64) for r_ in replicas], axis=0)]
            reduced_over = f"summed on the host over {G} replicas ({len(set(rep_devices))} device(s): RCCL needs one device per rank)"
    else:
        counters = rb.counters().astype(np.int64)
        g_counters = shard.reduce_counters(counters, device=dev)
        reduced_over = f"RCCL all_reduce over {world} rank(s)" if use_dist else "single GPU (no process group)"

    out = None
    if rank == 0:
        out = {
            "metric": f"reads/s ({m} bp, {'count' if args.count_only else 'count+locate'}), streamed",
            "value": args.total_reads / el, "unit": "reads/s", "n_gpus": world * G if args.replicas else world, "higher_is_better": True, "scaling": "strong",
            "value_excluding_read_generation": args.total_reads / max(el - t_gen, 1e-9),
            "seconds": el, "seconds_generating_reads": t_gen, "batches_per_gpu": nbatch, "reads_per_batch": N,
            "kernel_ms_one_batch": kernel_ms, "search_touched_per_read": touched, "roofline": roof,
            "dtype": "u64" if ix.pos_bytes == 8 else "u32/u64", "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[3] shape: {args.total_reads} synthetic {m} bp reads generated on the device per batch "
                                   f"(counter-based RNG), {'find_range' if args.count_only else 'find_range_w_toehold + locs_at'}, "
                                   f"index replicated x{world * G}" + (f" in ONE process (built once, rbg_replicate_many to devices {rep_devices})" if args.replicas else "") +
                                   ", reads sharded by contiguous global index (rbg_shard_bounds), no data-path collective, counters reduced at the end",
                       "index": {"L": args.L, "H": args.H, "n": int(n), "r": int(inp["r"]), "site_rate": args.site_rate, "true_bwt": True,
                                 "builder": "rowbowt_amd/tools/pangenome_bwt.py", "hbm_bytes": int(ix.hbm_bytes), "pos_bytes": int(ix.pos_bytes),
                                 "symbols_per_gather": int(ix.kmer_steps), "symbols_per_gather_requested": int(ix.kmer_steps_requested),
                                 "kmer_depths_with_tables": [d + 1 for d in range(8) if int(ix.depth_runs[d])],
                                 "hbm_free_at_load": int(ix.hbm_free_at_load), "hbm_budget": int(ix.hbm_budget),
                                 "rank_bucket_shift": int(ix.rank_bucket_shift), "phi_bucket_shift": int(ix.phi_bucket_shift),
                                 "rank_layout": int(ix.rank_layout), "ftab_k": int(ix.ftab_k),
                                 "build_runs_s": t_build, "flatten_upload_s": t_load}},
            "counters": {"reads": g_counters[0], "matched": g_counters[1], "sum_occ": g_counters[2], "sum_locs": g_counters[3],
                         "reduced_over": reduced_over},
        }
        if args.replicas:
            out["replicas"] = {"formed": G, "devices": rep_devices, "replicate_s": t_replicate, "identical_outputs_on_every_replica": same_on_replicas,
                               "per_replica": [{"device": rep_devices[ln.g], "reads": ln.done, "batches": ln.nbatch, "ms": (ln.t_end - ln.t_begin) * 1e3,
                                                "ms_generating_reads": ln.t_gen * 1e3} for ln in lanes],
                               "timing": "max over the replicas between two gates (threading.Barrier; one host thread and one HIP stream per replica)"}
            if same_on_replicas is False:
                print(json.dumps(out))
                raise SystemExit("REPLICA MISMATCH: a copy answers differently from the primary")
        if done != ge_all:
            print(json.dumps(out))
            raise SystemExit(f"{done} reads streamed by this process, {ge_all} expected")
        if g_counters[0] != args.total_reads:
            print(json.dumps(out))
            raise SystemExit(f"counter mismatch: {g_counters[0]} reads counted, {args.total_reads} streamed")

    # ---- properties + parity on the first reads of this rank's stream (outside the timed region) ---------
    if rank == 0 and (args.property_reads > 0 or args.check_reads > 0):
        npr = min(max(args.property_reads, args.check_reads), N)
        d_start = torch.empty(npr, dtype=torch.int64, device=dev)
        gen(gb, npr, d_start)
        search(npr)
        total = 0 if args.count_only else locate(npr)
        torch.cuda.synchronize()
        reads = d_seqs[:npr * m].view(npr, m)
        lo_t, hi_t = d_lo[:npr], d_hi[:npr]
        # an unmutated read must be found, and its own text position must be inside its match set
        same_as_text = (text_at(d_start[:, None] + torch.arange(m, device=dev)[None, :]) == reads).all(dim=1)
        ok_found = bool((hi_t >= lo_t)[same_as_text].all().item())
        ok_empty = bool(((hi_t >= lo_t) | ((lo_t == 1) & (hi_t == 0))).all().item())
        props = {"reads": npr, "unmutated_reads_all_found": ok_found, "empty_is_{1,0}": ok_empty, "unmutated": int(same_as_text.sum().item())}
        if not args.count_only:
            offs = d_loc_off[:npr + 1]
            occ_t = offs[1:] - offs[:-1]
            ok_occ = bool((torch.where(hi_t >= lo_t, hi_t - lo_t + 1, torch.zeros_like(lo_t)).clamp(max=max_hits if max_hits < 2**63 else 2**63 - 1) == occ_t).all().item())
            ridx = torch.repeat_interleave(torch.arange(npr, device=dev), occ_t)
            locs_t = ln0.d_locs[:total]
            bad = torch.zeros(total, dtype=torch.bool, device=dev)
            for j in range(m):
                bad |= text_at(locs_t + j) != reads[ridx, j]
            ok_match = not bool(bad.any().item())
            key = ridx * (int(n) + 1) + locs_t
            ok_distinct = int(torch.unique(key).numel()) == total
            # the read's own position is among its locations (uncapped runs only)
            own = torch.zeros(npr, dtype=torch.bool, device=dev)
            own.index_put_((ridx[locs_t == d_start[ridx]],), torch.tensor(True, device=dev))
            ok_own = bool(own[same_as_text].all().item()) if max_hits == MAXU else None
            props.update({"locations": total, "every_location_is_an_occurrence": ok_match, "locations_distinct": ok_distinct,
                          "occ_equals_range_width": ok_occ, "own_position_reported": ok_own})
            del ridx, locs_t, bad, key
        out["properties"] = props
        if not all(v for k_, v in props.items() if isinstance(v, bool)):
            print(json.dumps(out))
            raise SystemExit("PROPERTY FAILURE")
        nchk = min(args.check_reads, npr)
        if nchk:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import orc  # oracle: the checker only

            t0 = time.time()
            o = orc.Oracle.from_runs(inp["heads"], inp["lens"], inp["ssa"], inp["esa"])
            log(f"oracle built in {time.time() - t0:.1f}s")
            mem_line("with the oracle")
            h_seqs = reads[:nchk].cpu().numpy().reshape(-1)
            h_off = np.arange(nchk + 1, dtype=np.uint64) * m
            ncpu = min(os.cpu_count() or 1, 64)
            g_lo = d_lo[:nchk].cpu().numpy().view(np.uint64)
            g_hi = d_hi[:nchk].cpu().numpy().view(np.uint64)
            if args.count_only:
                wlo, whi = o.find_range_batch(h_seqs, h_off, nthreads=ncpu)
                ok = bool((g_lo == wlo).all() and (g_hi == whi).all())
                out["parity"] = {"reads_checked": nchk, "bit_exact_vs_oracle": ok}
            else:
                g_k = d_k[:nchk].cpu().numpy().view(np.uint64)
                g_off = d_loc_off[:nchk + 1].cpu().numpy().view(np.uint64)
                g_locs = ln0.d_locs[:int(g_off[-1])].cpu().numpy().view(np.uint64)
                wlo, whi, wk = o.find_range_w_toehold_batch(h_seqs, h_off, nthreads=ncpu)
                woff, wlocs = o.locs_at_batch(wlo, whi, wk, max_hits, nthreads=ncpu)
                ok = bool((g_lo == wlo).all() and (g_hi == whi).all() and (g_k == wk).all() and (g_off == woff).all() and (g_locs == wlocs).all())
                # the count-only kernel on the same reads
                chk(Lb.rbg_find_range_dev(rb.h, d_seqs.data_ptr(), d_off.data_ptr(), nchk, d_lo.data_ptr(), d_hi.data_ptr(), st), "find_range")
                torch.cuda.synchronize()
                ok1 = bool((d_lo[:nchk].cpu().numpy().view(np.uint64) == wlo).all() and (d_hi[:nchk].cpu().numpy().view(np.uint64) == whi).all())
                out["parity"] = {"reads_checked": nchk, "locs_checked": int(woff[-1]), "toeholds_above_2^32": int((wk[whi >= wlo] >= 2**32).sum()),
                                 "bit_exact_vs_oracle": ok and ok1, "count_only_kernel_bit_exact": ok1}
            o.close()
            if not out["parity"]["bit_exact_vs_oracle"]:
                print(json.dumps(out))
                raise SystemExit("PARITY FAILURE: HIP path disagrees with the oracle")
    if rank == 0:
        li = rb.layout_info()
        out["config"]["index"]["layout_info"] = {"run_fmt": int(li.run_fmt), "depth_mask_kept": int(li.depth_mask_kept), "depths_dropped_budget": int(li.depths_dropped_budget),
                                                 "rank_directories": int(li.rank_directories), "phi_directory": int(li.phi_directory),
                                                 "entries": [int(x) for x in li.entries], "fillers": [int(x) for x in li.fillers], "dir_bytes": [int(x) for x in li.dir_bytes],
                                                 "phi_entries": int(li.phi_entries), "phi_fillers": int(li.phi_fillers), "phi_dir_bytes": int(li.phi_dir_bytes),
                                                 "phi_dir_shift": int(li.phi_dir_shift), "phi_slots": int(li.phi_slots), "phi_slot_bytes": int(li.phi_slot_bytes),
                                                 "rec_bytes": [int(x) for x in li.rec_bytes], "rec_overflow": [int(x) for x in li.rec_overflow],
                                                 "budget_raised": int(li.budget_raised), "depths_kept": [d + 1 for d in range(8) if int(li.depth_mask_kept) >> d & 1],
                                                 "depths_with_records": [d + 1 for d in range(8) if int(li.rec_bytes[d])]}
        out["config"]["index"]["default_load"] = default_load
        out["config"]["index"]["documents"] = H if args.docs == "on" else 0
        locus_on = args.docs == "on" and os.environ.get("RBG_LOCATE_ORDER") != "abs" and (H >= 128 or os.environ.get("RBG_LOCATE_ORDER") == "locus")
        out["config"]["chain_order"] = ("by locus: offset inside the document, then document (rbg_set_docs: one document per haplotype; from 128 documents on)"
                                        if locus_on else "by absolute text position")
        out["config"]["index"]["text"] = "implicit (sampled from the pangenome's structure)" if implicit else "materialised in HBM"
        out["peaks"] = {"host_bytes": int(peaks["host_bytes"]), "host_limit_bytes": peaks["host_limit"], "hbm_bytes": int(peaks["hbm_bytes"])}
        print(json.dumps(out), flush=True)
        if args.out_json:
            with open(args.out_json, "w") as f:
                f.write(json.dumps(out) + "\n")
    for r_ in replicas[1:]:
        r_.close()
    rb.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
