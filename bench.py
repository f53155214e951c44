#!/usr/bin/env python3
"""bench.py -- reads/s of the rb_align hot path (exact-match count + locate) on MI355X.

One "step" = one pass of the hot path over one batch of synthetic reads already resident in HBM:
    K2  rbg_find_range_w_toehold_dev   (RowBowt::find_range_w_toehold, rowbowt.hpp:169-184)
    K3a rbg_locate_plan_dev            (occ + exclusive scan)
    K3b rbg_locate_order_dev           (radix sort of the toeholds: chain order for locality; result-neutral)
    K3c rbg_locate_fill_dev            (RowBowt::locs_at -> ToeholdSA::locate_range, toehold_sa.hpp:37-49)
which is BASELINE.json configs[2] ("1xMI355X count+locate: chr22-scale index, 10M reads"), the
configuration the metric "reads/s (100 bp, count+locate)" is quoted on.  The count-only rate
(configs[1], rbg_find_range_dev) is measured in the same run and reported beside it.

The index is SYNTHETIC (no real data here): rowbowt_amd/tools/synth_pangenome.py builds a
chr22-scale pangenome text (L x H), its suffix array (torch sorts on the GPU), and from that the
run-length BWT + run-boundary SA samples that rbg_build_from_runs flattens into HBM.  Index
construction, read synthesis, oracle construction and the CPU baseline are all OUTSIDE the timed
region.  Multi-GPU: the index is replicated, reads are sharded (each rank its own batch: weak
scaling), no data-path collective; one RCCL all-reduce carries the 4 global counters.

Launch: python bench.py [--gpus N --steps K --warmup W].  With N > 1 and no RANK in the environment this
process only LAUNCHES: before importing torch or touching HIP it starts N fresh children of itself (one rank
per GPU, rowbowt_amd/launch.py), relays rank 0's JSON line and exits non-zero if any rank does.  Under
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` the ranks already exist and --gpus
must equal WORLD_SIZE.  `n_gpus` in the line is the size of the RCCL group that was actually formed.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
MAXU = 2**64 - 1
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def load_launch():
    """rowbowt_amd/launch.py by file path: stdlib only, so the launching parent imports neither torch nor the
    package's ctypes binding"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("rbg_launch", os.path.join(ROOT, "rowbowt_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1,
                    help="GPUs of this node = ranks; N > 1 without RANK in the environment starts the N ranks itself")
    ap.add_argument("--launch-check", action="store_true",
                    help="start the ranks as --gpus asks, have each print the environment it was given as one JSON line and exit "
                         "(no GPU is touched: the launcher's own test)")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU per step")
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--L", type=int, default=40_000_000, help="base sequence length")
    ap.add_argument("--H", type=int, default=50, help="haplotypes")
    ap.add_argument("--site-rate", type=float, default=0.01)
    ap.add_argument("--seed", type=int, default=20240229)
    ap.add_argument("--max-hits", type=int, default=-1, help="-1 = 2^64-1 like rb_align.cpp:125")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check-reads", type=int, default=20000, help="reads compared bit-exactly with the oracle")
    ap.add_argument("--markers", action="store_true",
                    help="also run BASELINE.json configs[4] (rb_align -m: find_range + markers_at on a synthetic marker array "
                         "over the same index) and rb_markers' seeding kernel, checked against the oracle on --check-reads reads; "
                         "on by default on one GPU (about 25 s), off under torch.distributed.run unless asked for")
    ap.add_argument("--no-markers", action="store_true", help="skip the markers leg")
    ap.add_argument("--two-stream", action="store_true",
                    help="also time the K steps as successive batches on two HIP streams (informational; off by default so "
                         "that a profile of the default command holds undisturbed per-kernel durations)")
    ap.add_argument("--no-space-speed", action="store_true",
                    help="skip the space_speed block (the replica rebuilt at 4, 3, 2 and 1 symbols per gather; about a minute)")
    ap.add_argument("--pos-bytes", type=int, default=0, choices=(0, 4, 8), help="force the position width of the HBM layout (RBG_OPT_POS_BYTES); 0 = by n")
    ap.add_argument("--layout", default="auto", choices=("auto", "slots", "runs"), help="RBG_OPT_RANK_LAYOUT of the headline replica")
    ap.add_argument("--hbm-budget-gb", type=float, default=0.0,
                    help="RBG_OPT_HBM_BUDGET_MB of the headline replica: 0 (default) = the library's own default (a quarter of the free HBM: what a drop-in "
                         "caller gets -- since round 4 the run-indexed layout, 8.7 GB for the bench index, and the fastest row of space_speed); -1 = three "
                         "quarters of the HBM that is free at load (slot tables, five symbols per gather: 221 GB, the headline of rounds 1-3)")
    ap.add_argument("--rehearse-ranks", action="store_true",
                    help="test mode, never a measurement: the N ranks of --gpus share the GPUs that exist (rank % devices) and meet over gloo, so that "
                         "the multi-rank path -- launcher, index built once and read from the cache file by every rank, barriers, max-over-ranks "
                         "timing, counters reduced -- runs on a one-GPU box; the line says so and carries no vs_baseline")
    ap.add_argument("--via-cache", action="store_true",
                    help="build the replica through the native cache file even on one GPU (with --gpus N > 1 every rank does: rank 0 "
                         "writes it to node-local shared memory once, all ranks load it)")
    ap.add_argument("--replicas", type=int, default=0,
                    help="ONE process with G replicas (rbg_replicate_many: the index is built once, peer-copied to the other devices), a host thread + HIP stream + "
                         "read batch per replica; the line's value / n_gpus / ms_per_step then describe the G replicas (weak scaling: --reads per replica). "
                         "The driver's N-GPU contract stays one process per GPU (--gpus N)")
    ap.add_argument("--replica-devices", default="", help="devices of --replicas, comma separated (default 0, 1, ...; may repeat)")
    ap.add_argument("--docs", choices=("on", "off"), default="off",
                    help="attach the pangenome's document table (one document per haplotype: what the .docs file of a pangenome index lists, doclist.hpp:62-65) "
                         "through rbg_set_docs before the timed steps: K3 then orders its chains by locus (offset inside the document, then document) instead "
                         "of absolute text position -- result-neutral")
    ap.add_argument("--no-pangenome-shape", action="store_true",
                    help="skip the pangenome_shape block: BASELINE.json configs[3]'s index shape on one GPU (tools/pangenome_stream.py --preset driver in a "
                         "child process once this process has given its HBM back: a true BWT of r = 1.2e8 runs, a default rbg_load, 150 bp device-generated "
                         "reads, its own roofline and parity sample; about 50 s).  On by default on one GPU with the default workload, off under a launcher")
    ap.add_argument("--property-reads", type=int, default=1_000_000,
                    help="reads whose every reported location is checked against the text on the GPU (size-independent property)")
    return ap.parse_args()


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


# What this rank is doing, for the post-mortem of a failed multi-GPU run: a rank that dies names its stage on stderr
# (under rowbowt_amd/launch.py the parent then stops the others, which would otherwise wait in a collective for ever).
STAGE = {"name": "start", "t0": time.time(), "rank": int(os.environ.get("RANK", "0")), "trace": os.environ.get("RBG_BENCH_TRACE", "") not in ("", "0")}


def stage(name):
    STAGE["name"] = name
    if STAGE["trace"]:
        print(f"[bench] rank {STAGE['rank']} +{time.time() - STAGE['t0']:.1f}s: {name}", file=sys.stderr, flush=True)


class DeviceClocks:
    """The GPU's clocks, power and temperature while the timed steps run, sampled from sysfs by a thread (the amdgpu driver's pp_dpm_* / hwmon files of the
    card whose PCI address is the torch device's; a few reads every 5 ms, no GPU call) -- so that a bench line says under which clocks its kernel times were taken
    (K3 differs by 0.2-0.5 ms between boxes on the same code).  Everything is best effort: where the files are missing the block is None."""

    def __init__(self, torch, dev):
        import glob
        import threading
        self.dir, self.hwmon, self.samples, self._stop, self._th = None, None, [], threading.Event(), None
        try:
            pr = torch.cuda.get_device_properties(dev)
            want = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
            for c in glob.glob("/sys/class/drm/card*/device"):
                if want in os.path.realpath(c).lower() and os.path.exists(os.path.join(c, "pp_dpm_sclk")):
                    self.dir = c
                    hw = glob.glob(os.path.join(c, "hwmon", "hwmon*"))
                    self.hwmon = hw[0] if hw else None
                    break
        except Exception:
            self.dir = None

    @staticmethod
    def _cur_mhz(path):
        try:
            for ln in open(path):
                if "*" in ln:
                    return float(ln.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
        except Exception:
            pass
        return None

    @staticmethod
    def _num(path, scale):
        try:
            return float(open(path).read().strip()) * scale
        except Exception:
            return None

    def _sample(self):
        d = {k: self._cur_mhz(os.path.join(self.dir, f"pp_dpm_{k}")) for k in ("sclk", "mclk", "fclk")}
        if self.hwmon:
            d["power_w"] = self._num(os.path.join(self.hwmon, "power1_input"), 1e-6)
            d["temp_c"] = self._num(os.path.join(self.hwmon, "temp2_input"), 1e-3)
        return d

    def start(self):
        if not self.dir:
            return
        import threading

        def loop():
            while not self._stop.is_set():
                self.samples.append(self._sample())
                self._stop.wait(0.005)
        self._th = threading.Thread(target=loop, daemon=True)
        self._th.start()

    def stop(self):
        if not self._th:
            return None
        self._stop.set()
        self._th.join(timeout=2)
        out = {"samples": len(self.samples), "source": self.dir + "/pp_dpm_{sclk,mclk,fclk} (+ hwmon power1_input, temp2_input), every 5 ms over the timed steps"}
        if self.hwmon:
            out["power_cap_w"] = self._num(os.path.join(self.hwmon, "power1_cap"), 1e-6)
        for k in ("sclk", "mclk", "fclk", "power_w", "temp_c"):
            v = sorted(x[k] for x in self.samples if x.get(k) is not None)
            if v:
                out[k if k.endswith(("_w", "_c")) else k + "_mhz"] = {"min": v[0], "median": v[len(v) // 2], "max": v[-1]}
        return out


def cpu_budget():
    """CPUs this process may use: the smaller of the visible ones and the container's CPU quota (cgroup v2 cpu.max)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        a, b = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if a != "max":
            n = min(n, max(1, int(float(a) / float(b) + 0.5)))
    except Exception:
        pass
    return max(1, n)


def run():
    args = parse()
    launch = load_launch()
    if not launch.under_launcher():
        if args.gpus > 1:
            # the parent: no torch, no HIP -- N fresh children, one rank each (never re-exec a process that holds a GPU)
            raise SystemExit(launch.run_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus, check_devices=not (args.launch_check or args.rehearse_ranks)))
        if args.gpus < 1:
            raise SystemExit("--gpus must be at least 1")
        rank, local_rank, world = 0, 0, 1
    else:
        rank, local_rank, world = launch.check_world(args.gpus)   # refuses --gpus != WORLD_SIZE
    if args.launch_check:
        launch.echo_rank()
        return
    # the markers leg (BASELINE.json configs[4]) rides along on one GPU unless switched off
    args.markers = (args.markers or (world == 1 and "RANK" not in os.environ)) and not args.no_markers
    if args.replicas and world > 1:
        raise SystemExit("--replicas (one process, G replicas) and --gpus N (one process per GPU) are two ways to use several GPUs: pick one")
    if args.replicas and not args.no_space_speed:
        # (the space_speed block rebuilds the replica in other forms; the --replicas leg replicates and scales the HEADLINE replica)
        args.no_space_speed = True
        log(rank, "--replicas: skipping the space_speed block so that the replicated index is the headline replica")
    STAGE["rank"] = rank
    stage("import torch")
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if args.rehearse_ranks:
        local_rank = local_rank % torch.cuda.device_count()
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible -- "
                         f"--gpus {args.gpus} exceeds this node")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if args.rehearse_ranks else dev   # where the tensors of the collectives live (gloo: host)
    use_dist = world > 1 or "RANK" in os.environ  # under torch.distributed.run: always go through RCCL
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        stage(f"init_process_group ({'gloo' if args.rehearse_ranks else 'nccl = RCCL'}, world {world}, device {local_rank}, "
              f"{os.environ['MASTER_ADDR']}:{os.environ['MASTER_PORT']})")
        if args.rehearse_ranks:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # nccl == RCCL on ROCm
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"RCCL group has {dist.get_world_size()} ranks, --gpus asked for {args.gpus}")
        world = dist.get_world_size()   # n_gpus of the line = the group that was actually formed

    stage("load librbg.so")
    import rowbowt_amd as ra
    from rowbowt_amd import shard
    from rowbowt_amd.tools import synth_pangenome as sp

    L = ra.lib()
    if args.pos_bytes or args.layout != "auto":
        from rowbowt_amd import capi as _c
        if args.pos_bytes:
            _c.set_default_option(_c.OPT_POS_BYTES, args.pos_bytes)
        if args.layout != "auto":
            _c.set_default_option(_c.OPT_RANK_LAYOUT, _c.LAYOUT_RUNS if args.layout == "runs" else _c.LAYOUT_SLOTS)
    m = args.read_len
    max_hits = MAXU if args.max_hits < 0 else args.max_hits
    from rowbowt_amd import capi as _cb
    free_at_start, _tot = torch.cuda.mem_get_info(dev)
    if args.hbm_budget_gb < 0:
        budget_mb = int(free_at_start * 3 // 4) >> 20
    else:
        budget_mb = int(args.hbm_budget_gb * 1e9) >> 20
    _cb.set_default_option(_cb.OPT_HBM_BUDGET_MB, budget_mb)

    # ---- synthesis (outside the timed region) --------------------------------------------------
    t0 = time.time()
    stage("synthesise the text")
    text, info = sp.make_text(args.L, args.H, args.site_rate, args.seed, dev)   # (every rank: its reads are sampled from it)
    # The index is made ONCE per node: rank 0 builds the suffix array and the run-length BWT and writes the native cache
    # file (rbg_convert_runs) to node-local shared memory; every rank then loads its replica from it (rbg_load_cache).
    cache_path = None
    if world > 1 or args.via_cache:
        shm = "/dev/shm" if os.path.isdir("/dev/shm") else (os.environ.get("TMPDIR") or "/tmp")
        cache_path = os.path.join(shm, f"rbg_bench_{os.environ.get('MASTER_PORT', '0')}_{args.L}_{args.H}_{args.seed}.rbgpu")
    inp, marker_arrays, t_sa, t_cache_write, cache_bytes = None, None, 0.0, 0.0, 0
    if rank == 0 or world == 1:
        stage("suffix array + run-length BWT (rank 0)")
        sa = sp.suffix_array(text)
        torch.cuda.synchronize()
        t_sa = time.time() - t0
        inp = sp.index_inputs(text, sa)
        marker_arrays = sp.marker_array(text, info, sa, w=10) if args.markers else None
        del sa
        torch.cuda.empty_cache()
        log(rank, f"synthetic pangenome: L={args.L} H={args.H} n={inp['n']} r={inp['r']} n/r={inp['n'] / inp['r']:.1f} "
                  f"(suffix array {t_sa:.1f}s)")
        if cache_path:
            from rowbowt_amd import capi as _capi
            stage(f"write the native cache file {cache_path} (rank 0)")
            tc = time.time()
            _capi.convert_runs(inp["heads"], inp["lens"], inp["ssa"], inp["esa"], out_path=cache_path)
            t_cache_write, cache_bytes = time.time() - tc, os.path.getsize(cache_path)
    t_wait = time.time()
    if cache_path:
        if use_dist:
            stage("barrier: wait for rank 0's cache file")
            dist.barrier()   # the file is complete
        t0 = time.time()
        stage(f"rbg_load_cache({cache_path}) on device {local_rank}")
        rb = ra.RowBowt.from_cache(cache_path, ra.LoadRbwtFlag.SA, device=local_rank)
        t_load = time.time() - t0
        if use_dist:
            stage("barrier: every rank has read the cache file")
            dist.barrier()   # every rank has read it
        if rank == 0:
            os.unlink(cache_path)
    else:
        t0 = time.time()
        stage(f"rbg_build_from_runs on device {local_rank}")
        rb = ra.RowBowt.from_runs(inp["heads"], inp["lens"], inp["ssa"], inp["esa"], device=local_rank)
        t_load = time.time() - t0
    t_wait = t0 - t_wait   # (ranks > 0: the time spent waiting for rank 0's suffix array and cache file)
    if args.docs == "on":
        rb.set_docs([f"hap{h_}" for h_ in range(info["H"])], [h_ * info["unit"] for h_ in range(info["H"])])
    ix = rb.info()
    if inp is None:   # (ranks > 0: what the line's config block and the checks need is on rank 0 only)
        inp = {"n": int(ix.n), "r": int(ix.r)}
    log(rank, f"index replica: {ix.hbm_bytes / 1e6:.1f} MB HBM, pos_bytes={ix.pos_bytes}, sigma={ix.sigma}, "
              f"flatten+upload {time.time() - t0:.1f}s")

    # weak scaling: the global batch is world x --reads; this rank owns the contiguous block
    # shard_bounds() gives it and synthesises exactly those reads
    gb, ge = shard.shard_bounds(args.reads * world, rank, world)
    N = ge - gb
    stage("sample this rank's reads")
    reads, _ = sp.sample_reads(text, info, N, m, seed=args.seed + 2 + rank, sub_rate=0.1)
    torch.cuda.empty_cache()
    d_seqs = reads.reshape(-1)
    if d_seqs.numel() % 16:  # reads are fetched as aligned 16-byte chunks
        d_seqs = torch.cat([d_seqs, torch.zeros(16 - d_seqs.numel() % 16, dtype=torch.uint8, device=dev)])
    d_off = (torch.arange(N + 1, device=dev, dtype=torch.int64) * m)
    d_lo, d_hi, d_k = (torch.empty(N, dtype=torch.int64, device=dev) for _ in range(3))
    d_loc_off = torch.empty(N + 1, dtype=torch.int64, device=dev)
    tmp_bytes = L.rbg_locate_plan_tmp_bytes(N)
    d_tmp = torch.empty(tmp_bytes, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream()
    st = stream.cuda_stream

    def chk(rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed: {L.rbg_strerror(rc).decode()}")

    def k_count():
        chk(L.rbg_find_range_dev(rb.h, d_seqs.data_ptr(), d_off.data_ptr(), N, d_lo.data_ptr(), d_hi.data_ptr(), st), "find_range")

    def k_toehold():
        chk(L.rbg_find_range_w_toehold_dev(rb.h, d_seqs.data_ptr(), d_off.data_ptr(), N, d_lo.data_ptr(), d_hi.data_ptr(),
                                           d_k.data_ptr(), st), "find_range_w_toehold")

    def k_plan():
        chk(L.rbg_locate_plan_dev(rb.h, d_lo.data_ptr(), d_hi.data_ptr(), N, max_hits, d_loc_off.data_ptr(),
                                  d_tmp.data_ptr(), tmp_bytes, st), "locate_plan")

    # size the ragged output once (same reads every step => same total)
    stage("first launch of the hot path (sizes the ragged output)")
    k_toehold()
    k_plan()
    total_locs = int(d_loc_off[-1].item())
    d_locs = torch.empty(max(total_locs, 1), dtype=torch.int64, device=dev)
    n_matched = int((d_hi.view(torch.int64) >= d_lo.view(torch.int64)).sum().item())
    log(rank, f"reads/GPU={N} x {m} bp, matched={n_matched}, sum occ={total_locs} (mean occ/matched read "
              f"{total_locs / max(n_matched, 1):.1f})")

    ws_bytes = L.rbg_locate_order_ws_bytes(N)
    d_ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)

    def k_order():
        chk(L.rbg_locate_order_dev(rb.h, d_k.data_ptr(), N, d_ws.data_ptr(), ws_bytes, st), "locate_order")

    def k_fill():
        chk(L.rbg_locate_fill_dev(rb.h, d_lo.data_ptr(), d_hi.data_ptr(), d_k.data_ptr(), N, max_hits,
                                  d_loc_off.data_ptr(), d_locs.data_ptr(), d_ws.data_ptr(), st), "locate_fill")

    def step():
        k_toehold()
        k_plan()
        k_order()
        k_fill()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps, events=None):
        barrier()
        t = time.perf_counter()
        for s in range(steps):
            fn(s) if events else fn()
        barrier()
        return time.perf_counter() - t

    # ---- count+locate (headline) ---------------------------------------------------------------
    stage("warmup + the timed steps (barrier on both sides)")
    for _ in range(args.warmup):
        step()
    K = args.steps
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(K)]

    def step_ev(s):
        ev[s][0].record(stream)
        k_toehold()
        ev[s][1].record(stream)
        k_plan()
        ev[s][2].record(stream)
        k_order()
        ev[s][4].record(stream)
        k_fill()
        ev[s][3].record(stream)

    rb.counters_reset()
    clocks = DeviceClocks(torch, dev)
    clocks.start()
    el = timed(step_ev, K, events=True)
    device_state = clocks.stop()
    counters = rb.counters().astype(np.int64)
    ms_toe = float(np.mean([ev[s][0].elapsed_time(ev[s][1]) for s in range(K)]))
    ms_plan = float(np.mean([ev[s][1].elapsed_time(ev[s][2]) for s in range(K)]))
    ms_order = float(np.mean([ev[s][2].elapsed_time(ev[s][4]) for s in range(K)]))
    ms_fill = float(np.mean([ev[s][4].elapsed_time(ev[s][3]) for s in range(K)]))

    # ---- count-only (configs[1]) ---------------------------------------------------------------
    stage("count-only steps")
    for _ in range(max(1, args.warmup)):
        k_count()
    evc = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(K)]

    def count_ev(s):
        evc[s][0].record(stream)
        k_count()
        evc[s][1].record(stream)

    el_count = timed(count_ev, K, events=True)
    ms_count = float(np.mean([evc[s][0].elapsed_time(evc[s][1]) for s in range(K)]))

    # ---- capped locate (SURVEY 8d config 3's diagnostic): the same step with max_hits = 1 ----------
    d_loc_off1 = torch.empty(N + 1, dtype=torch.int64, device=dev)
    d_locs1 = torch.empty(N, dtype=torch.int64, device=dev)

    def step_capped():
        k_toehold()
        chk(L.rbg_locate_plan_dev(rb.h, d_lo.data_ptr(), d_hi.data_ptr(), N, 1, d_loc_off1.data_ptr(), d_tmp.data_ptr(), tmp_bytes, st), "locate_plan")
        k_order()
        chk(L.rbg_locate_fill_dev(rb.h, d_lo.data_ptr(), d_hi.data_ptr(), d_k.data_ptr(), N, 1, d_loc_off1.data_ptr(), d_locs1.data_ptr(),
                                  d_ws.data_ptr(), st), "locate_fill")

    step_capped()
    el_cap = timed(step_capped, K)
    capped_ok = bool((d_locs1[:int(d_loc_off1[-1].item())] == d_k[d_hi >= d_lo]).all().item())  # the first location is the toehold
    del d_loc_off1, d_locs1
    step()  # leave the uncapped results in the buffers for the checks below

    # ---- the same step storing 4-byte locations (rbg_locate_fill_dev32: device pipelines on an index with 4-byte
    # positions; informational -- `value` keeps the API's 64-bit locations, toehold_sa.hpp:37-49) -----------------
    u32_block = None
    if ix.pos_bytes == 4:
        d_locs32 = torch.empty(max(total_locs, 1), dtype=torch.int32, device=dev)

        def k_fill32():
            chk(L.rbg_locate_fill_dev32(rb.h, d_lo.data_ptr(), d_hi.data_ptr(), d_k.data_ptr(), N, max_hits, d_loc_off.data_ptr(),
                                        d_locs32.data_ptr(), d_ws.data_ptr(), st), "locate_fill_dev32")

        def step32():
            k_toehold()
            k_plan()
            k_order()
            k_fill32()

        step32()
        el32 = timed(step32, K)
        ev32 = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev32[0].record(stream)
        k_fill32()
        ev32[1].record(stream)
        torch.cuda.synchronize()
        same32 = bool(((d_locs32[:total_locs].to(torch.int64) & 0xFFFFFFFF) == (d_locs[:total_locs] & 0xFFFFFFFF)).all().item())
        u32_block = {"value": N * K / el32, "unit": "reads/s (this rank)", "ms_per_step": el32 / K * 1e3, "k_locate_fill_ms": ev32[0].elapsed_time(ev32[1]),
                     "same_locations_as_the_64_bit_walk": same32,
                     "workload": "count+locate with the locations stored as 32 bits (rbg_locate_fill_dev32; informational)"}
        del d_locs32

    # ---- the same K count+locate steps as successive batches on two HIP streams (informational) --
    # K2 is bound by gather requests, the toehold-ordered K3 is not: the next batch's search overlaps
    # this batch's locate.  Outputs are double-buffered; the headline `value` stays the plain
    # one-stream figure above so that the per-kernel durations are undisturbed.
    el_pipe, same_out = 0.0, None
    if args.two_stream:
      alt = dict(lo=torch.empty_like(d_lo), hi=torch.empty_like(d_hi), k=torch.empty_like(d_k), loc_off=torch.empty_like(d_loc_off),
                 locs=torch.empty_like(d_locs), tmp=torch.empty_like(d_tmp), ws=torch.empty_like(d_ws))
      main = dict(lo=d_lo, hi=d_hi, k=d_k, loc_off=d_loc_off, locs=d_locs, tmp=d_tmp, ws=d_ws)
      streams2 = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]

      def step_on(buf, s):
          sp_ = s.cuda_stream
          chk(L.rbg_find_range_w_toehold_dev(rb.h, d_seqs.data_ptr(), d_off.data_ptr(), N, buf["lo"].data_ptr(), buf["hi"].data_ptr(),
                                             buf["k"].data_ptr(), sp_), "find_range_w_toehold")
          chk(L.rbg_locate_plan_dev(rb.h, buf["lo"].data_ptr(), buf["hi"].data_ptr(), N, max_hits, buf["loc_off"].data_ptr(),
                                    buf["tmp"].data_ptr(), tmp_bytes, sp_), "locate_plan")
          chk(L.rbg_locate_order_dev(rb.h, buf["k"].data_ptr(), N, buf["ws"].data_ptr(), ws_bytes, sp_), "locate_order")
          chk(L.rbg_locate_fill_dev(rb.h, buf["lo"].data_ptr(), buf["hi"].data_ptr(), buf["k"].data_ptr(), N, max_hits,
                                    buf["loc_off"].data_ptr(), buf["locs"].data_ptr(), buf["ws"].data_ptr(), sp_), "locate_fill")

      def pipelined():
          for s in range(K):
              step_on(main if s % 2 == 0 else alt, streams2[s % 2])

      pipelined()
      el_pipe = timed(pipelined, 1)
      same_out = bool((alt["locs"] == d_locs).all().item()) and bool((alt["loc_off"] == d_loc_off).all().item())
      del alt

    # ---- markers path (BASELINE.json configs[4], rb_align -m: rb_align.cpp:133-143), optional ------
    mk_block = None
    stage("capped / u32 / markers legs")
    if args.markers:
        rb.set_markers(*marker_arrays)
        d_mk_off = torch.empty(N + 1, dtype=torch.int64, device=dev)

        def k_mplan():
            chk(L.rbg_markers_plan_dev(rb.h, d_lo.data_ptr(), d_hi.data_ptr(), N, d_mk_off.data_ptr(), d_tmp.data_ptr(), tmp_bytes, st),
                "markers_plan")

        k_count()
        k_mplan()
        total_mk = int(d_mk_off[-1].item())
        d_mk = torch.empty(max(total_mk, 1), dtype=torch.int64, device=dev)

        def mstep():
            k_count()
            k_mplan()
            chk(L.rbg_markers_fill_dev(rb.h, d_lo.data_ptr(), d_hi.data_ptr(), N, d_mk_off.data_ptr(), d_mk.data_ptr(), st), "markers_fill")

        for _ in range(max(1, args.warmup)):
            mstep()
        el_mk = timed(mstep, K)
        n_with = int(((d_mk_off[1:] - d_mk_off[:-1]) > 0).sum().item())
        # the two marker kernels of the step apart (HIP events), and the step's bytes: K1's as run (kernels["k_find_range<count>"] below) + per read the
        # range (16) and its offset (8 + 8), two directory entries (8) and about two run starts / ends (16) per matched read, 16 per marker copied
        e_m = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        k_count()
        e_m[0].record(stream)
        k_mplan()
        e_m[1].record(stream)
        chk(L.rbg_markers_fill_dev(rb.h, d_lo.data_ptr(), d_hi.data_ptr(), N, d_mk_off.data_ptr(), d_mk.data_ptr(), st), "markers_fill")
        e_m[2].record(stream)
        torch.cuda.synchronize()
        ms_mplan, ms_mfill = e_m[0].elapsed_time(e_m[1]), e_m[1].elapsed_time(e_m[2])
        mk_bytes = 2 * (N * 32 + n_matched * 24) + 16 * total_mk
        mk_block = {"value": N * K / el_mk, "unit": "reads/s (this rank)", "ms_per_step": el_mk / K * 1e3,
                    "kernels_ms": {"markers_plan(k_markers_count+scan)": ms_mplan, "markers_fill": ms_mfill},
                    "roofline": {"bound": "hbm", "kernel": "k_find_range<count> (the step's dominant kernel: its roofline is kernels[k_find_range<count>] of this line); the two marker kernels:",
                                 "marker_kernels_alg_bytes": mk_bytes, "marker_kernels_ms": ms_mplan + ms_mfill,
                                 "achieved": mk_bytes / ((ms_mplan + ms_mfill) * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": mk_bytes / ((ms_mplan + ms_mfill) * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None},
                    "workload": "BASELINE.json configs[4] in rb_align -m form: find_range + markers_at(range), synthetic marker array "
                                "(w=10 rows before every variant site, both alleles)",
                    "marker_runs": int(len(marker_arrays[0])), "marker_values": int(len(marker_arrays[3])),
                    "markers_reported": total_mk, "reads_with_markers": n_with}

        # rb_markers' kernel (next-row f4): get_markers_greedy_seeding on every read and its reverse
        # complement (rb_markers.cpp:396-413), wsize 19 / max_range 1000 = rb_markers' defaults
        comp = torch.arange(256, dtype=torch.uint8, device=dev)
        for a_, b_ in zip(b"ACGT", b"TGCA"):
            comp[a_] = b_
        rc_reads = comp[reads.flip(1).long()]
        d_seqs2 = torch.cat([reads.reshape(-1), rc_reads.reshape(-1), torch.zeros(16, dtype=torch.uint8, device=dev)])
        del rc_reads
        d_off2 = (torch.arange(2 * N + 1, device=dev, dtype=torch.int64) * m)
        d_soff = torch.empty(2 * N + 1, dtype=torch.int64, device=dev)
        d_moff = torch.empty(2 * N + 1, dtype=torch.int64, device=dev)
        tmp2 = int(L.rbg_locate_plan_tmp_bytes(2 * N))
        d_tmp2 = torch.empty(tmp2, dtype=torch.uint8, device=dev)
        WS, MR = 19, 1000

        def k_splan():
            chk(L.rbg_marker_seeds_plan_dev(rb.h, d_seqs2.data_ptr(), d_off2.data_ptr(), 2 * N, WS, MR, 0, d_soff.data_ptr(),
                                            d_moff.data_ptr(), d_tmp2.data_ptr(), tmp2, st), "marker_seeds_plan")

        k_splan()
        n_seeds, n_smk = int(d_soff[-1].item()), int(d_moff[-1].item())
        d_srec = torch.empty(max(n_seeds, 1) * 6, dtype=torch.int64, device=dev)
        d_smk = torch.empty(max(n_smk, 1), dtype=torch.int64, device=dev)

        def sstep():
            k_splan()
            chk(L.rbg_marker_seeds_fill_dev(rb.h, d_seqs2.data_ptr(), d_off2.data_ptr(), 2 * N, WS, MR, 0, d_soff.data_ptr(),
                                            d_moff.data_ptr(), d_srec.data_ptr(), d_smk.data_ptr(), st), "marker_seeds_fill")

        sstep()
        el_sd = timed(sstep, K)
        # the same pair with the LOG between its phases (rbg_marker_seeds_plan_log_dev / _fill_log_dev: reads walked once)
        log_bytes = int(L.rbg_marker_seeds_log_bytes(rb.h, 2 * N, 0))
        d_log = torch.empty(log_bytes, dtype=torch.uint8, device=dev)
        ref_srec, ref_smk = d_srec.clone(), d_smk.clone()

        def sstep_log():
            chk(L.rbg_marker_seeds_plan_log_dev(rb.h, d_seqs2.data_ptr(), d_off2.data_ptr(), 2 * N, WS, MR, 0, d_soff.data_ptr(), d_moff.data_ptr(),
                                                d_tmp2.data_ptr(), tmp2, d_log.data_ptr(), log_bytes, st), "marker_seeds_plan_log")
            chk(L.rbg_marker_seeds_fill_log_dev(rb.h, d_seqs2.data_ptr(), d_off2.data_ptr(), 2 * N, WS, MR, 0, d_soff.data_ptr(), d_moff.data_ptr(),
                                                d_srec.data_ptr(), d_smk.data_ptr(), d_log.data_ptr(), log_bytes, st), "marker_seeds_fill_log")

        d_srec.fill_(-1); d_smk.fill_(-1)
        sstep_log()
        same_log = bool((d_srec == ref_srec).all().item()) and bool((d_smk == ref_smk).all().item())
        el_sl = timed(sstep_log, K)
        del ref_srec, ref_smk
        # greedy seeds (rbg_greedy_longest_seed_dev: get_seeds_greedy_w_sample reduced by locate_from_longest_seed, rowbowt.hpp:222-256,
        # :669-677) on the forward reads, min_length 20; outputs kept for the run-indexed row of space_speed
        d_g = [torch.empty(N, dtype=torch.int64, device=dev) for _ in range(5)]

        def gstep():
            chk(L.rbg_greedy_longest_seed_dev(rb.h, d_seqs.data_ptr(), d_off.data_ptr(), N, 20, *(t.data_ptr() for t in d_g), st), "greedy_longest_seed")

        gstep()
        el_g = timed(gstep, K)
        # ---- what the seeding kernels touched, and their rooflines (run-indexed layout: the instrumented instantiations, outside every timed region;
        # DESIGN.md 3).  Bytes as run: per sequence and walk its offsets (16) and 16 per read chunk; the plan's two counts (16); 64 per bucket
        # record, 8 per run-list entry scanned, 28 per narrowing round, the ftab entry, 4 / 6 per re-sample; per marker query 32 per bucket record (one or
        # two), behind an overflowing record 8 per run start / end and value offset read; the fill's seed records (48 each) and marker values (8 + 8 each).
        seed_roof = {}
        if int(ix.rank_layout) == 2:
            SD = ("steps", "slots", "dense", "searched_ranks", "ftab", "resamples", "read_chunks", "symbols", "marker_queries", "marker_dir", "marker_probes",
                  "marker_off", "marker_vals", "seed_recs", "sequences")
            Pm = int(ix.pos_bytes)
            ftab_bm, samp_bm = (16 if Pm == 4 else 32), (4 if Pm == 4 else 6)
            mkdir_bm = 4 if os.environ.get("RBG_MK_REC", "1")[:1] == "0" else 32     # a marker query reads 32-byte bucket records (round 6), or 4-byte directory entries
            li_m = rb.layout_info()
            rec_bm = 64 if any(int(x) for x in li_m.rec_bytes) else (8 if Pm == 4 else 16)
            d_sst = torch.zeros(16, dtype=torch.int64, device=dev)

            def seed_bytes(v, walks, counts_out):
                return (v["sequences"] * 16 + 16 * v["read_chunks"] + counts_out + rec_bm * v["slots"] + 8 * v["dense"] + 28 * v["searched_ranks"] + ftab_bm * v["ftab"]
                        + samp_bm * v["resamples"] + mkdir_bm * v["marker_dir"] + 8 * v["marker_probes"] + 8 * v["marker_off"] + 16 * v["marker_vals"] + 48 * v["seed_recs"])

            chk(L.rbg_marker_seeds_stats_dev(rb.h, d_seqs2.data_ptr(), d_off2.data_ptr(), 2 * N, WS, MR, d_soff.data_ptr(), d_moff.data_ptr(), d_tmp2.data_ptr(), tmp2,
                                             d_srec.data_ptr(), d_smk.data_ptr(), d_sst.data_ptr(), st), "marker_seeds_stats")
            torch.cuda.synchronize()
            same_inst = (int(d_soff[-1].item()), int(d_moff[-1].item())) == (n_seeds, n_smk)
            v_ms = dict(zip(SD, d_sst.cpu().numpy().tolist()[:15]))
            b_ms = seed_bytes(v_ms, 2, 2 * N * 16)
            t_ms = el_sd / K
            seed_roof["marker_seeds"] = {"bound": "hbm", "kernel": "k_marker_seeds_runs (count walk + fill walk)", "achieved": b_ms / t_ms / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": b_ms / t_ms / 1e9 / HBM_PEAK_GBS, "alg_bytes_per_step": b_ms, "ms": t_ms * 1e3, "touched": v_ms,
                                         "per_sequence_and_walk": {k_: v_ms[k_] / max(v_ms["sequences"], 1) for k_ in SD[:-1]},
                                         "sectors_per_sequence_and_walk": (v_ms["slots"] + (v_ms["dense"] + 7) // 8 + v_ms["searched_ranks"] + v_ms["ftab"] + v_ms["marker_dir"]
                                                                           + v_ms["marker_probes"] + v_ms["marker_off"]) / max(v_ms["sequences"], 1),
                                         "same_outputs_as_the_timed_instantiation": same_inst, "traffic": None}
            d_sst.zero_()
            chk(L.rbg_greedy_longest_seed_stats_dev(rb.h, d_seqs.data_ptr(), d_off.data_ptr(), N, 20, *(t.data_ptr() for t in d_g), d_sst.data_ptr(), st), "greedy_seed_stats")
            torch.cuda.synchronize()
            v_g = dict(zip(SD, d_sst.cpu().numpy().tolist()[:15]))
            b_g = seed_bytes(v_g, 1, N * 40)
            t_g = el_g / K
            seed_roof["greedy_seed"] = {"bound": "hbm", "kernel": "k_greedy_seed_runs", "achieved": b_g / t_g / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": b_g / t_g / 1e9 / HBM_PEAK_GBS, "alg_bytes_per_step": b_g, "ms": t_g * 1e3, "touched": v_g,
                                        "per_sequence": {k_: v_g[k_] / max(v_g["sequences"], 1) for k_ in SD[:-1]}, "traffic": None}
            # counter traffic where the stamped PMC file of this library holds the seeding kernels (tools/run_profiles_pmc.sh with --markers)
            try:
                import hashlib as _hl
                pjm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
                if pjm.get("_librbg_sha256") == _hl.sha256(open(os.path.join(ROOT, "rowbowt_amd", "librbg.so"), "rb").read()).hexdigest():
                    for key_, name_ in (("marker_seeds", "k_marker_seeds_runs (plan + fill)"), ("greedy_seed", "k_greedy_seed_runs")):
                        e_ = pjm.get(name_)
                        if isinstance(e_, dict) and e_.get("hbm_bytes_per_launch"):
                            seed_roof[key_]["traffic"] = e_["hbm_bytes_per_launch"]
                            seed_roof[key_]["tcc_miss_per_launch"] = e_.get("tcc_miss_per_launch")
            except Exception:
                pass
        slot_other = {"greedy_seed_ms": el_g / K * 1e3, "marker_seeds_ms": el_sd / K * 1e3, "marker_seeds_logged_ms": el_sl / K * 1e3}
        ref_g = [t.clone() for t in d_g]
        ref_seed_counts = (n_seeds, n_smk)
        mk_block["greedy_seed"] = {"value": N * K / el_g, "unit": "reads/s (this rank)", "ms_per_step": el_g / K * 1e3, "min_length": 20}
        mk_block["marker_seeds"] = {"value": N * K / el_sd, "unit": "reads/s (this rank; each read = both strands)",
                                    "ms_per_step": el_sd / K * 1e3, "wsize": WS, "max_range": MR,
                                    "seed_records": n_seeds, "markers_collected": n_smk,
                                    "two_walks_ms_per_step": el_sd / K * 1e3,
                                    "logged": {"value": N * K / el_sl, "ms_per_step": el_sl / K * 1e3, "log_bytes": log_bytes,
                                               "identical_to_the_two_walk_pair": same_log,
                                               "workload": "rbg_marker_seeds_plan_log_dev + _fill_log_dev: the count pass logs, the fill pass copies"},
                                    "workload": "rb_markers default mode: get_markers_greedy_seeding on read + reverse complement"}
        if seed_roof:
            mk_block["marker_seeds"]["roofline"] = seed_roof["marker_seeds"]
            mk_block["greedy_seed"]["roofline"] = seed_roof["greedy_seed"]

    # ---- what the kernels touched: one pass of the INSTRUMENTED instantiations on the same batch (outside every
    # timed region; same outputs).  The bytes of the algorithm as run follow from these counts.
    d_stats = torch.zeros(16, dtype=torch.int64, device=dev)

    def search_stats(toehold):   # (works on both layouts; on the run-indexed one the sums count that layout's accesses: include/rbg.h)
        d_stats.zero_()
        chk(L.rbg_find_range_stats_dev(rb.h, d_seqs.data_ptr(), d_off.data_ptr(), N, d_lo.data_ptr(), d_hi.data_ptr(),
                                       d_k.data_ptr() if toehold else None, d_stats.data_ptr(), st), "find_range_stats")
        torch.cuda.synchronize()
        v = d_stats.cpu().numpy().tolist()
        return dict(zip(("steps", "slots", "dense", "searched_ranks", "ftab", "resamples", "read_chunks", "symbols"), v[:8]))

    st_count = search_stats(False)
    st_toe = search_stats(True)
    k_plan()
    k_order()
    d_stats.zero_()
    chk(L.rbg_locate_fill_stats_dev(rb.h, d_lo.data_ptr(), d_hi.data_ptr(), d_k.data_ptr(), N, max_hits, d_loc_off.data_ptr(),
                                    d_locs.data_ptr(), d_ws.data_ptr(), d_stats.data_ptr(), st), "locate_fill_stats")
    torch.cuda.synchronize()
    st_loc = dict(zip(("phi_steps", "phi_searched", "chains", "locs"), d_stats.cpu().numpy().tolist()[:4]))
    rb.counters_reset()

    # max over ranks, counters over RCCL
    stage("all_reduce(MAX) of the times, all_reduce(SUM) of the counters")
    el_own = el
    t_el = torch.tensor([el, el_count, el_pipe], dtype=torch.float64, device=cdev)
    if use_dist:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
    el, el_count, el_pipe = float(t_el[0].item()), float(t_el[1].item()), float(t_el[2].item())
    g_counters = shard.reduce_counters(counters, device=cdev)  # the only collective of the run: 4 x u64 over RCCL
    # what a post-mortem of a multi-GPU run needs, per rank (outside the timed region; an all_gather of eight doubles)
    stage("all_gather of the per-rank diagnostics")
    mine = torch.tensor([rank, local_rank, t_load, t_wait, el_own / K * 1e3, ms_toe, ms_fill, float(ix.hbm_bytes)], dtype=torch.float64, device=cdev)
    if use_dist:
        rows_t = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(rows_t, mine)
    else:
        rows_t = [mine]
    per_rank = [{"rank": int(r_[0].item()), "device": int(r_[1].item()), "load_s": float(r_[2].item()), "wait_for_rank0_s": float(r_[3].item()),
                 "ms_per_step": float(r_[4].item()), "k2_ms": float(r_[5].item()), "k3_ms": float(r_[6].item()), "hbm_bytes": int(r_[7].item())} for r_ in rows_t]
    ranks_seen = len({r_["rank"] for r_ in per_rank})

    out = None
    if rank == 0:
        total_reads = N * world * K
        value = total_reads / el
        # ---- bytes of the algorithm AS RUN, from the instrumented pass: what each kernel has to
        # move for the steps it actually executed, at the sizes it is stored in HBM -- no sector padding, no re-reads.
        P = int(ix.pos_bytes)
        slot_b = int(ix.slot_bytes) or 16
        ftab_entry = 16 if P == 4 else 32

        runs_layout = int(ix.rank_layout) == 2
        li = rb.layout_info() if runs_layout else None
        rec_on = bool(li and any(li.rec_bytes))

        def search_bytes(sv, toehold):
            if runs_layout:
                # run-indexed layout, format 2 (rbg_runs2_device.hpp): "slots" = bucket records fetched (64 bytes each) or, without records,
                # directory gathers (two neighbouring entries: 8 or 16 bytes); "dense" = run-list entries the scans needed (8 bytes each);
                # "searched_ranks" = narrowing rounds (seven 4-byte pivots each); a re-sample = one gather of P
                per_slot = 64 if rec_on else (16 if P == 8 else 8)
                return (N * (16 + (24 if toehold else 16)) + 16 * sv["read_chunks"] + ftab_entry * sv["ftab"] + per_slot * sv["slots"]
                        + 8 * sv["dense"] + 28 * sv["searched_ranks"] + P * sv["resamples"])
            # per read: its two offsets (16) and its outputs (lo, hi [, toehold]); per 16-byte read chunk fetched;
            # per ftab entry; per 16-byte rank slot; per 2-byte dense-table row; per rank searched in a run list:
            # two ord entries + ~3 probes of a {start, cum} pair; per materialised re-sample: ord (4) + sample (P)
            return (N * (16 + (24 if toehold else 16)) + 16 * sv["read_chunks"] + ftab_entry * sv["ftab"] + 16 * sv["slots"]
                    + 2 * sv["dense"] + (8 + 3 * 2 * P) * sv["searched_ranks"] + (4 + P) * sv["resamples"])

        # K3, ordered walk: per read the sorted toehold (8), its permutation entry (4) and two loc_off entries (16);
        # per phi step one PhiSlot (4 x P); per searched step two ord entries + ~3 probes of a PhiEnt; 8 per location stored
        loc_bytes = N * 28 + 4 * P * st_loc["phi_steps"] + (8 + 3 * 2 * P) * st_loc["phi_searched"] + 8 * st_loc["locs"]
        ref_toe = (57 * m + 24) * N      # the reference's one-symbol-per-step algorithm, for comparison only
        ref_fill = 24 * total_locs
        ref_count = (49 * m + 16) * N
        kernels = {
            "k_find_range<toehold>": {"ms": ms_toe, "alg_bytes": search_bytes(st_toe, True), "ref_alg_bytes": ref_toe, "touched": st_toe},
            "k_locate_fill": {"ms": ms_fill, "alg_bytes": loc_bytes, "ref_alg_bytes": ref_fill, "touched": st_loc},
            "locate_plan(k_occ+scan)": {"ms": ms_plan, "alg_bytes": 24 * N, "ref_alg_bytes": 24 * N},
            "locate_order(radix sort of toeholds)": {"ms": ms_order, "alg_bytes": 24 * N, "ref_alg_bytes": 24 * N},
            "k_find_range<count>": {"ms": ms_count, "alg_bytes": search_bytes(st_count, False), "ref_alg_bytes": ref_count, "touched": st_count},
        }
        dom = max(("k_find_range<toehold>", "k_locate_fill"), key=lambda k: kernels[k]["ms"])
        dom_s = kernels[dom]["ms"] * 1e-3
        ach = kernels[dom]["alg_bytes"] / dom_s / 1e9
        # PMC traffic: only the committed passes taken with THIS library build on THIS workload count
        # (profiles/pmc_traffic.json carries the sha256 of the librbg.so that was profiled)
        import hashlib
        so_hash = hashlib.sha256(open(os.path.join(ROOT, "rowbowt_amd", "librbg.so"), "rb").read()).hexdigest()
        traffic = misses = None
        pmc_by_kernel = {}   # kernels' counter traffic (HBM bytes, L2 misses per launch) where the stamped PMC file holds them
        pmc_note = "no PMC passes committed for this build"
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        default_workload = (args.L, args.H, args.reads, args.read_len, args.site_rate) == (40_000_000, 50, 10_000_000, 100, 0.01)
        if os.path.exists(pmc) and default_workload:
            try:
                pj = json.load(open(pmc))
                if pj.get("_librbg_sha256") == so_hash:
                    pkey = dom + (f" [runs, pos_bytes {P}]" if runs_layout else "")
                    traffic = pj.get(pkey, {}).get("hbm_bytes_per_launch")
                    misses = pj.get(pkey, {}).get("tcc_miss_per_launch")
                    pmc_note = pj.get("_source")
                    sfx = f" [runs, pos_bytes {P}]" if runs_layout else ""
                    for kk_ in kernels:
                        ent_ = pj.get(kk_ + sfx) if kk_.startswith("k_find_range") else pj.get(kk_)
                        if isinstance(ent_, dict) and ent_.get("hbm_bytes_per_launch"):
                            pmc_by_kernel[kk_] = ent_
                else:
                    pmc_note = "profiles/pmc_traffic.json was taken with a different librbg.so build: dropped"
            except Exception:
                traffic = misses = None
        # request ceiling: tools/gather_ceiling.hip's sweep (committed result; measured on an MI355X of this pool)
        ceiling = None
        # what an L2-served scattered request costs beside a miss (tools/gather_width.hip: one dependent miss + H gathers from a 512 KB table per step)
        hit_cost_s = None
        try:
            gw = json.load(open(os.path.join(ROOT, "profiles", "r06_gather_width.json")))["rows"]
            mix_rows = {}
            for r_ in gw:
                h_ = [v for k_, v in r_.items() if k_.startswith("bytes_per_access")][0]
                if h_ <= 0 and r_["waves_per_simd"] == 4:
                    mix_rows[-h_] = r_["G_accesses_per_s"] * 1e9
            if 0 in mix_rows and 2 in mix_rows:
                hit_cost_s = (1.0 / mix_rows[2] - 1.0 / mix_rows[0]) / 2.0
        except Exception:
            pass
        try:
            ceiling = json.load(open(os.path.join(ROOT, "profiles", "gather_ceiling.json")))["peak_G_gathers_per_s"]
        except Exception:
            pass
        # dependent random gathers the search kernel issues, from its own counts (independent of the PMC file): rank
        # slots, dense-table rows, ftab entries, the two gathers of a re-sample, run-list probes.  The read's own
        # bytes and the outputs are NOT in it: they are sequential sectors (2-3 per read), not index gathers --
        # TCC_MISS of the same launch = these gathers + those sectors (273M = 231M + 42M on the default workload).
        gathers = None
        if dom == "k_find_range<toehold>" and runs_layout:
            # records / directory gathers, one sector per eight scanned entries, one per narrowing round, ftab entries, re-samples
            gathers = st_toe["slots"] + (st_toe["dense"] + 7) // 8 + st_toe["searched_ranks"] + st_toe["ftab"] + st_toe["resamples"]
        elif dom == "k_find_range<toehold>":
            gathers = st_toe["slots"] + st_toe["dense"] + st_toe["ftab"] + 2 * st_toe["resamples"] + 4 * st_toe["searched_ranks"]
        out = {
            "metric": f"reads/s ({args.read_len} bp, count+locate)",
            "value": value,
            "unit": "reads/s",
            "n_gpus": world,
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": el / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64" if ix.pos_bytes == 8 else "u32/u64",
            "data": "synthetic",
            "config": {
                "workload": f"count+locate (find_range_w_toehold + locs_at, max_hits={'2^64-1' if max_hits == MAXU else max_hits}) on a "
                            f"synthetic chr22-scale pangenome r-index, {N} x {m} bp reads per GPU per step (BASELINE.json configs[2])",
                "index": {"L": args.L, "H": args.H, "n": int(inp["n"]), "r": int(inp["r"]), "site_rate": args.site_rate,
                          "hbm_bytes": int(ix.hbm_bytes), "pos_bytes": int(ix.pos_bytes), "seed": args.seed,
                          "slot_bytes": slot_b, "symbols_per_gather": int(ix.kmer_steps), "symbols_per_gather_requested": int(ix.kmer_steps_requested),
                          "hbm_free_at_load": int(ix.hbm_free_at_load), "hbm_budget": int(ix.hbm_budget),
                          "rank_layout": "runs" if int(ix.rank_layout) == 2 else "slots",
                          "hbm_budget_source": ("bench.py --hbm-budget-gb -1: three quarters of the free HBM" if args.hbm_budget_gb < 0
                                                else "the library's default (a quarter of the free HBM)" if args.hbm_budget_gb == 0 else f"--hbm-budget-gb {args.hbm_budget_gb}"), "ftab_k": int(ix.ftab_k), "depth_runs": [int(x) for x in ix.depth_runs],
                          # what the load's budget rules decided (rbg_layout_info; tools/layout_rules_table.py makes DESIGN.md 2c's table from these)
                          **({"layout_info": {"budget_raised": int(li.budget_raised), "depths_kept": [d_ + 1 for d_ in range(8) if int(li.depth_mask_kept) >> d_ & 1],
                                              "depths_with_records": [d_ + 1 for d_ in range(8) if int(li.rec_bytes[d_])], "depths_dropped_budget": int(li.depths_dropped_budget),
                                              "entries": [int(x) for x in li.entries], "rec_bytes": [int(x) for x in li.rec_bytes], "rec_overflow": [int(x) for x in li.rec_overflow],
                                              "phi_slots": int(li.phi_slots), "phi_slot_bytes": int(li.phi_slot_bytes), "phi_entries": int(li.phi_entries),
                                              "rank_directories": int(li.rank_directories), "phi_directory": int(li.phi_directory)}} if li else {})},
                "reads_per_gpu": N, "read_len": m, "substituted_fraction": 0.1,
                "documents": info["H"] if args.docs == "on" else 0,
                "chain_order": ("by locus: offset inside the document, then document"
                                if args.docs == "on" and os.environ.get("RBG_LOCATE_ORDER") != "abs" and (info["H"] >= 128 or os.environ.get("RBG_LOCATE_ORDER") == "locus")
                                else "by absolute text position"),
                "parallelism": f"index replicated x{world}, reads sharded, no data-path collective",
                **({"rehearsal": f"NOT A MEASUREMENT: {world} ranks on {torch.cuda.device_count()} GPU(s), collectives over gloo (--rehearse-ranks)"}
                   if args.rehearse_ranks else {}),
            },
            "two_stream_pipeline": ({"value": N * world * K / el_pipe, "unit": "reads/s", "ms_per_step": el_pipe / K * 1e3,
                                     "identical_output": same_out,
                                     "workload": "the same K count+locate steps issued as successive batches on two HIP streams"}
                                    if args.two_stream else None),
            "markers": mk_block,
            "locations_u32": u32_block,
            "capped_max_hits_1": {"value": N * K / el_cap, "unit": "reads/s (this rank)", "ms_per_step": el_cap / K * 1e3,
                                  "first_location_is_the_toehold": capped_ok,
                                  "workload": "count+locate with max_hits = 1 (diagnostic: search cost without the phi walks)"},
            "count_only": {"value": N * world * K / el_count, "unit": "reads/s", "ms_per_step": el_count / K * 1e3,
                           "workload": "BASELINE.json configs[1]: find_range only"},
            "counters": {"reads": g_counters[0], "matched": g_counters[1], "sum_occ": g_counters[2], "sum_locs": g_counters[3],
                         "reduced_over": (f"{'gloo (rehearsal)' if args.rehearse_ranks else 'RCCL'} all_reduce over {world} rank(s)" if use_dist else "single GPU (no process group)")},
            # the multi-GPU run's post-mortem block: who ran where, how long each rank loaded / waited / stepped, what the one-per-node cache cost
            "device_state_during_the_timed_steps": device_state,
            "per_rank": per_rank,
            "rccl_ranks_seen": ranks_seen,
            "collective_backend": ("gloo (rehearsal)" if args.rehearse_ranks else "nccl (RCCL)") if use_dist else None,
            "cache_write_s": t_cache_write if cache_path else None, "cache_bytes": cache_bytes if cache_path else None,
            "cache_path": cache_path, "suffix_array_s": t_sa,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                         "alg_bytes_per_launch": kernels[dom]["alg_bytes"], "kernel_ms": kernels[dom]["ms"],
                         # sectors the fabric moved (PMC) against the same peak, and how much of that was padding
                         "sector_frac": (traffic / dom_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "padding_ratio": (traffic / kernels[dom]["alg_bytes"]) if traffic else None,
                         # what holds a kernel of dependent 16-byte gathers: requests/s against the ceiling tools/gather_ceiling.hip
                         # measured (profiles/gather_ceiling.json); `gathers` = the kernel's own count of loads that can miss
                         "request_roof": ({"unit": "G random 16-byte gathers/s", "gathers_per_launch": gathers, "achieved": gathers / dom_s / 1e9,
                                           "peak": ceiling, "frac": gathers / dom_s / 1e9 / ceiling,
                                           "tcc_miss_per_launch": misses,
                                           "layout": "runs (records / directory gathers + scanned sectors + rounds + ftab + re-samples)" if runs_layout else "slots",
                                           "note": "peak = the most random 16-byte gathers per second tools/gather_ceiling.hip gets out of this chip "
                                                   "(flat from 1 to 16 loads in flight per lane and 2 to 8 waves per SIMD: a throughput limit of the "
                                                   "memory system, not latency); achieved = the kernel's dependent index gathers only, its sequential "
                                                   "read/output sectors excluded"} if ceiling and gathers else None),
                         "reference_byte_model": {"bytes_per_launch": kernels[dom]["ref_alg_bytes"],
                                                  "GBps": kernels[dom]["ref_alg_bytes"] / dom_s / 1e9,
                                                  "note": "SURVEY 8d's bytes of the reference's one-symbol-per-step algorithm over this kernel's time: a "
                                                          "speed statement (how fast that algorithm would have to stream), not a roofline"},
                         "pmc": pmc_note, "librbg_sha256": so_hash,
                         "note": "achieved = bytes of the algorithm as run (counted by the instrumented instantiation of the same kernel on the "
                                 "same batch: executed steps x slots touched x 16 + read chunks + ftab entries + re-sample gathers + offsets + "
                                 "outputs) / mean kernel duration (HIP events on the launch stream)"},
            # per kernel: `frac_of_hbm_peak` prices the bytes of the algorithm AS RUN (every executed step's bytes, whether a cache or HBM served them:
            # where neighbouring chains share sectors -- K3 in toehold order -- it EXCEEDS what HBM moved); `traffic_*` are the counters' HBM bytes
            # (FETCH_SIZE + WRITE_SIZE of the stamped PMC file) over the same duration: the HBM figure
            "kernels": {k: {"ms": v["ms"], "alg_bytes": v["alg_bytes"], "alg_GBps": v["alg_bytes"] / (v["ms"] * 1e-3) / 1e9,
                            "frac_of_hbm_peak": v["alg_bytes"] / (v["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "frac_of_hbm_peak_is": "bytes of the algorithm as run, cache-served bytes included",
                            **({"traffic_bytes": pmc_by_kernel[k]["hbm_bytes_per_launch"], "tcc_miss_per_launch": pmc_by_kernel[k].get("tcc_miss_per_launch"),
                                "traffic_frac_of_hbm_peak": pmc_by_kernel[k]["hbm_bytes_per_launch"] / (v["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "alg_bytes_over_traffic": v["alg_bytes"] / pmc_by_kernel[k]["hbm_bytes_per_launch"]} if k in pmc_by_kernel else {}),
                            # the request model: misses at the gather ceiling + L2-served requests at what tools/gather_width.hip measured for one
                            **({"ceiling_model": {"miss_ms": pmc_by_kernel[k]["tcc_miss_per_launch"] / (ceiling * 1e9) * 1e3,
                                                  "l2_hit_ms": pmc_by_kernel[k]["tcc_hit_per_launch"] * hit_cost_s * 1e3,
                                                  "model_ms": (pmc_by_kernel[k]["tcc_miss_per_launch"] / (ceiling * 1e9) + pmc_by_kernel[k]["tcc_hit_per_launch"] * hit_cost_s) * 1e3,
                                                  "measured_over_model": v["ms"] / ((pmc_by_kernel[k]["tcc_miss_per_launch"] / (ceiling * 1e9) + pmc_by_kernel[k]["tcc_hit_per_launch"] * hit_cost_s) * 1e3),
                                                  "note": "L2 misses / the random-gather ceiling (profiles/gather_ceiling.json) + L2 hits x the cost of an L2-served scattered "
                                                          "request beside a miss (profiles/r06_gather_width.json, H = 2 row): 1.0 = the kernel runs at what its requests cost"}}
                               if k in pmc_by_kernel and ceiling and hit_cost_s and pmc_by_kernel[k].get("tcc_miss_per_launch") and pmc_by_kernel[k].get("tcc_hit_per_launch") else {}),
                            "touched": v.get("touched")}
                        for k, v in kernels.items()},
            "per_read": {"lf_gathers": st_toe["steps"] / N, "slots": st_toe["slots"] / N, "symbols_consumed": st_toe["symbols"] / N,
                         "read_chunks": st_toe["read_chunks"] / N, "resamples": st_toe["resamples"] / N,
                         "phi_steps": st_loc["phi_steps"] / N},
        }

    # ---- full-size property check (outside the timed region): every location reported for the
    # first --property-reads reads really is an occurrence, locations of a read are distinct and
    # their number equals hi-lo+1; unmatched reads report {1,0}
    if rank == 0 and args.property_reads > 0:
        stage("full-size properties (rank 0)")
        step()
        torch.cuda.synchronize()
        npr = min(args.property_reads, N)
        offs = d_loc_off[:npr + 1]
        nloc = int(offs[-1].item())
        occ_t = offs[1:] - offs[:-1]
        lo_t, hi_t = d_lo[:npr], d_hi[:npr]
        ok_occ = bool((torch.where(hi_t >= lo_t, hi_t - lo_t + 1, torch.zeros_like(lo_t)) == occ_t).all().item())
        ok_empty = bool(((hi_t >= lo_t) | ((lo_t == 1) & (hi_t == 0))).all().item())
        ridx = torch.repeat_interleave(torch.arange(npr, device=dev), occ_t)
        locs_t = d_locs[:nloc]
        bad = torch.zeros(nloc, dtype=torch.bool, device=dev)
        for j in range(m):
            bad |= text[locs_t + j] != reads[ridx, j]
        ok_match = not bool(bad.any().item())
        key = ridx * (int(inp["n"]) + 1) + locs_t
        ok_distinct = int(torch.unique(key).numel()) == nloc
        out["properties_full_size"] = {"reads": npr, "locations": nloc, "every_location_is_an_occurrence": ok_match,
                                       "locations_distinct": ok_distinct, "occ_equals_range_width": ok_occ,
                                       "empty_is_{1,0}": ok_empty}
        del ridx, locs_t, bad, key
        if not (ok_match and ok_distinct and ok_occ and ok_empty):
            print(json.dumps(out))
            raise SystemExit("PROPERTY FAILURE at full size")
    text_for_replicas = text if args.replicas else None   # (the --replicas leg samples each replica's batch from it; that leg skips the space_speed rebuilds, so the
    #                                                          text is not resident beside a 221 GB replica)
    del text
    torch.cuda.empty_cache()

    # ---- parity sample + CPU baseline (rank 0, N=1 only; never part of the timed region) ------
    if rank == 0 and not (args.no_cpu_baseline and args.check_reads == 0):
        stage("oracle parity sample + cpu_baseline (rank 0)")
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import orc  # oracle: checker + cpu_baseline only

        t0 = time.time()
        o = orc.Oracle.from_runs(inp["heads"], inp["lens"], inp["ssa"], inp["esa"])
        log(rank, f"oracle built in {time.time() - t0:.1f}s")
        nchk = min(args.check_reads, N)
        ncopy = min(N, max(nchk, 1 if args.no_cpu_baseline else 50000))
        h_seqs = reads[:ncopy].cpu().numpy().reshape(-1)
        if nchk:
            h_off = (np.arange(nchk + 1, dtype=np.uint64) * m)
            step()
            torch.cuda.synchronize()
            g_lo = d_lo[:nchk].cpu().numpy().view(np.uint64)
            g_hi = d_hi[:nchk].cpu().numpy().view(np.uint64)
            g_k = d_k[:nchk].cpu().numpy().view(np.uint64)
            g_off = d_loc_off[:nchk + 1].cpu().numpy().view(np.uint64)
            g_locs = d_locs[:int(g_off[-1])].cpu().numpy().view(np.uint64)
            ncpu = min(os.cpu_count() or 1, 64)
            wlo, whi, wk = o.find_range_w_toehold_batch(h_seqs, h_off, nthreads=ncpu)
            woff, wlocs = o.locs_at_batch(wlo, whi, wk, max_hits, nthreads=ncpu)
            ok = bool((g_lo == wlo).all() and (g_hi == whi).all() and (g_k == wk).all() and (g_off == woff).all()
                      and (g_locs == wlocs).all())
            # K1 (rbg_find_range_dev, the count-only kernel) against the oracle's find_range on the same reads
            k_count()
            torch.cuda.synchronize()
            c_lo = d_lo[:nchk].cpu().numpy().view(np.uint64)
            c_hi = d_hi[:nchk].cpu().numpy().view(np.uint64)
            olo, ohi = o.find_range_batch(h_seqs, h_off, nthreads=ncpu)
            ok1 = bool((c_lo == olo).all() and (c_hi == ohi).all())
            out["parity"] = {"reads_checked": nchk, "locs_checked": int(woff[-1]), "bit_exact_vs_oracle": ok and ok1,
                             "count_toehold_locate_bit_exact": ok, "count_only_kernel_bit_exact": ok1}
            if not (ok and ok1):
                print(json.dumps(out))
                raise SystemExit("PARITY FAILURE: HIP path disagrees with the oracle")
        if args.markers and nchk:
            o.set_markers(*marker_arrays)
            mstep()
            torch.cuda.synchronize()
            m_lo = d_lo[:nchk].cpu().numpy().view(np.uint64)
            m_hi = d_hi[:nchk].cpu().numpy().view(np.uint64)
            m_off = d_mk_off[:nchk + 1].cpu().numpy().view(np.uint64)
            m_val = d_mk[:int(m_off[-1])].cpu().numpy().view(np.uint64)
            okm = True
            for i in range(nchk):
                if m_val[int(m_off[i]):int(m_off[i + 1])].tolist() != o.markers_at(int(m_lo[i]), int(m_hi[i])):
                    okm = False
                    break
            out["markers"]["parity"] = {"reads_checked": nchk, "markers_checked": int(m_off[-1]), "bit_exact_vs_oracle": okm}
            # marker seeds: first ncs forward reads and first ncs reverse complements against the oracle
            ncs = min(nchk, 4000)
            sstep()
            torch.cuda.synchronize()
            oks, n_rec = True, 0
            for base in (0, N):
                so = d_soff[base:base + ncs + 1].cpu().numpy().view(np.uint64)
                rec = d_srec[6 * int(so[0]):6 * int(so[-1])].cpu().numpy().view(np.uint64).reshape(-1, 6)
                mlo, mhi = (int(rec[0, 4]), int(rec[-1, 5])) if len(rec) else (0, 0)
                mkv = d_smk[mlo:mhi].cpu().numpy().view(np.uint64)
                sq = d_seqs2[base * m:(base + ncs) * m].cpu().numpy().reshape(ncs, m)
                for i in range(ncs):
                    want = o.markers_greedy_seeding(sq[i].tobytes(), WS, MR)
                    got = rec[int(so[i] - so[0]):int(so[i + 1] - so[0])]
                    if len(got) != len(want):
                        oks = False
                        break
                    for g, (wl, wh, wqs, wqe, wm) in zip(got, want):
                        if (int(g[0]), int(g[1]), int(g[2]), int(g[3])) != (wl, wh, wqs, wqe) or mkv[int(g[4]) - mlo:int(g[5]) - mlo].tolist() != wm:
                            oks = False
                    n_rec += len(want)
                    if not oks:
                        break
            out["markers"]["marker_seeds"]["parity"] = {"sequences_checked": 2 * ncs, "seed_records_checked": n_rec, "bit_exact_vs_oracle": oks}
            if not oks:
                print(json.dumps(out))
                raise SystemExit("PARITY FAILURE (marker seeds): HIP path disagrees with the oracle")
            if not okm:
                print(json.dumps(out))
                raise SystemExit("PARITY FAILURE (markers): HIP path disagrees with the oracle")
        if world == 1 and not args.no_cpu_baseline:
            # bounded sample, single thread like rb_align's serial loop (rb_align.cpp:176-178)
            probe = min(1000, N)
            p_off = (np.arange(probe + 1, dtype=np.uint64) * m)
            per_read = 0.0
            for _warm in range(2):   # the second pass has the index's pages and caches warm, like the sample after it
                t0 = time.perf_counter()
                plo, phi, pk = o.find_range_w_toehold_batch(h_seqs[:probe * m], p_off, nthreads=1)
                o.locs_at_batch(plo, phi, pk, max_hits, nthreads=1)
                per_read = (time.perf_counter() - t0) / probe
            ns = int(max(probe, min(ncopy, args.cpu_seconds / max(per_read, 1e-9))))
            s_off = (np.arange(ns + 1, dtype=np.uint64) * m)
            t0 = time.perf_counter()
            slo, shi, sk = o.find_range_w_toehold_batch(h_seqs[:ns * m], s_off, nthreads=1)
            o.locs_at_batch(slo, shi, sk, max_hits, nthreads=1)
            dt = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": ns / dt, "unit": "reads/s", "cores": 1, "kind": "port",
                                   "sample": f"first {ns} reads of the same batch, count+locate, oracle/rb_oracle.c "
                                             f"(CPU restatement of the reference algorithm), {dt:.1f}s",
                                   "host": f"{os.cpu_count()} logical CPUs, CPU quota {cpu_budget()}"}
            # all the cores this process may use (the container's CPU quota counts: cgroup cpu.max), for scale only
            ncpu = cpu_budget()
            na = min(ncopy, ns * min(ncpu, 64))
            a_off = (np.arange(na + 1, dtype=np.uint64) * m)
            t0 = time.perf_counter()
            alo, ahi, ak = o.find_range_w_toehold_batch(h_seqs[:na * m], a_off, nthreads=ncpu)
            o.locs_at_batch(alo, ahi, ak, max_hits, nthreads=ncpu)
            dta = time.perf_counter() - t0
            out["cpu_baseline_all_cores"] = {"value": na / dta, "unit": "reads/s", "cores": ncpu, "kind": "port",
                                             "sample": f"first {na} reads, OpenMP over reads, {dta:.1f}s"}
            # SURVEY 8d's optional mode: the same port answering rank / select / access from Elias-Fano vectors and a
            # Huffman-shaped wavelet tree (what sdsl holds for the reference) instead of decoded arrays -- the reference's
            # memory behaviour; a third of the sample, one thread; results must be the plain mode's
            o.set_reference_shaped(True)
            nr_ = max(probe, ns // 3)
            r_off = (np.arange(nr_ + 1, dtype=np.uint64) * m)
            t0 = time.perf_counter()
            rlo, rhi, rk_ = o.find_range_w_toehold_batch(h_seqs[:nr_ * m], r_off, nthreads=1)
            _r_off, r_locs = o.locs_at_batch(rlo, rhi, rk_, max_hits, nthreads=1)
            dtr = time.perf_counter() - t0
            o.set_reference_shaped(False)
            same_r = bool((rlo == slo[:nr_]).all() and (rhi == shi[:nr_]).all() and (rk_ == sk[:nr_]).all())
            out["cpu_baseline"]["reference_shaped"] = {"value": nr_ / dtr, "unit": "reads/s", "cores": 1,
                                                       "sample": f"first {nr_} reads, sd_vector-like Elias-Fano + wt_huff-like wavelet tree in the port (orc_set_reference_shaped), {dtr:.1f}s",
                                                       "same_results_as_the_arrays": same_r}
            if not same_r:
                print(json.dumps(out))
                raise SystemExit("oracle: reference-shaped mode disagrees with the decoded arrays")
        o.close()

    # ---- space for speed (rank 0, N=1): the same two search kernels with the replica rebuilt at fewer symbols
    # per gather.  The headline ran at the deepest level that fitted; these rows say what the
    # same call costs on a device with less free HBM.
    if rank == 0 and world == 1 and not args.no_space_speed:
        stage("space_speed block")
        from rowbowt_amd import capi

        def time_search():
            res = {}
            for name, fn in (("k_find_range<count>", k_count), ("k_find_range<toehold>", k_toehold)):
                fn()
                e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                e[0].record(stream)
                for t in range(3):
                    fn()
                    e[t + 1].record(stream)
                torch.cuda.synchronize()
                res[name] = min(e[t].elapsed_time(e[t + 1]) for t in range(3))
            return res

        def time_locate(ms):
            """K3 of the replica in `rb` on this batch (after its own K2, plan and order), and the end-to-end rate of the four launches"""
            k_toehold(); k_plan(); k_order()
            best = None
            for _ in range(3):
                e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                e[0].record(stream)
                k_fill()
                e[1].record(stream)
                torch.cuda.synchronize()
                t = e[0].elapsed_time(e[1])
                best = t if best is None else min(best, t)
            ms["k_locate_fill"] = best
            return N / ((ms["k_find_range<toehold>"] + best + ms_plan + ms_order) * 1e-3)

        headline_runs = int(ix.rank_layout) == capi.LAYOUT_RUNS
        budget34_mb = int(free_at_start * 3 // 4) >> 20   # the slot rows are built under three quarters of the free HBM whatever the headline's budget

        def describe(rb_):
            """what a replica is: layout, symbols per step, bytes -- and, run-indexed, its depth set and rank / phi structures"""
            i_, d_ = rb_.info(), {}
            if int(i_.rank_layout) == capi.LAYOUT_RUNS:
                li_ = rb_.layout_info()
                d_ = {"depths": [x + 1 for x in range(8) if int(li_.depth_mask_kept) >> x & 1],
                      "ranks": "bucket records" if sum(int(x) for x in li_.rec_bytes) > 0 else "directories + run lists",
                      "phi": "slots of about n / r rows" if int(li_.phi_slots) else "list of sampled positions + directory"}
            return {"layout": "runs" if int(i_.rank_layout) == capi.LAYOUT_RUNS else "slots", **d_, "symbols_per_gather": int(i_.kmer_steps), "hbm_bytes": int(i_.hbm_bytes)}

        ms0 = time_search()
        rows = [{**describe(rb), "headline_replica": True, "ms": ms0, "count_locate_reads_per_s": time_locate(ms0)}]
        top = int(ix.kmer_steps)
        slot_other_row = None
        # reference outputs of the whole batch from the slot tables (oracle-checked above on a sample), for the run-indexed row
        step()
        torch.cuda.synchronize()
        ref_out = [t.clone() for t in (d_lo, d_hi, d_k, d_loc_off)]
        ref_locs = d_locs[:total_locs].clone()
        for lvl in range(5 if headline_runs else top - 1, 0, -1):
            rb.close()
            torch.cuda.empty_cache()
            with capi.default_option(capi.OPT_KMER_STEPS, lvl), capi.default_option(capi.OPT_RANK_LAYOUT, capi.LAYOUT_SLOTS), \
                    capi.default_option(capi.OPT_HBM_BUDGET_MB, max(budget_mb, budget34_mb)):   # the caller's own settings are put back afterwards
                rb = ra.RowBowt.from_runs(inp["heads"], inp["lens"], inp["ssa"], inp["esa"], device=local_rank)
            ms_l = time_search()
            rate_l = time_locate(ms_l)
            step()
            same_l = all(bool((a == b).all().item()) for a, b in zip(ref_out, (d_lo, d_hi, d_k, d_loc_off))) and bool((ref_locs == d_locs[:total_locs]).all().item())
            rows.append({**describe(rb), "ms": ms_l, "identical_to_the_headline_replica_on_the_whole_batch": same_l, "count_locate_reads_per_s": rate_l})
            if not same_l:
                out["space_speed"] = {"rows": rows}
                print(json.dumps(out))
                raise SystemExit("PARITY FAILURE: a slot-table replica disagrees with the headline replica")
            if headline_runs and args.markers and slot_other_row is None:   # the kernels beside the rb_align path on slot tables (five symbols), same batch
                rb.set_markers(*marker_arrays)
                gstep(); sstep(); sstep_log()
                slot_other_row = {}
                for name, fn in (("greedy_seed_ms", gstep), ("marker_seeds_ms", sstep), ("marker_seeds_logged_ms", sstep_log)):
                    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                    e[0].record(stream); fn(); e[1].record(stream); fn(); e[2].record(stream)
                    torch.cuda.synchronize()
                    slot_other_row[name] = min(e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]))
                rows[-1]["other_kernels"] = {"slot_layout": slot_other_row}
        # the run-indexed layout (space proportional to r; wave-cooperative predecessor search): same batch, same outputs
        rb.close()
        torch.cuda.empty_cache()
        # (run lists for all eight k-mer depths in this row; the rows after it leave depths out: RBG_OPT_RUN_DEPTHS, 0x8B --
        #  depths 1, 2, 4, 8 -- is the library's default)
        with capi.default_option(capi.OPT_RANK_LAYOUT, capi.LAYOUT_RUNS), capi.default_option(capi.OPT_RUN_DEPTHS, 0xFF):
            rb = ra.RowBowt.from_runs(inp["heads"], inp["lens"], inp["ssa"], inp["esa"], device=local_rank)
        ms_r = time_search()
        step()
        rate_r = time_locate(ms_r)
        run_phi_slots = int(rb.layout_info().phi_slots)   # > 0: phi through slots of about n / r rows (RBG_OPT_RUN_PHI; the library's choice here)
        # bytes of the run-indexed kernels AS RUN (instrumented instantiations; sums as include/rbg.h lists them for this
        # layout): per read its offsets + outputs + read chunks + ftab entry; per directory gather 8; per run-list entry a
        # probe needed 2P; per narrowing round 16 pivot keys of P; per materialised re-sample P.  K3: per read 28, per phi
        # step one 8-byte directory gather, per sampled position a probe needed 2P, 8 per location stored.
        Pr = int(rb.info().pos_bytes)
        rs_count, rs_toe = search_stats(False), search_stats(True)
        k_toehold(); k_plan(); k_order()
        d_stats.zero_()
        chk(L.rbg_locate_fill_stats_dev(rb.h, d_lo.data_ptr(), d_hi.data_ptr(), d_k.data_ptr(), N, max_hits, d_loc_off.data_ptr(),
                                        d_locs.data_ptr(), d_ws.data_ptr(), d_stats.data_ptr(), st), "locate_fill_stats")
        torch.cuda.synchronize()
        rs_loc = dict(zip(("phi_steps", "phi_searched" if run_phi_slots else "probe_entries", "chains", "locs"), d_stats.cpu().numpy().tolist()[:4]))
        rb.counters_reset()

        # format 2 (rbg_runs2_device.hpp; the default since round 4): a directory gather is two neighbouring entries of 4 bytes (8 at 8-byte
        # positions: count + the rank's high part), a run-list entry 8 bytes at either width, a narrowing round seven 4-byte pivots, a sample 4 / 6
        # bytes; a phi step reads two 4-byte counts (+ one 8-byte super count at 8-byte positions) and entries of 8 / 12 bytes.
        run_recs = sum(int(x) for x in rb.layout_info().rec_bytes) > 0   # bucket records (RBG_OPT_RUN_REC): a "directory gather" is one 64-byte record
        b_dir, b_ent, b_narrow, b_samp = (64 if run_recs else 8 if Pr == 4 else 16), 8, 28, (4 if Pr == 4 else 6)
        b_phi_dir, b_phi_ent = (8 if Pr == 4 else 16), (8 if Pr == 4 else 12)

        def run_search_bytes(sv, toehold):
            return (N * (16 + (24 if toehold else 16)) + 16 * sv["read_chunks"] + (16 if Pr == 4 else 32) * sv["ftab"] + b_dir * sv["slots"]
                    + b_ent * sv["dense"] + b_narrow * sv["searched_ranks"] + b_samp * sv["resamples"])

        if run_phi_slots:   # the slot kernel's K3 (k_locate_fill): one PhiSlot (4 x P; 16 packed) per step, the searched steps as on the slot layout
            k3_bytes = N * 28 + (16 if (Pr == 4 or (int(ix.n) >> 38) == 0) else 32) * rs_loc["phi_steps"] + (8 + 3 * 2 * Pr) * rs_loc["phi_searched"] + 8 * rs_loc["locs"]
        else:
            k3_bytes = N * 28 + b_phi_dir * rs_loc["phi_steps"] + b_phi_ent * rs_loc["probe_entries"] + 8 * rs_loc["locs"]
        run_alg = {"k_find_range<count>": run_search_bytes(rs_count, False), "k_find_range<toehold>": run_search_bytes(rs_toe, True), "k_locate_fill": k3_bytes}
        run_roof = {kk: {"alg_bytes": v, "alg_GBps": v / (ms_r[kk] * 1e-3) / 1e9, "frac_of_hbm_peak": v / (ms_r[kk] * 1e-3) / 1e9 / HBM_PEAK_GBS}
                    for kk, v in run_alg.items()}
        run_touched = {"per_read": {"search_steps": rs_toe["steps"] / N, "directory_gathers": rs_toe["slots"] / N, "run_list_entries_probed": rs_toe["dense"] / N,
                                    "narrowing_rounds": rs_toe["searched_ranks"] / N, "resamples": rs_toe["resamples"] / N,
                                    "phi_steps": rs_loc["phi_steps"] / N, "phi_entries_probed": rs_loc.get("probe_entries", 0) / N,
                                    "phi_steps_searched_beside_their_slot": rs_loc.get("phi_searched", 0) / N},
                       "search": rs_toe, "locate": rs_loc}
        same = all(bool((a == b).all().item()) for a, b in zip(ref_out, (d_lo, d_hi, d_k, d_loc_off))) and bool((ref_locs == d_locs[:total_locs]).all().item())
        rows.append({"layout": "runs", "depths": [d + 1 for d in range(8) if int(rb.layout_info().depth_mask_kept) >> d & 1], "symbols_per_gather": int(rb.info().kmer_steps), "hbm_bytes": int(rb.info().hbm_bytes), "ms": ms_r,
                     "identical_to_the_headline_replica_on_the_whole_batch": same, "roofline": run_roof, "touched": run_touched,
                     "phi": "slots of about n / r rows" if run_phi_slots else "list of sampled positions + directory",
                     "ranks": "bucket records (one 64-byte record per bucket)" if run_recs else "directories + run lists",
                     "count_locate_reads_per_s": rate_r})
        if args.markers:
            # the kernels beside the rb_align path on this layout (cooperative since round 3: k_runs_seeds.hip), same batch
            rb.set_markers(*marker_arrays)
            gstep()
            same_g = all(bool((a == b).all().item()) for a, b in zip(ref_g, d_g))
            sstep()
            same_s = (int(d_soff[-1].item()), int(d_moff[-1].item())) == ref_seed_counts
            run_other = {}
            assert int(L.rbg_marker_seeds_log_bytes(rb.h, 2 * N, 0)) <= log_bytes
            sstep_log()
            same_s = same_s and (int(d_soff[-1].item()), int(d_moff[-1].item())) == ref_seed_counts
            for name, fn in (("greedy_seed_ms", gstep), ("marker_seeds_ms", sstep), ("marker_seeds_logged_ms", sstep_log)):
                e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                e[0].record(stream); fn(); e[1].record(stream); fn(); e[2].record(stream)
                torch.cuda.synchronize()
                run_other[name] = min(e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]))
            rows[-1]["other_kernels"] = {"runs_layout": run_other, "slot_layout": slot_other_row if headline_runs else slot_other, "greedy_seeds_identical_to_slot_path": same_g,
                                         "marker_seed_counts_identical_to_slot_path": same_s}
            same = same and same_g and same_s
        # the same layout with run lists for some of the k-mer depths only (RBG_OPT_RUN_DEPTHS; include/rbg.h): a step takes
        # the longest stretch a kept depth covers -- the space of the depths left out against a step more per ragged stretch
        # (0x8B twice: the library's own choice -- bucket records and phi slots on an index this small -- and the minimal form of the
        #  layout: directories + run lists, phi over the list of sampled positions; 0x81: the first and the deepest only; 0x15: depths 1, 3, 5 --
        #  the default replica of round 4, five symbols per step)
        for depth_mask, minimal in (() if not same else ((0x8B, False), (0x8B, True), (0x81, False), (0x15, False))):
            rb.close()
            torch.cuda.empty_cache()
            with capi.default_option(capi.OPT_RANK_LAYOUT, capi.LAYOUT_RUNS), capi.default_option(capi.OPT_RUN_DEPTHS, depth_mask), \
                    capi.default_option(capi.OPT_RUN_REC, 1 if minimal else 0), capi.default_option(capi.OPT_RUN_PHI, 1 if minimal else 0):
                rb = ra.RowBowt.from_runs(inp["heads"], inp["lens"], inp["ssa"], inp["esa"], device=local_rank)
            ms_d = time_search()
            step()
            rate_d = time_locate(ms_d)
            step()
            same_d = all(bool((a == b).all().item()) for a, b in zip(ref_out, (d_lo, d_hi, d_k, d_loc_off))) and bool((ref_locs == d_locs[:total_locs]).all().item())
            li_d = rb.layout_info()
            row = {"layout": "runs", "depths": [d + 1 for d in range(8) if depth_mask >> d & 1], "symbols_per_gather": int(rb.info().kmer_steps),
                   "ranks": "bucket records" if sum(int(x) for x in li_d.rec_bytes) > 0 else "directories + run lists",
                   "phi": "slots of about n / r rows" if int(li_d.phi_slots) else "list of sampled positions + directory",
                   "hbm_bytes": int(rb.info().hbm_bytes), "ms": ms_d, "identical_to_the_headline_replica_on_the_whole_batch": same_d,
                   "count_locate_reads_per_s": rate_d}
            if args.markers:
                rb.set_markers(*marker_arrays)
                gstep()
                same_g = all(bool((a == b).all().item()) for a, b in zip(ref_g, d_g))
                sstep()
                same_s = (int(d_soff[-1].item()), int(d_moff[-1].item())) == ref_seed_counts
                other = {}
                sstep_log()
                same_s = same_s and (int(d_soff[-1].item()), int(d_moff[-1].item())) == ref_seed_counts
                for name, fn in (("greedy_seed_ms", gstep), ("marker_seeds_ms", sstep), ("marker_seeds_logged_ms", sstep_log)):
                    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                    e[0].record(stream); fn(); e[1].record(stream); fn(); e[2].record(stream)
                    torch.cuda.synchronize()
                    other[name] = min(e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]))
                row["other_kernels"] = {"runs_layout": other, "greedy_seeds_identical_to_slot_path": same_g, "marker_seed_counts_identical_to_slot_path": same_s}
                same_d = same_d and same_g and same_s
            rows.append(row)
            same = same and same_d
        if not same:
            out["space_speed"] = {"rows": rows}
            print(json.dumps(out))
            raise SystemExit("PARITY FAILURE: the run-indexed layout disagrees with the slot tables")
        # what the headline costs: the best count+locate rate among the replicas of at most 64 GB, and the rate of the replica a DEFAULT
        # rbg_load builds -- no option set: budget = a quarter of the free HBM, RBG_LAYOUT_AUTO (slot tables if all five symbols per
        # step fit that, else the run-indexed layout) -- built and timed here like the other rows
        headline_is_default = args.hbm_budget_gb == 0 and args.layout == "auto" and not args.pos_bytes
        rb.close()
        torch.cuda.empty_cache()
        with capi.default_option(capi.OPT_HBM_BUDGET_MB, 0), capi.default_option(capi.OPT_RANK_LAYOUT, capi.LAYOUT_AUTO), capi.default_option(capi.OPT_KMER_STEPS, capi.MAX_KMER_DEPTH):
            rb = ra.RowBowt.from_runs(inp["heads"], inp["lens"], inp["ssa"], inp["esa"], device=local_rank)
        ms_df = time_search()
        step()
        rate_df = time_locate(ms_df)
        step()
        same_df = all(bool((a == b).all().item()) for a, b in zip(ref_out, (d_lo, d_hi, d_k, d_loc_off))) and bool((ref_locs == d_locs[:total_locs]).all().item())
        ix_df, li_df = rb.info(), rb.layout_info()
        runs_df = int(ix_df.rank_layout) == capi.LAYOUT_RUNS
        out["value_library_default"] = {"value": rate_df, "hbm_bytes": int(ix_df.hbm_bytes), "layout": "runs" if runs_df else "slots",
                                        "symbols_per_gather": int(ix_df.kmer_steps), "hbm_budget": int(ix_df.hbm_budget), "ms": ms_df,
                                        "identical_to_the_headline_replica_on_the_whole_batch": same_df,
                                        **({"ranks": "bucket records" if sum(int(x) for x in li_df.rec_bytes) > 0 else "directories + run lists",
                                            "phi": "slots of about n / r rows" if int(li_df.phi_slots) else "list of sampled positions + directory",
                                            "depths": [d + 1 for d in range(8) if int(li_df.depth_mask_kept) >> d & 1]} if runs_df else {}),
                                        "budget": "a quarter of the HBM free at load (rbg_load with no option set: RBG_LAYOUT_AUTO)",
                                        "is_the_headline_replica": headline_is_default}
        if not same_df:
            print(json.dumps(out))
            raise SystemExit("PARITY FAILURE: the library-default replica disagrees with the headline replica")
        del ref_out, ref_locs
        within = [r_ for r_ in rows if r_["hbm_bytes"] <= 64e9]
        out["value_at_64GB"] = max((r_["count_locate_reads_per_s"] for r_ in within), default=None)
        out["space_speed"] = {"unit": "ms per launch of this run's batch (best of 3)", "rows": rows,
                              "note": "RBG_OPT_KMER_STEPS / the HBM budget rule pick the row; rbg_info reports which (symbols_per_gather, "
                                      "hbm_free_at_load, hbm_budget)"}

    # ---- --replicas G: the same count+locate step on G replicas held by THIS process (SURVEY 8e's replication without a process
    # per GPU: one build, rbg_replicate_many, no data-path collective; the reference is one process with one index, rb_align.cpp:176-178)
    if rank == 0 and args.replicas and world == 1:
        import threading
        G = args.replicas
        devs = [int(x) for x in args.replica_devices.split(",")] if args.replica_devices else list(range(G))
        if len(devs) != G or devs[0] != local_rank or any(d < 0 or d >= torch.cuda.device_count() for d in devs):
            raise SystemExit(f"--replica-devices must name {G} visible devices, the first of them {local_rank}: {devs}")
        t0 = time.time()
        reps = [rb] + (rb.replicate_many(devs[1:]) if G > 1 else [])
        torch.cuda.synchronize()
        t_rep = time.time() - t0
        step()                       # the primary's results of its own batch: what every copy must reproduce
        torch.cuda.synchronize()
        want = [t.cpu() for t in (d_lo, d_hi, d_k, d_loc_off, d_locs[:total_locs])]

        class Lane:
            def __init__(self, g):
                self.g, self.rb, self.dev = g, reps[g], torch.device("cuda", devs[g])
                with torch.cuda.device(self.dev):
                    self.stream = torch.cuda.Stream(self.dev)
                    self.st = self.stream.cuda_stream
                    text_g = text_for_replicas if self.dev == dev else text_for_replicas.to(self.dev)
                    # replica g's block of the global batch of G x --reads reads (replica 0: the batch the headline ran)
                    rd = reads if g == 0 else sp.sample_reads(text_g, info, N, m, seed=args.seed + 2 + g, sub_rate=0.1)[0]
                    self.seqs = torch.cat([rd.reshape(-1), torch.zeros(32, dtype=torch.uint8, device=self.dev)])
                    self.off = torch.arange(N + 1, device=self.dev, dtype=torch.int64) * m
                    self.lo, self.hi, self.k = (torch.empty(N, dtype=torch.int64, device=self.dev) for _ in range(3))
                    self.loc_off = torch.empty(N + 1, dtype=torch.int64, device=self.dev)
                    self.tmp = torch.empty(tmp_bytes, dtype=torch.uint8, device=self.dev)
                    self.ws = torch.empty(ws_bytes, dtype=torch.uint8, device=self.dev)
                    self.locs = None
                    self.same = None
                    torch.cuda.synchronize(self.dev)   # (the buffers were made on the device's default stream; the lane works on its own)
                    if g > 0:   # (the copy against the primary, on the primary's batch)
                        keep, self.seqs = self.seqs, d_seqs.to(self.dev)
                        torch.cuda.synchronize(self.dev)
                        self.search()
                        self.plan()
                        self.stream.synchronize()
                        self.locs = torch.empty(max(int(self.loc_off[-1].item()), 1), dtype=torch.int64, device=self.dev)
                        self.order()
                        self.fill()
                        self.stream.synchronize()
                        got = [t.cpu() for t in (self.lo, self.hi, self.k, self.loc_off, self.locs[:int(self.loc_off[-1].item())])]
                        self.same = all(a.shape == b.shape and bool((a == b).all()) for a, b in zip(got, want))
                        self.seqs = keep
                    self.search()
                    self.plan()
                    self.stream.synchronize()
                    self.total = int(self.loc_off[-1].item())
                    self.locs = torch.empty(max(self.total, 1), dtype=torch.int64, device=self.dev)
                self.t0 = self.t1 = 0.0

            def search(self):
                chk(L.rbg_find_range_w_toehold_dev(self.rb.h, self.seqs.data_ptr(), self.off.data_ptr(), N, self.lo.data_ptr(), self.hi.data_ptr(), self.k.data_ptr(), self.st), "find_range_w_toehold")

            def plan(self):
                chk(L.rbg_locate_plan_dev(self.rb.h, self.lo.data_ptr(), self.hi.data_ptr(), N, max_hits, self.loc_off.data_ptr(), self.tmp.data_ptr(), tmp_bytes, self.st), "locate_plan")

            def order(self):
                chk(L.rbg_locate_order_dev(self.rb.h, self.k.data_ptr(), N, self.ws.data_ptr(), ws_bytes, self.st), "locate_order")

            def fill(self):
                chk(L.rbg_locate_fill_dev(self.rb.h, self.lo.data_ptr(), self.hi.data_ptr(), self.k.data_ptr(), N, max_hits, self.loc_off.data_ptr(), self.locs.data_ptr(),
                                          self.ws.data_ptr(), self.st), "locate_fill")

            def run(self, gate):
                with torch.cuda.device(self.dev):
                    for _ in range(max(1, args.warmup)):
                        self.search(); self.plan(); self.order(); self.fill()
                    self.stream.synchronize()
                    gate()
                    self.t0 = time.perf_counter()
                    for _ in range(K):
                        self.search(); self.plan(); self.order(); self.fill()
                    self.stream.synchronize()
                    self.t1 = time.perf_counter()
                    gate()

        lanes = [Lane(g) for g in range(G)]
        for r_ in reps:
            r_.counters_reset()
        gate_obj, errs = threading.Barrier(G), []

        def lane_thread(ln):
            try:
                ln.run(gate_obj.wait)
            except BaseException as e:   # noqa: BLE001
                errs.append(e)
                gate_obj.abort()
        ths = [threading.Thread(target=lane_thread, args=(ln,)) for ln in lanes]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if errs:
            raise errs[0]
        el_r = max(ln.t1 for ln in lanes) - min(ln.t0 for ln in lanes)
        if len(set(devs)) == G and G > 1:
            rc_ = [int(x) for x in _cb.counters_allreduce_local(reps)]
            how = f"rbg_counters_allreduce_local over the {G} replicas (one grouped RCCL all-reduce)"
        else:
            rc_ = [int(x) for x in np.sum([r_.counters().astype(np.int64) for r_ in reps], axis=0)]
            how = f"summed on the host ({len(set(devs))} device(s) for {G} replicas: RCCL needs one device per rank)"
        same_all = all(ln.same for ln in lanes[1:])
        steps_counted = K + max(1, args.warmup)
        out["replicas_one_process"] = {
            "formed": G, "devices": devs, "replicate_s": t_rep,
            # what was replicated (the headline replica: --replicas skips the space_speed rebuilds) and what each target's peer copies cost
            "replicated": {"layout": "runs" if int(rb.info().rank_layout) == 2 else "slots", "hbm_bytes": int(rb.info().hbm_bytes), "is_the_headline_replica": True},
            "peer_copies": [{"device": devs[g], **reps[g].replicate_stats()} for g in range(1, G)],
            "fan_out_GBps": (sum(int(r_.info().hbm_bytes) for r_ in reps[1:]) / t_rep / 1e9) if G > 1 and t_rep > 0 else None,
            "value": G * N * K / el_r, "unit": "reads/s", "ms_per_step": el_r / K * 1e3,
            "per_replica_ms_per_step": [(ln.t1 - ln.t0) / K * 1e3 for ln in lanes], "locations_per_step": [ln.total for ln in lanes],
            "every_copy_identical_to_the_primary_on_its_batch": same_all,
            "counters": {"reads": rc_[0], "matched": rc_[1], "sum_occ": rc_[2], "sum_locs": rc_[3], "reduced_over": how,
                         "as_streamed": rc_[0] == G * N * steps_counted and rc_[2] == rc_[3] == steps_counted * sum(ln.total for ln in lanes)},
            "timing": "max over the replicas between two gates (one host thread and one HIP stream per replica)"}
        if not same_all or not out["replicas_one_process"]["counters"]["as_streamed"]:
            print(json.dumps(out))
            raise SystemExit("REPLICA FAILURE: a copy answers differently from the primary, or the reduced counters are not the stream's")
        # the line describes what ran: G replicas of one process
        out["single_replica"] = {"value": out["value"], "ms_per_step": out["ms_per_step"]}
        out.update({"value": G * N * K / el_r, "n_gpus": G, "ms_per_step": el_r / K * 1e3})
        out["config"]["parallelism"] = (f"index replicated x{G} in ONE process (built once, rbg_replicate_many to devices {devs}), {N} reads per replica per step, "
                                        "no data-path collective")
        for r_ in reps[1:]:
            r_.close()

    rb.close()
    return out, dict(rank=rank, world=world, use_dist=use_dist, args=args, dev=dev)


def pangenome_shape_block(args, dev):
    """BASELINE.json configs[3]'s index (r >= 1e8, positions beyond 32 bits, 150 bp reads made on the device) on this one GPU, through a DEFAULT
    rbg_load in a fresh child process (tools/pangenome_stream.py --preset driver), started once run() has returned and with it every buffer and
    the replica of the headline.  Not part of `value`: its own reads/s, kernel times, roofline, layout decisions and parity sample."""
    import gc
    import subprocess

    import torch
    gc.collect()
    torch.cuda.empty_cache()
    free_now, _t = torch.cuda.mem_get_info(dev)
    env_pg = {k: v for k, v in os.environ.items() if not k.startswith("RBG_")}
    t0 = time.time()
    try:
        pg = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pangenome_stream.py"), "--preset", "driver"], capture_output=True, text=True, env=env_pg, cwd=ROOT,
                            timeout=900)
    except subprocess.TimeoutExpired as e:
        return {"seconds_for_the_block": time.time() - t0, "hbm_free_when_started": int(free_now), "error": f"the child did not finish within 900 s: {e}", "fatal": False}
    blk = {"seconds_for_the_block": time.time() - t0, "hbm_free_when_started": int(free_now),
           "command": "python tools/pangenome_stream.py --preset driver   (a child process: a default rbg_load in a fresh process)"}
    try:
        pj = json.loads(pg.stdout.strip().splitlines()[-1])
        blk.update({"value": pj["value"], "unit": pj["unit"], "metric": pj["metric"], "value_excluding_read_generation": pj["value_excluding_read_generation"],
                    "kernel_ms_one_batch": pj["kernel_ms_one_batch"], "search_touched_per_read": pj["search_touched_per_read"], "roofline": pj["roofline"],
                    "index": pj["config"]["index"], "workload": pj["config"]["workload"], "reads_per_batch": pj["reads_per_batch"], "batches": pj["batches_per_gpu"],
                    "counters": pj["counters"], "parity": pj.get("parity"), "properties": pj.get("properties"), "peaks": pj.get("peaks")})
    except Exception as e:   # noqa: BLE001
        blk["error"] = f"{type(e).__name__}: {e}; rc {pg.returncode}; stderr tail: {pg.stderr[-1500:]}"
    if pg.returncode != 0 and "error" not in blk:
        blk["error"] = f"rc {pg.returncode}; stderr tail: {pg.stderr[-1500:]}"
    # a WRONG answer of the child is fatal to the whole line (its own exit says PARITY / PROPERTY / REPLICA FAILURE or a counter mismatch); a child that could not run
    # (no memory left on a shared device, a timeout) is reported in the block and on stderr and leaves the headline line standing
    if "error" in blk:
        wrong = any(w in (pg.stderr or "") for w in ("PARITY FAILURE", "PROPERTY FAILURE", "counter mismatch")) or \
            (blk.get("parity") is not None and not blk["parity"].get("bit_exact_vs_oracle", True))
        blk["fatal"] = bool(wrong)
    return blk


def main():
    res = run()
    if res is None:
        return
    out, c = res
    args, rank, world, use_dist = c["args"], c["rank"], c["world"], c["use_dist"]
    default_workload = (args.L, args.H, args.reads, args.read_len, args.site_rate) == (40_000_000, 50, 10_000_000, 100, 0.01)
    if rank == 0 and world == 1 and not use_dist and default_workload and not args.no_pangenome_shape and not args.replicas:
        stage("pangenome_shape block (child process)")
        out["pangenome_shape"] = pangenome_shape_block(args, c["dev"])
        if "error" in out["pangenome_shape"]:
            print("[bench] pangenome_shape block failed: " + out["pangenome_shape"]["error"], file=sys.stderr, flush=True)
            if out["pangenome_shape"].get("fatal"):
                print(json.dumps(out))
                raise SystemExit("pangenome_shape block: WRONG ANSWERS from the child (see stderr)")
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        import torch.distributed as dist
        stage("final barrier")
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except BaseException as e:   # noqa: BLE001
        if not (isinstance(e, SystemExit) and e.code in (0, None)):
            print(f"[bench] rank {STAGE['rank']} (pid {os.getpid()}, LOCAL_RANK {os.environ.get('LOCAL_RANK', '-')}) FAILED at stage "
                  f"'{STAGE['name']}' after {time.time() - STAGE['t0']:.1f}s: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        raise
