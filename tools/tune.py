#!/usr/bin/env python3
"""Layout/launch sweep on the bench index (GPU box only): builds the synthetic pangenome once, then
one index replica per (rank_shift, phi_shift, block) and times the three kernels."""
import argparse, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rowbowt_amd as ra
from rowbowt_amd import capi
if os.environ.get("RBG_TUNE_LIB"):  # A/B of two builds of the library
    capi._SO = os.path.join(ROOT, "rowbowt_amd", os.environ["RBG_TUNE_LIB"])
from rowbowt_amd.tools import synth_pangenome as sp

ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=40_000_000)
ap.add_argument("--H", type=int, default=50)
ap.add_argument("--reads", type=int, default=10_000_000)
ap.add_argument("--configs", default="-1:-1:256")
args = ap.parse_args()
dev = torch.device("cuda:0")
text, info = sp.make_text(args.L, args.H, 0.01, 20240229, dev)
sa = sp.suffix_array(text)
inp = sp.index_inputs(text, sa)
del sa
N, m = args.reads, 100
reads, _ = sp.sample_reads(text, info, N, m, seed=20240231, sub_rate=0.1)
del text
torch.cuda.empty_cache()
d_seqs = reads.reshape(-1)
d_off = torch.arange(N + 1, device=dev, dtype=torch.int64) * m
d_lo, d_hi, d_k = (torch.empty(N, dtype=torch.int64, device=dev) for _ in range(3))
d_loc_off = torch.empty(N + 1, dtype=torch.int64, device=dev)
L = ra.lib()
tmp_bytes = L.rbg_locate_plan_tmp_bytes(N)
d_tmp = torch.empty(tmp_bytes, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
MAXU = 2**64 - 1
d_locs = None
print(f"n={inp['n']} r={inp['r']}", flush=True)
for cfg in args.configs.split(","):
    parts = [int(x) for x in cfg.split(":")]
    rs, ps, bt = parts[:3]
    ks = parts[3] if len(parts) > 3 else 5
    pb = parts[4] if len(parts) > 4 else 0
    ra.set_default_option(capi.OPT_POS_BYTES, pb)
    ra.set_default_option(capi.OPT_KMER_STEPS, ks)
    ra.set_default_option(capi.OPT_RANK_BUCKET_SHIFT, rs)
    ra.set_default_option(capi.OPT_PHI_BUCKET_SHIFT, ps)
    ra.set_default_option(capi.OPT_BLOCK_THREADS, bt)
    if os.environ.get("RBG_TUNE_DEEP_SHIFT"):
        ra.set_default_option(capi.OPT_DEEP_BUCKET_SHIFT, int(os.environ["RBG_TUNE_DEEP_SHIFT"]))
    if os.environ.get("RBG_TUNE_FTAB_K"):
        ra.set_default_option(capi.OPT_FTAB_K, int(os.environ["RBG_TUNE_FTAB_K"]))
    if os.environ.get("RBG_TUNE_DENSE"):
        ra.set_default_option(capi.OPT_DENSE_OVERFLOW, int(os.environ["RBG_TUNE_DENSE"]))
    if os.environ.get("RBG_TUNE_LAYOUT"):   # "runs": the run-indexed layout (space proportional to r); optional sixth field of a config = LDS KB of its top level
        ra.set_default_option(capi.OPT_RANK_LAYOUT, capi.LAYOUT_RUNS if os.environ["RBG_TUNE_LAYOUT"] == "runs" else capi.LAYOUT_SLOTS)
        if len(parts) > 6:   # seventh field: RBG_RANK_REC (runs per bucket record; 0 = directories and run lists only)
            os.environ["RBG_RANK_REC"] = str(parts[6])
        else:
            os.environ.pop("RBG_RANK_REC", None)
        if len(parts) > 7:   # eighth field: RBG_RANK_DIR_RUNS x 4 (runs per directory bucket, in quarters)
            os.environ["RBG_RANK_DIR_RUNS"] = str(parts[7] / 4)
        else:
            os.environ.pop("RBG_RANK_DIR_RUNS", None)
    if os.environ.get("RBG_TUNE_BUDGET_MB"):
        ra.set_default_option(capi.OPT_HBM_BUDGET_MB, int(os.environ["RBG_TUNE_BUDGET_MB"]))
    t0 = time.time()
    rb = ra.RowBowt.from_runs(inp["heads"], inp["lens"], inp["ssa"], inp["esa"], device=0)
    i = rb.info()
    def t(fn, reps=3):
        fn(); torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        e[0].record()
        for _ in range(reps): fn()
        e[1].record(); torch.cuda.synchronize()
        return e[0].elapsed_time(e[1]) / reps
    ms_c = t(lambda: L.rbg_find_range_dev(rb.h, d_seqs.data_ptr(), d_off.data_ptr(), N, d_lo.data_ptr(), d_hi.data_ptr(), st))
    ms_t = t(lambda: L.rbg_find_range_w_toehold_dev(rb.h, d_seqs.data_ptr(), d_off.data_ptr(), N, d_lo.data_ptr(), d_hi.data_ptr(), d_k.data_ptr(), st))
    wsb_p = L.rbg_pack_ws_bytes(N, N * m)
    d_pws = torch.empty(wsb_p, dtype=torch.uint8, device=dev)
    ms_pk = t(lambda: L.rbg_pack_reads_dev(rb.h, d_seqs.data_ptr(), d_off.data_ptr(), N, N * m, d_pws.data_ptr(), wsb_p, st))
    ms_pc = t(lambda: L.rbg_find_range_packed_dev(rb.h, d_pws.data_ptr(), d_seqs.data_ptr(), d_off.data_ptr(), N, N * m, d_lo.data_ptr(), d_hi.data_ptr(), st))
    ms_pt = t(lambda: L.rbg_find_range_w_toehold_packed_dev(rb.h, d_pws.data_ptr(), d_seqs.data_ptr(), d_off.data_ptr(), N, N * m, d_lo.data_ptr(), d_hi.data_ptr(), d_k.data_ptr(), st))
    L.rbg_locate_plan_dev(rb.h, d_lo.data_ptr(), d_hi.data_ptr(), N, MAXU, d_loc_off.data_ptr(), d_tmp.data_ptr(), tmp_bytes, st)
    if d_locs is None:
        d_locs = torch.empty(int(d_loc_off[-1].item()), dtype=torch.int64, device=dev)
    ms_f0 = t(lambda: L.rbg_locate_fill_dev(rb.h, d_lo.data_ptr(), d_hi.data_ptr(), d_k.data_ptr(), N, MAXU, d_loc_off.data_ptr(), d_locs.data_ptr(), None, st))
    wsb = L.rbg_locate_order_ws_bytes(N)
    d_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    ms_o = t(lambda: L.rbg_locate_order_dev(rb.h, d_k.data_ptr(), N, d_ws.data_ptr(), wsb, st))
    ms_f = t(lambda: L.rbg_locate_fill_dev(rb.h, d_lo.data_ptr(), d_hi.data_ptr(), d_k.data_ptr(), N, MAXU, d_loc_off.data_ptr(), d_locs.data_ptr(), d_ws.data_ptr(), st))
    d_qs, d_qe = (torch.empty(N, dtype=torch.int64, device=dev) for _ in range(2))
    ms_g = t(lambda: L.rbg_greedy_longest_seed_dev(rb.h, d_seqs.data_ptr(), d_off.data_ptr(), N, 20, d_lo.data_ptr(), d_hi.data_ptr(), d_qs.data_ptr(), d_qe.data_ptr(), d_k.data_ptr(), st))
    print(f"cfg pos_bytes={i.pos_bytes} kmer_steps={i.kmer_steps} depth_runs={list(i.depth_runs)} ftab_k={i.ftab_k} rank_shift={i.rank_bucket_shift}({rs}) phi_shift={i.phi_bucket_shift}({ps}) block={bt}: hbm={i.hbm_bytes/1e9:.2f}GB "
          f"rank_ovf={i.rank_slots_overflow}/{i.rank_slots} phi_ovf={i.phi_slots_overflow}/{i.phi_slots}  "
          f"count={ms_c:.2f}ms toehold={ms_t:.2f}ms pack={ms_pk:.2f}ms packed_count={ms_pc:.2f}ms packed_toehold={ms_pt:.2f}ms fill(unordered)={ms_f0:.2f}ms order={ms_o:.2f}ms fill={ms_f:.2f}ms greedy_seed={ms_g:.2f}ms  build={time.time()-t0:.1f}s", flush=True)
    rb.close()
