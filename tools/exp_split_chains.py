#!/usr/bin/env python3
"""Experiment (GPU box): what would walking every range from two toeholds buy K3?  Same phi steps, same
outputs, but each read's chain is cut into `parts` equal pieces whose starting SA values are taken from a
first full run (so this is the upper bound for a two-ended walk: phi only, no second toehold cost in K2)."""
import argparse, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rowbowt_amd as ra
from rowbowt_amd.tools import synth_pangenome as sp

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=10_000_000)
args = ap.parse_args()
dev = torch.device("cuda:0")
text, info = sp.make_text(40_000_000, 50, 0.01, 20240229, dev)
sa = sp.suffix_array(text)
inp = sp.index_inputs(text, sa)
del sa
N, m = args.reads, 100
reads, _ = sp.sample_reads(text, info, N, m, seed=20240231, sub_rate=0.1)
del text
torch.cuda.empty_cache()
rb = ra.RowBowt.from_runs(inp["heads"], inp["lens"], inp["ssa"], inp["esa"], device=0)
L = ra.lib()
st = torch.cuda.current_stream().cuda_stream
MAXU = 2**64 - 1
d_seqs = reads.reshape(-1)
d_off = torch.arange(N + 1, device=dev, dtype=torch.int64) * m
d_lo, d_hi, d_k = (torch.empty(N, dtype=torch.int64, device=dev) for _ in range(3))
L.rbg_find_range_w_toehold_dev(rb.h, d_seqs.data_ptr(), d_off.data_ptr(), N, d_lo.data_ptr(), d_hi.data_ptr(), d_k.data_ptr(), st)


def run(lo, hi, k, tag):
    n = lo.numel()
    loc_off = torch.empty(n + 1, dtype=torch.int64, device=dev)
    tb = L.rbg_locate_plan_tmp_bytes(n)
    tmp = torch.empty(tb, dtype=torch.uint8, device=dev)
    L.rbg_locate_plan_dev(rb.h, lo.data_ptr(), hi.data_ptr(), n, MAXU, loc_off.data_ptr(), tmp.data_ptr(), tb, st)
    total = int(loc_off[-1].item())
    locs = torch.empty(total, dtype=torch.int64, device=dev)
    wsb = L.rbg_locate_order_ws_bytes(n)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)

    def t(fn, reps=3):
        fn(); torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        e[0].record()
        for _ in range(reps): fn()
        e[1].record(); torch.cuda.synchronize()
        return e[0].elapsed_time(e[1]) / reps
    ms_o = t(lambda: L.rbg_locate_order_dev(rb.h, k.data_ptr(), n, ws.data_ptr(), wsb, st))
    ms_f = t(lambda: L.rbg_locate_fill_dev(rb.h, lo.data_ptr(), hi.data_ptr(), k.data_ptr(), n, MAXU, loc_off.data_ptr(), locs.data_ptr(), ws.data_ptr(), st))
    print(f"{tag}: chains={n} locs={total} order={ms_o:.2f}ms fill={ms_f:.2f}ms", flush=True)
    return loc_off, locs


loc_off, locs = run(d_lo, d_hi, d_k, "one toehold per read")
occ = torch.where(d_hi >= d_lo, d_hi - d_lo + 1, torch.zeros_like(d_lo))
for parts in (2, 4):
    los, his, ks = [], [], []
    for p in range(parts):
        # piece p covers ranks [a, b) counted from hi downwards; its rows are [hi-b+1, hi-a]
        a = occ * p // parts
        b = occ * (p + 1) // parts
        nonempty = b > a
        hi_p = torch.where(nonempty, d_hi - a, torch.zeros_like(d_hi))
        lo_p = torch.where(nonempty, d_hi - b + 1, torch.ones_like(d_lo))
        idx = (loc_off[:-1] + a).clamp(max=max(locs.numel() - 1, 0))
        k_p = torch.where(nonempty, locs[idx], torch.zeros_like(d_k))
        los.append(lo_p); his.append(hi_p); ks.append(k_p)
    lo2, hi2, k2 = (torch.stack(x, dim=1).reshape(-1).contiguous() for x in (los, his, ks))
    off2, locs2 = run(lo2, hi2, k2, f"{parts} pieces per read")
    assert bool((locs2 == locs).all().item()), "pieces must reproduce the same location list"
