#!/usr/bin/env python3
"""Experiment: does ordering the phi chains by toehold text position (DRAM-row / L2 locality) speed up
k_locate_fill?  GPU box only; not part of the library."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rowbowt_amd as ra
from rowbowt_amd import capi
from rowbowt_amd.tools import synth_pangenome as sp

dev = torch.device("cuda:0")
text, info = sp.make_text(40_000_000, 50, 0.01, 20240229, dev)
sa = sp.suffix_array(text)
inp = sp.index_inputs(text, sa)
del sa
N, m = 10_000_000, 100
reads, _ = sp.sample_reads(text, info, N, m, seed=20240231, sub_rate=0.1)
del text
torch.cuda.empty_cache()
ra.set_default_option(capi.OPT_KMER_STEPS, 2)  # small replica: only the locate kernel is studied
rb = ra.RowBowt.from_runs(inp["heads"], inp["lens"], inp["ssa"], inp["esa"], device=0)
L = ra.lib()
d_seqs = reads.reshape(-1)
d_off = torch.arange(N + 1, device=dev, dtype=torch.int64) * m
d_lo, d_hi, d_k = (torch.empty(N, dtype=torch.int64, device=dev) for _ in range(3))
d_loc_off = torch.empty(N + 1, dtype=torch.int64, device=dev)
tmp_bytes = L.rbg_locate_plan_tmp_bytes(N)
d_tmp = torch.empty(tmp_bytes, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
MAXU = 2**64 - 1
L.rbg_find_range_w_toehold_dev(rb.h, d_seqs.data_ptr(), d_off.data_ptr(), N, d_lo.data_ptr(), d_hi.data_ptr(), d_k.data_ptr(), st)
L.rbg_locate_plan_dev(rb.h, d_lo.data_ptr(), d_hi.data_ptr(), N, MAXU, d_loc_off.data_ptr(), d_tmp.data_ptr(), tmp_bytes, st)
total = int(d_loc_off[-1].item())
d_locs = torch.empty(total, dtype=torch.int64, device=dev)
d_locs2 = torch.empty(total, dtype=torch.int64, device=dev)

def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    for _ in range(reps): fn()
    e[1].record(); torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]) / reps

base = t(lambda: L.rbg_locate_fill_dev(rb.h, d_lo.data_ptr(), d_hi.data_ptr(), d_k.data_ptr(), N, MAXU, d_loc_off.data_ptr(), d_locs.data_ptr(), None, st))
print(f"fill, input order: {base:.2f} ms")
for name, key in (("toehold k", d_k), ("k mod unit (locus)", d_k % info["unit"]), ("lo (SA order)", d_lo)):
    t0 = time.perf_counter()
    perm = torch.argsort(key)
    torch.cuda.synchronize()
    t_sort = (time.perf_counter() - t0) * 1e3
    lo_p, hi_p, k_p = d_lo[perm].contiguous(), d_hi[perm].contiguous(), d_k[perm].contiguous()
    off_p = torch.cat([d_loc_off[:-1][perm], d_loc_off[-1:]]).contiguous()
    ms = t(lambda: L.rbg_locate_fill_dev(rb.h, lo_p.data_ptr(), hi_p.data_ptr(), k_p.data_ptr(), N, MAXU, off_p.data_ptr(), d_locs2.data_ptr(), None, st))
    same = bool((d_locs == d_locs2).all().item())
    print(f"fill, chains ordered by {name}: {ms:.2f} ms (torch argsort {t_sort:.1f} ms) identical output: {same}")
