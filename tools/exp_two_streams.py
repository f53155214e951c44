#!/usr/bin/env python3
"""Experiment: successive batches on two HIP streams so that batch i+1's backward search overlaps
batch i's locate (K2 is gather-request-bound, the ordered K3 is not).  GPU box only."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rowbowt_amd as ra
from rowbowt_amd.tools import synth_pangenome as sp

dev = torch.device("cuda:0")
text, info = sp.make_text(40_000_000, 50, 0.01, 20240229, dev)
sa = sp.suffix_array(text)
inp = sp.index_inputs(text, sa)
del sa
N, m = 10_000_000, 100
L = ra.lib()
MAXU = 2**64 - 1
NB = 2
bufs = []
for b in range(NB):
    reads, _ = sp.sample_reads(text, info, N, m, seed=20240231 + b, sub_rate=0.1)
    bufs.append(dict(seqs=reads.reshape(-1), off=torch.arange(N + 1, device=dev, dtype=torch.int64) * m,
                     lo=torch.empty(N, dtype=torch.int64, device=dev), hi=torch.empty(N, dtype=torch.int64, device=dev),
                     k=torch.empty(N, dtype=torch.int64, device=dev), loc_off=torch.empty(N + 1, dtype=torch.int64, device=dev)))
del text
torch.cuda.empty_cache()
rb = ra.RowBowt.from_runs(inp["heads"], inp["lens"], inp["ssa"], inp["esa"], device=0)
tmp_bytes = L.rbg_locate_plan_tmp_bytes(N)
ws_bytes = L.rbg_locate_order_ws_bytes(N)
for b in bufs:
    b["tmp"] = torch.empty(tmp_bytes, dtype=torch.uint8, device=dev)
    b["ws"] = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)

def step(b, st):
    L.rbg_find_range_w_toehold_dev(rb.h, b["seqs"].data_ptr(), b["off"].data_ptr(), N, b["lo"].data_ptr(), b["hi"].data_ptr(), b["k"].data_ptr(), st)
    L.rbg_locate_plan_dev(rb.h, b["lo"].data_ptr(), b["hi"].data_ptr(), N, MAXU, b["loc_off"].data_ptr(), b["tmp"].data_ptr(), tmp_bytes, st)
    if "locs" in b:
        L.rbg_locate_order_dev(rb.h, b["k"].data_ptr(), N, b["ws"].data_ptr(), ws_bytes, st)
        L.rbg_locate_fill_dev(rb.h, b["lo"].data_ptr(), b["hi"].data_ptr(), b["k"].data_ptr(), N, MAXU, b["loc_off"].data_ptr(), b["locs"].data_ptr(), b["ws"].data_ptr(), st)

s0 = torch.cuda.current_stream()
for b in bufs:
    step(b, s0.cuda_stream)
    torch.cuda.synchronize()
    b["locs"] = torch.empty(int(b["loc_off"][-1].item()), dtype=torch.int64, device=dev)
streams = [torch.cuda.Stream() for _ in range(NB)]
K = 8
for mode in ("one stream", "two streams"):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(K):
            b = bufs[s % NB]
            st = s0 if mode == "one stream" else streams[s % NB]
            step(b, st.cuda_stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"{mode}: {dt / K * 1e3:.2f} ms per step -> {N * K / dt:.3e} reads/s")
