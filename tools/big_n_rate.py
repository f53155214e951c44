#!/usr/bin/env python3
"""Rates at n > 2^32 (GPU box only).  No text of that size can be suffix-sorted here, but the search only sees
the run-length BWT and its samples: a synthetic run list (random heads/lengths, one terminator, distinct
random samples; tests/test_gpu_parity.py::test_positions_beyond_32_bits checks this construction against the
oracle) gives an index with 8-byte positions, and reads that match are read off LF walks done on the GPU.
Prints the kernel times of count, count+toehold and locate for N reads of m symbols."""
import argparse, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rowbowt_amd as ra
from rowbowt_amd import capi

ap = argparse.ArgumentParser()
ap.add_argument("--runs", type=int, default=20_000_000)
ap.add_argument("--max-len", type=int, default=500)
ap.add_argument("--reads", type=int, default=1_000_000)  # the LF walks that make the reads run on the host API: 4 M reads take 10 minutes
ap.add_argument("--read-len", type=int, default=100)
args = ap.parse_args()
rng = np.random.default_rng(4242)
r = args.runs
sym = np.frombuffer(b"ACGT", dtype=np.uint8)
step = rng.integers(1, 4, size=r, dtype=np.int64)
step[0] = 0
heads = sym[np.cumsum(step) % 4]
lens = rng.integers(1, args.max_len, size=r, dtype=np.int64).astype(np.uint64)
heads[r // 3], lens[r // 3] = 1, 1
n = int(lens.sum())
stride = n // (2 * r)
vals = np.arange(2 * r, dtype=np.uint64) * np.uint64(stride) + rng.integers(0, stride, size=2 * r).astype(np.uint64)
rng.shuffle(vals)
t0 = time.time()
rb = ra.RowBowt.from_runs(heads, lens, vals[:r].copy(), vals[r:].copy(), device=0)
i = rb.info()
print(f"n={i.n} r={i.r} pos_bytes={i.pos_bytes} kmer_steps={i.kmer_steps} hbm={i.hbm_bytes/1e9:.1f}GB load={time.time()-t0:.1f}s", flush=True)
# matching reads: LF walks, all reads in lock step (c_t = bwt[row_t], row_{t+1} = LF(row_t, c_t)); read = reversed symbols
N, m = args.reads, args.read_len
starts = np.concatenate([[0], np.cumsum(lens.astype(np.int64))]).astype(np.int64)
rows = rng.integers(0, n, size=N).astype(np.uint64)
reads = np.empty((N, m), dtype=np.uint8)
for t in range(m):
    c = heads[np.searchsorted(starts, rows.astype(np.int64), side="right") - 1]
    term = c == 1                      # a walk that reaches the terminator: continue with another symbol (the read just stops matching)
    c = np.where(term, sym[0], c)
    reads[:, m - 1 - t] = c
    nlo, nhi = rb.LF(rows, rows, c)
    rows = np.where(nhi >= nlo, nlo, rows)
dev = torch.device("cuda:0")
d_seqs = torch.from_numpy(reads.reshape(-1)).to(dev)
d_off = torch.arange(N + 1, device=dev, dtype=torch.int64) * m
d_lo, d_hi, d_k = (torch.empty(N, dtype=torch.int64, device=dev) for _ in range(3))
d_loc_off = torch.empty(N + 1, dtype=torch.int64, device=dev)
L = ra.lib()
st = torch.cuda.current_stream().cuda_stream
tmp_bytes = L.rbg_locate_plan_tmp_bytes(N)
d_tmp = torch.empty(tmp_bytes, dtype=torch.uint8, device=dev)
MAXH = 64
def t(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    for _ in range(reps): fn()
    e[1].record(); torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]) / reps
ms_c = t(lambda: L.rbg_find_range_dev(rb.h, d_seqs.data_ptr(), d_off.data_ptr(), N, d_lo.data_ptr(), d_hi.data_ptr(), st))
ms_t = t(lambda: L.rbg_find_range_w_toehold_dev(rb.h, d_seqs.data_ptr(), d_off.data_ptr(), N, d_lo.data_ptr(), d_hi.data_ptr(), d_k.data_ptr(), st))
matched = int((d_hi >= d_lo).sum().item())
L.rbg_locate_plan_dev(rb.h, d_lo.data_ptr(), d_hi.data_ptr(), N, MAXH, d_loc_off.data_ptr(), d_tmp.data_ptr(), tmp_bytes, st)
nloc = int(d_loc_off[-1].item())
d_locs = torch.empty(max(nloc, 1), dtype=torch.int64, device=dev)
wsb = L.rbg_locate_order_ws_bytes(N)
d_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
ms_o = t(lambda: L.rbg_locate_order_dev(rb.h, d_k.data_ptr(), N, d_ws.data_ptr(), wsb, st))
ms_f = t(lambda: L.rbg_locate_fill_dev(rb.h, d_lo.data_ptr(), d_hi.data_ptr(), d_k.data_ptr(), N, MAXH, d_loc_off.data_ptr(), d_locs.data_ptr(), d_ws.data_ptr(), st))
print(f"{N} x {m} symbols, matched {matched}, locations {nloc} (max_hits {MAXH}): count {ms_c:.2f} ms = {N/ms_c*1e3:.3e} reads/s, "
      f"count+toehold {ms_t:.2f} ms = {N/ms_t*1e3:.3e} reads/s, order {ms_o:.2f} ms, fill {ms_f:.2f} ms; "
      f"count+locate {N/(ms_t+ms_o+ms_f)*1e3:.3e} reads/s", flush=True)
rb.close()
