#!/usr/bin/env python3
"""Sweep of the host pipeline's chunk size (RBG_HOST_CHUNK_READS) for rbg_find_range on the bench index: where the
PCIe-inclusive rate of the host-pointer calls comes from.  GPU box only."""
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
seqs = reads.cpu().numpy().reshape(-1)
off = (np.arange(N + 1, dtype=np.uint64) * m)
del text, reads
torch.cuda.empty_cache()
ra.set_default_option(capi.OPT_KMER_STEPS, int(os.environ.get("SWEEP_KMER_STEPS", "5")))
rb = ra.RowBowt.from_runs(inp["heads"], inp["lens"], inp["ssa"], inp["esa"], device=0)
L = ra.lib()
lo, hi = (np.zeros(N, np.uint64) for _ in range(2))
p = lambda a: a.ctypes.data_as(capi.VP)
for chunk in [int(x) for x in os.environ.get("SWEEP_CHUNKS", "125000,250000,500000,1000000,2000000,5000000").split(",")]:
    os.environ["RBG_HOST_CHUNK_READS"] = str(chunk)
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        assert L.rbg_find_range(rb.h, p(seqs), p(off), N, p(lo), p(hi)) == 0
        best = min(best, time.perf_counter() - t0)
    print(f"chunk {chunk:8d} reads: rbg_find_range {best * 1e3:7.2f} ms -> {N / best:.3e} reads/s", flush=True)
