#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in (host-pointer) entry points on the bench index: the number
DESIGN.md quotes beside (never instead of) the HBM-resident `value`.  GPU box only."""
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
rb = ra.RowBowt.from_runs(inp["heads"], inp["lens"], inp["ssa"], inp["esa"], device=0)
L = ra.lib()
lo, hi, k, cnt = (np.zeros(N, np.uint64) for _ in range(4))
p = lambda a: a.ctypes.data_as(capi.VP)
def cpu_quota():
    try:
        a, b = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return "no CPU quota" if a == "max" else f"CPU quota {float(a) / float(b):.0f}"
    except Exception:
        return "CPU quota unknown"
print(f"host-pointer API, {N} x {m} bp reads in pageable host memory (numpy), n={inp['n']} r={inp['r']}, {os.cpu_count()} logical CPUs, {cpu_quota()}, "
      f"RBG_HOST_THREADS={os.environ.get('RBG_HOST_THREADS', 'default')}; best of 4 calls, and the mean of 20 calls back to back (what a CPU quota lets through)")
for packed, what in ((1, "2-bit codes packed by CPU threads (default)"), (0, "bytes over PCIe")):
    capi.set_default_option(capi.OPT_PACKED_READS, packed)
    res = {}
    for name, fn in (("rbg_find_range", lambda: L.rbg_find_range(rb.h, p(seqs), p(off), N, p(lo), p(hi))),
                     ("rbg_count", lambda: L.rbg_count(rb.h, p(seqs), p(off), N, p(cnt))),
                     ("rbg_find_range_w_toehold", lambda: L.rbg_find_range_w_toehold(rb.h, p(seqs), p(off), N, p(lo), p(hi), p(k)))):
        best = 1e9
        for rep in range(4):
            t0 = time.perf_counter()
            assert fn() == 0
            best = min(best, time.perf_counter() - t0)
        t0 = time.perf_counter()
        for rep in range(20):
            assert fn() == 0
        res[name] = (best, (time.perf_counter() - t0) / 20)
    print(f"  reads cross as {what}:")
    for name, (t, ts) in res.items():
        print(f"    {name:28s} {t * 1e3:8.1f} ms  -> {N / t:.3e} reads/s   sustained {ts * 1e3:8.1f} ms -> {N / ts:.3e} reads/s")
capi.set_default_option(capi.OPT_PACKED_READS, 1)
t1 = time.perf_counter()
lo, hi, k = rb.find_range_w_toehold(seqs, off)
t2 = time.perf_counter()
loc_off, locs = rb.locs_at(lo, hi, k)
t3 = time.perf_counter()
print(f"  rbg_locs_at ({len(locs)} locations, {len(locs) * 8 / 1e9:.1f} GB D2H)  {t3 - t2:.3f} s")
print(f"  count+locate end to end            {t3 - t1:.3f} s  -> {N / (t3 - t1):.3e} reads/s")
