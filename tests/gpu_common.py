"""Helpers shared by the GPU parity files (tests/test_gpu_*.py)."""
import json
import os
import re
import sys

import numpy as np
import pytest

import golden_values as G
import orc
import rowbowt_amd as ra
from rowbowt_amd import capi

__all__ = ["ROOT", "DEFAULT_KMER_STEPS", "split", "_check_marker_seeds", "_run_cli", "_run_rb_markers", "_random_run_index", "_lf_walk_reads", "_with_layout",
           "_run_indexed_checks"]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAXU = G.MAXU
ALL = ra.LoadRbwtFlag.SA | ra.LoadRbwtFlag.MA
DEFAULT_KMER_STEPS = capi.MAX_KMER_DEPTH   # the library's default for RBG_OPT_KMER_STEPS (what a test that changed it puts back)


def split(off, vals):
    return [vals[int(off[i]):int(off[i + 1])].tolist() for i in range(len(off) - 1)]


def _check_marker_seeds(rb, o, reads, wsize, max_range, ftab_k=0):
    seqs, off = ra.pack_reads(reads)
    seed_off, seeds, mk = rb.get_markers_greedy_seeding(seqs, off, wsize, max_range, ftab_k)
    nseed = nmk = 0
    for i, q in enumerate(reads):
        want = o.markers_greedy_seeding(q, wsize, max_range, ftab_k)
        got = seeds[int(seed_off[i]):int(seed_off[i + 1])]
        assert len(got) == len(want), (i, q)
        for g, (wl, wh, wqs, wqe, wm) in zip(got, want):
            assert (int(g[0]), int(g[1]), int(g[2]), int(g[3])) == (wl, wh, wqs, wqe), (i, q)
            assert mk[int(g[4]):int(g[5])].tolist() == wm, (i, q)
            nmk += len(wm)
        nseed += len(want)
    return nseed, nmk


# ---- the rb_align-compatible CLI: byte-exact stdout (reference src/rb_align.cpp:118-145) ---------
def _run_cli(args, env=None):
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rowbowt_amd", "rb_align")
    p = subprocess.run([exe] + args, capture_output=True, timeout=120, env=dict(os.environ, **env) if env else None)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


# ---- the rb_markers-compatible CLI (reference src/rb_markers.cpp, default seeding mode) ----------
def _run_rb_markers(args):
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rowbowt_amd", "rb_markers")
    p = subprocess.run([exe] + args, capture_output=True, timeout=120)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


def _random_run_index(rng, r, max_len, term_at=None):
    """A synthetic run list (random heads over ACGT with neighbouring runs different, random lengths, distinct random
    samples below n) with one terminator run: rank, LF, the toehold bookkeeping and phi are arithmetic on these arrays
    alone, so they define the answers completely -- for the oracle and for the device alike (no text needed).
    One caveat: the samples of such a list are not LF-consistent as a BWT's are, so the toehold after a MULTI-symbol step is
    defined by them only up to the choice between two run ends that coincide (the same SA value in a BWT): compare ranges at
    any depth, toeholds and locations at single-symbol steps (or on a true BWT: synth.SynthIndex, the pangenome streams)."""
    sym = np.frombuffer(b"ACGT", dtype=np.uint8)
    step = rng.integers(1, 4, size=r, dtype=np.int64)
    step[0] = 0
    heads = sym[np.cumsum(step) % 4]
    lens = rng.integers(1, max_len, size=r, dtype=np.int64).astype(np.uint64)
    t = r // 3 if term_at is None else term_at
    heads[t], lens[t] = 1, 1
    n = int(lens.sum())
    stride = n // (2 * r)
    vals = (np.arange(2 * r, dtype=np.uint64) * np.uint64(stride) + rng.integers(0, stride, size=2 * r).astype(np.uint64))
    rng.shuffle(vals)
    return heads, lens, vals[:r].copy(), vals[r:].copy(), n


def _lf_walk_reads(o, heads, lens, n, rng, count, max_len):
    """reads that match: c0 = bwt[i0], i1 = LF(i0), c1 = bwt[i1], ... is matched by the pattern c_k ... c1 c0"""
    starts = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
    reads = []
    for row in rng.integers(0, n, size=count):
        row, m, q = int(row), int(rng.integers(1, max_len)), bytearray()
        for _ in range(m):
            c = int(heads[np.searchsorted(starts, row, side="right") - 1])
            q.append(c)
            row = o.LF(row, row, c)[0]
        reads.append(bytes(q[::-1]))
    return reads


def _with_layout(layout, build):
    ra.set_default_option(capi.OPT_RANK_LAYOUT, layout)
    try:
        return build()
    finally:
        ra.set_default_option(capi.OPT_RANK_LAYOUT, capi.LAYOUT_AUTO)


def _run_indexed_checks(S, rb):
    """every query of the run-indexed layout against the oracle (rb is closed at the end)"""
    o = orc.Oracle.from_runs(S.heads, S.lens, S.ssa, S.esa)
    reads = S.sample_reads(3000, 60, seed=5, sub_rate=0.15, ragged=True)
    reads += [b"", b"A", b"N", b"ACGTN", b"NACGT", b"ACNGT", b"AC", b"ACG", b"acgt", bytes([1]), bytes([255]) * 3, bytes([0]),
              S.text[:500].tobytes(), S.text[:501].tobytes(), b"A" + bytes([1]), bytes([1]) + b"A",
              S.text[-30:].tobytes(), S.text[-31:-1].tobytes(), S.text[-2:].tobytes()]
    seqs, off = ra.pack_reads(reads)
    lo, hi, k = rb.find_range_w_toehold(seqs, off)
    lo1, hi1 = rb.find_range(seqs, off)
    wlo, whi, wk = o.find_range_w_toehold_batch(seqs, off)
    assert (lo == wlo).all() and (hi == whi).all() and (k == wk).all()
    assert (lo1 == wlo).all() and (hi1 == whi).all()
    for max_hits in (MAXU, 1, 3, 0):
        loc_off, locs = rb.locs_at(lo, hi, k, max_hits)
        woff, wlocs = o.locs_at_batch(wlo, whi, wk, max_hits)
        assert (loc_off == woff).all() and (locs == wlocs).all()
    # batches that do not fill a wave, and a single read
    for cnt in (1, 63, 65, 130):
        s2, o2 = ra.pack_reads(reads[:cnt])
        l2, h2, k2 = rb.find_range_w_toehold(s2, o2)
        assert (l2 == wlo[:cnt]).all() and (h2 == whi[:cnt]).all() and (k2 == wk[:cnt]).all()
    # the kernels that are not on the rb_align path answer their ranks lane by lane there
    rng = np.random.default_rng(3)
    rows = rng.integers(0, S.n, 500).astype(np.uint64)
    his = np.minimum(rows + rng.integers(0, 50, 500).astype(np.uint64), np.uint64(S.n - 1))
    cs = rng.choice(np.frombuffer(b"ACGT\x01N", dtype=np.uint8), 500)
    nlo, nhi = rb.LF(rows, his, cs)
    for j in range(500):
        assert (int(nlo[j]), int(nhi[j])) == o.LF(int(rows[j]), int(his[j]), int(cs[j]))
    ms, me, mo, mv = S.markers(wsize=10)
    rb.set_markers(ms, me, mo, mv)
    o.set_markers(ms, me, mo, mv)
    nseed, nmk = _check_marker_seeds(rb, o, reads[:300] + reads[-19:], 10, 1000)
    assert nseed > 330 and nmk > 20
    _check_marker_seeds(rb, o, reads[:120], 19, 4)
    _check_marker_seeds(rb, o, reads[:60], 10, 1000, ftab_k=3)      # (rb_markers --ftab: that mode's kernel answers lane by lane)
    goff, glocs = rb.find_locs_greedy_seeding(*ra.pack_reads(reads[:200] + reads[-19:]), 10)
    for i, q in enumerate(reads[:200] + reads[-19:]):
        assert glocs[int(goff[i]):int(goff[i + 1])].tolist() == o.greedy_locate(q, 10)[0]
    # find_range_w_markers (rowbowt.hpp:292-339): the windowed search, cooperative on this layout as well
    sub = reads[:400] + reads[-19:]
    s3, o3 = ra.pack_reads(sub)
    for wsize, max_range in ((10, MAXU), (7, 4), (25, 1000), (51, MAXU)):
        lo3, hi3, mk_off3, mk3 = rb.find_range_w_markers(s3, o3, wsize, max_range)
        got3 = split(mk_off3, mk3)
        for i, q in enumerate(sub):
            (wl, wh), wm = o.find_range_w_markers(q, wsize, max_range)
            assert (int(lo3[i]), int(hi3[i])) == (wl, wh) and got3[i] == wm, (i, q, wsize)
    rb.close()
    o.close()
