"""GPU parity (through the C-ABI, bit-exact against the oracle; needs an MI355X): markers, marker seeds, greedy seeds (next-row f4) on both layouts."""
import json
import os
import re
import sys

import numpy as np
import pytest

import golden_values as G
import orc
import rowbowt_amd as ra
from rowbowt_amd.shard import shard_bounds
from rowbowt_amd import capi
from synth import SynthIndex
from gpu_common import *  # noqa: F401,F403  (helpers shared by the GPU parity files)

pytestmark = pytest.mark.gpu
MAXU = G.MAXU
ALL = ra.LoadRbwtFlag.SA | ra.LoadRbwtFlag.MA


def test_synth_markers(synth):
    S = synth
    rb = ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0)
    o = orc.Oracle.from_runs(S.heads, S.lens, S.ssa, S.esa)
    ms, me, mo, mv = S.markers(wsize=10)
    assert len(ms) > 50 and int(np.diff(mo).max()) >= 1
    rb.set_markers(ms, me, mo, mv)
    o.set_markers(ms, me, mo, mv)
    reads = S.sample_reads(1500, 50, seed=9, sub_rate=0.1) + [b"ACGT", b""]
    seqs, off = ra.pack_reads(reads)
    lo, hi = rb.find_range(seqs, off)
    mk_off, mk = rb.markers_at(lo, hi)  # rb_align -m (rb_align.cpp:138)
    got = split(mk_off, mk)
    n_nonempty = 0
    for i, q in enumerate(reads):
        want = o.markers_at(int(lo[i]), int(hi[i]))
        assert got[i] == want
        n_nonempty += bool(want)
    assert n_nonempty > 100
    for wsize, max_range in ((10, MAXU), (7, MAXU), (10, 4), (25, 1000), (50, MAXU), (51, MAXU)):
        lo2, hi2, mk_off2, mk2 = rb.find_range_w_markers(seqs, off, wsize, max_range)
        got2 = split(mk_off2, mk2)
        for i, q in enumerate(reads):
            (wl, wh), wm = o.find_range_w_markers(q, wsize, max_range)
            assert (int(lo2[i]), int(hi2[i])) == (wl, wh)
            assert got2[i] == wm
    rb.close()
    o.close()


@pytest.mark.parametrize("shape", ["sparse", "crowded", "wide", "mixed"])
def test_marker_bucket_records_against_the_arrays(synth, shape, monkeypatch):
    """The marker directory's 32-byte bucket records (round 6; rbg_dev.h MkRec, rbg_device.hpp marker_query): MarkerArray::at_range(lo, hi) as rb_align -m and the
    marker seeds ask it (rowbowt.hpp:272-290, :437-441) answered from the records of the buckets of lo and hi.  Against plain arithmetic on the arrays (the values of
    every run with start <= hi && end >= lo, in run order) and against the library with RBG_MK_REC=0, on marker arrays the fixtures do not have: one-row runs packed so
    that buckets overflow their three listed runs, runs wider than many buckets, runs of up to nine values, ranges that start or end beyond the BWT."""
    S = synth
    rng = np.random.default_rng({"sparse": 1, "crowded": 2, "wide": 3, "mixed": 4}[shape])
    n = S.n
    if shape == "sparse":
        starts = np.sort(rng.choice(n - 4, size=300, replace=False))
        starts = starts[np.concatenate([[True], np.diff(starts) > 3])]
        ends = starts + rng.integers(0, 3, size=len(starts))
    elif shape == "crowded":      # four of five rows start a one-row run: every bucket overflows
        rows = np.arange(0, n - 1)
        starts = rows[rows % 5 != 4]
        ends = starts.copy()
    elif shape == "wide":         # a few runs hundreds of rows wide
        cuts = np.sort(rng.choice(n - 2, size=60, replace=False))
        starts, ends = cuts[0::2][:29], cuts[1::2][:29] - 1
        keep = ends >= starts
        starts, ends = starts[keep], ends[keep]
    else:
        parts, pos = [], 0
        while pos < n - 50:
            w = int(rng.choice([0, 0, 1, 2, 40, 700]))
            parts.append((pos, min(pos + w, n - 1)))
            pos += w + int(rng.choice([1, 1, 2, 9, 300]))
        starts, ends = np.array([a for a, _ in parts]), np.array([b for _, b in parts])
    assert (starts[1:] > ends[:-1]).all() and int(ends[-1]) < n
    per = rng.integers(1, 10 if shape != "crowded" else 3, size=len(starts))
    off = np.concatenate([[0], np.cumsum(per)]).astype(np.uint64)
    vals = (rng.integers(0, 1 << 40, size=int(off[-1]), dtype=np.uint64) | (rng.integers(0, 3, size=int(off[-1])).astype(np.uint64) << np.uint64(60)))
    starts, ends = starts.astype(np.uint64), ends.astype(np.uint64)
    lo = rng.integers(0, n, size=6000).astype(np.uint64)
    width = rng.choice([0, 0, 1, 5, 60, 1500], size=6000).astype(np.uint64)
    hi = np.minimum(lo + width, np.uint64(n + 40))           # (some ranges end beyond the BWT)
    lo[:20] = n + 3                                           # (and some start there: nothing)
    hi[:20] = n + 9
    got = {}
    for rec in ("1", "0"):
        monkeypatch.setenv("RBG_MK_REC", rec)
        with capi.default_option(capi.OPT_RANK_LAYOUT, capi.LAYOUT_RUNS):
            rb = ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0)
        rb.set_markers(starts, ends, off, vals)
        got[rec] = rb.markers_at(lo, hi)
        rb.close()
    assert (got["1"][0] == got["0"][0]).all() and (got["1"][1] == got["0"][1]).all()
    m_off, m = got["1"]
    for i in range(len(lo)):
        f, l = np.searchsorted(ends, lo[i], side="left"), np.searchsorted(starts, hi[i], side="right")
        want = vals[int(off[f]):int(off[l])] if l > f else vals[:0]
        assert (m[int(m_off[i]):int(m_off[i + 1])] == want).all(), (shape, i, int(lo[i]), int(hi[i]))
    assert int(m_off[-1]) > 1000


def test_marker_seeds_small(small, simple_reads, error_reads):
    """get_markers_greedy_seeding (rowbowt.hpp:406-482, no ftab) on the reference's fixture"""
    rb, o = small
    reads = simple_reads + error_reads + [b"", b"A", b"NNNN", b"ACGTNACGT", simple_reads[0] + b"N" + simple_reads[2]]
    for wsize, max_range in ((19, 1000), (5, 1000), (1, MAXU), (0, 10), (10, 2), (21, 1000)):
        nseed, _ = _check_marker_seeds(rb, o, reads, wsize, max_range)
        assert nseed >= len(reads)
    # with an ftab loaded (rowbowt.hpp:430-433, :454-464); reads shorter than K take the documented miss
    for K, wsize in ((4, 5), (6, 19), (10, 9), (1, 3), (12, 19)):
        _check_marker_seeds(rb, o, reads, wsize, 1000, ftab_k=K)


def test_marker_seeds_synth(synth):
    S = synth
    rb = ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0)
    o = orc.Oracle.from_runs(S.heads, S.lens, S.ssa, S.esa)
    reads = S.sample_reads(1200, 60, seed=21, sub_rate=0.5) + [b"ACGT", b""]
    # without a marker array every mbuf is empty, the seeds are the same (rowbowt.hpp:273,283)
    nseed0, nmk0 = _check_marker_seeds(rb, o, reads, 10, MAXU)
    assert nmk0 == 0 and nseed0 > len(reads)
    ms, me, mo, mv = S.markers(wsize=10)
    rb.set_markers(ms, me, mo, mv)
    o.set_markers(ms, me, mo, mv)
    tot = 0
    for wsize, max_range in ((10, MAXU), (7, 1000), (19, 1000), (3, 6), (60, MAXU)):
        nseed, nmk = _check_marker_seeds(rb, o, reads, wsize, max_range)
        assert nseed == nseed0   # seeds do not depend on the windows
        tot += nmk
    assert tot > 1000
    differs = 0
    for K, wsize in ((5, 10), (8, 7), (10, 19), (3, 2)):
        _check_marker_seeds(rb, o, reads, wsize, 1000, ftab_k=K)
        differs += any(o.markers_greedy_seeding(q, wsize, 1000, K) != o.markers_greedy_seeding(q, wsize, 1000) for q in reads[:300])
    assert differs >= 1   # the ftab variant really is a different seeding (k-mer misses restart further left)
    for ks in (1, 2, 3, 4):
        capi.set_default_option(capi.OPT_KMER_STEPS, ks)
        try:
            rb2 = ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0)
            rb2.set_markers(ms, me, mo, mv)
            _check_marker_seeds(rb2, o, reads[:400], 10, 1000)
            _check_marker_seeds(rb2, o, reads[:400], 10, 1000, ftab_k=7)
            rb2.close()
        finally:
            capi.set_default_option(capi.OPT_KMER_STEPS, 5)
    rb.close()
    o.close()


@pytest.mark.parametrize("layout", [capi.LAYOUT_SLOTS, capi.LAYOUT_RUNS])
def test_marker_seeds_logged_fill(synth, layout):
    """rbg_marker_seeds_plan_log_dev / _fill_log_dev (the reads walked once: the plan logs every sequence's seed records
    and marker places, the fill copies) against the two-walk pair rbg_marker_seeds_plan_dev / _fill_dev, which
    _check_marker_seeds pins to the oracle (rowbowt.hpp:406-482): same offsets, records and markers -- with the default
    quota, with a quota of two seeds per sequence (most sequences exceed it and are walked again from the list), with
    the tool's --ftab mode, on both layouts; a log area too small for two seeds per sequence is refused."""
    import torch
    S = synth
    rb = _with_layout(layout, lambda: ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0))
    o = orc.Oracle.from_runs(S.heads, S.lens, S.ssa, S.esa)
    ms, me, mo, mv = S.markers(wsize=10)
    rb.set_markers(ms, me, mo, mv)
    o.set_markers(ms, me, mo, mv)
    reads = S.sample_reads(3000, 90, seed=41, sub_rate=0.3, ragged=True)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    reads += [q[::-1].translate(comp) for q in reads[:1500]] + [b"", b"A", b"NNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNN", S.text[:700].tobytes()]
    _check_marker_seeds(rb, o, reads[:150] + reads[-4:], 10, 1000)           # the host API (it goes through the log too)
    seqs, off = ra.pack_reads(reads)
    N = len(reads)
    dev = torch.device("cuda:0")
    d_seqs = torch.from_numpy(np.concatenate([seqs, np.zeros(16 + (-len(seqs)) % 16, np.uint8)])).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    L = ra.lib()
    tmp_bytes = L.rbg_locate_plan_tmp_bytes(N)
    d_tmp = torch.empty(tmp_bytes, dtype=torch.uint8, device=dev)

    def run(wsize, max_range, ftab_k, log_bytes):
        d_soff, d_moff = (torch.full((N + 1,), -1, dtype=torch.int64, device=dev) for _ in range(2))
        d_log = torch.empty(max(log_bytes, 16), dtype=torch.uint8, device=dev) if log_bytes is not None else None
        if d_log is None:
            assert L.rbg_marker_seeds_plan_dev(rb.h, d_seqs.data_ptr(), d_off.data_ptr(), N, wsize, max_range, ftab_k, d_soff.data_ptr(),
                                               d_moff.data_ptr(), d_tmp.data_ptr(), tmp_bytes, st) == 0
        else:
            assert L.rbg_marker_seeds_plan_log_dev(rb.h, d_seqs.data_ptr(), d_off.data_ptr(), N, wsize, max_range, ftab_k, d_soff.data_ptr(),
                                                   d_moff.data_ptr(), d_tmp.data_ptr(), tmp_bytes, d_log.data_ptr(), log_bytes, st) == 0
        ns, nm = int(d_soff[-1].item()), int(d_moff[-1].item())
        d_rec = torch.full((max(ns, 1) * 6,), -1, dtype=torch.int64, device=dev)
        d_mk = torch.full((max(nm, 1),), -1, dtype=torch.int64, device=dev)
        if d_log is None:
            assert L.rbg_marker_seeds_fill_dev(rb.h, d_seqs.data_ptr(), d_off.data_ptr(), N, wsize, max_range, ftab_k, d_soff.data_ptr(),
                                               d_moff.data_ptr(), d_rec.data_ptr(), d_mk.data_ptr(), st) == 0
        else:
            assert L.rbg_marker_seeds_fill_log_dev(rb.h, d_seqs.data_ptr(), d_off.data_ptr(), N, wsize, max_range, ftab_k, d_soff.data_ptr(),
                                                   d_moff.data_ptr(), d_rec.data_ptr(), d_mk.data_ptr(), d_log.data_ptr(), log_bytes, st) == 0
        torch.cuda.synchronize()
        over = None
        if d_log is not None:   # the list of sequences over quota sits behind the per-sequence area: its length is the log's last aligned block
            over = True
        return d_soff.cpu().numpy(), d_moff.cpu().numpy(), d_rec[:ns * 6].cpu().numpy(), d_mk[:nm].cpu().numpy()

    for wsize, max_range, ftab_k in ((10, 1000, 0), (19, 6, 0), (10, 1000, 3)):
        want = run(wsize, max_range, ftab_k, None)
        assert want[2].size > 6 * N and (want[2] != -1).all()
        for q in (0, 2, 5, 40):
            lb = int(L.rbg_marker_seeds_log_bytes(rb.h, N, q))
            got = run(wsize, max_range, ftab_k, lb)
            for a, b in zip(want, got):
                assert (a == b).all(), (wsize, max_range, ftab_k, q)
    # an area that cannot hold two seeds per sequence, or an unaligned one, is refused
    d_soff, d_moff = (torch.empty(N + 1, dtype=torch.int64, device=dev) for _ in range(2))
    d_small = torch.empty(N * 40, dtype=torch.uint8, device=dev)
    assert L.rbg_marker_seeds_plan_log_dev(rb.h, d_seqs.data_ptr(), d_off.data_ptr(), N, 10, 1000, 0, d_soff.data_ptr(), d_moff.data_ptr(),
                                           d_tmp.data_ptr(), tmp_bytes, d_small.data_ptr(), N * 40, st) == -4
    lb = int(L.rbg_marker_seeds_log_bytes(rb.h, N, 0))
    d_big = torch.empty(lb + 16, dtype=torch.uint8, device=dev)
    assert L.rbg_marker_seeds_plan_log_dev(rb.h, d_seqs.data_ptr(), d_off.data_ptr(), N, 10, 1000, 0, d_soff.data_ptr(), d_moff.data_ptr(),
                                           d_tmp.data_ptr(), tmp_bytes, d_big.data_ptr() + 4, lb, st) == -4
    rb.close()
    o.close()


# ---- next-row f4: greedy seeding (rowbowt.hpp:222-256, :633-685) ---------------------------------
def test_greedy_seeding_golden(small, error_reads):
    rb, o = small
    seqs, off = ra.pack_reads(error_reads)
    loc_off, locs = rb.find_locs_greedy_seeding(seqs, off, 10)  # rb_tests.cpp:73,80
    got = split(loc_off, locs)
    for g, want in zip(got, G.GREEDY_LOCS_PREFIX):  # rb_tests.cpp:83-95
        if want is None:
            assert g == []
        else:
            assert g[: len(want)] == want
    for i, q in enumerate(error_reads):
        assert got[i] == o.greedy_locate(q, 10)[0]


@pytest.mark.parametrize("ksteps", [3, 1])
def test_greedy_seeding_vs_oracle(synth, ksteps):
    S = synth
    ra.set_default_option(capi.OPT_KMER_STEPS, ksteps)
    try:
        rb = ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0)
    finally:
        ra.set_default_option(capi.OPT_KMER_STEPS, DEFAULT_KMER_STEPS)
    o = orc.Oracle.from_runs(S.heads, S.lens, S.ssa, S.esa)
    rng = np.random.default_rng(17)
    reads = []
    for r in S.sample_reads(5000, 80, seed=33, sub_rate=0.0):  # > 4096: the host path orders the phi chains
        r = bytearray(r)
        for _ in range(int(rng.integers(0, 4))):  # 0..3 substitutions -> several seeds per read
            p = int(rng.integers(len(r)))
            r[p] = int(rng.choice([c for c in b"ACGTN" if c != r[p]]))
        reads.append(bytes(r))
    reads += [b"", b"A", b"N", b"NNNN", b"ACGT" * 5, S.text[:64].tobytes()]
    seqs, off = ra.pack_reads(reads)
    for min_length, max_hits in ((10, MAXU), (1, MAXU), (0, MAXU), (25, 2), (81, MAXU)):
        lo, hi, qs, qe, k = rb.greedy_longest_seed(seqs, off, min_length)
        loc_off, locs = rb.find_locs_greedy_seeding(seqs, off, min_length, max_hits)
        got = split(loc_off, locs)
        nseeds = 0
        for i, q in enumerate(reads):
            wlocs, seed = o.greedy_locate(q, min_length, max_hits)
            assert got[i] == wlocs, (i, min_length)
            if wlocs or seed[1] >= seed[0] and seed[3] > seed[2]:
                assert (int(lo[i]), int(hi[i]), int(qs[i]), int(qe[i]), int(k[i])) == seed
                nseeds += 1
        if min_length <= 25:
            assert nseeds > 3000
    rb.close()
    o.close()


def test_seed_walks_with_staged_and_unstaged_reads(synth):
    """The seeding kernels of the run-indexed layout stage their waves' reads as 2-bit codes in LDS like K2 (round 6); a wave with a read longer than 256 symbols or
    with a symbol outside the k-mer alphabet walks bytes.  Greedy seeds (get_seeds_greedy_w_sample reduced by locate_from_longest_seed, rowbowt.hpp:222-256, :669-677)
    and marker seeds (get_markers_greedy_seeding, :406-482) against the oracle on batches that mix both kinds wave by wave and inside a wave."""
    S = synth
    rng = np.random.default_rng(23)
    with capi.default_option(capi.OPT_RANK_LAYOUT, capi.LAYOUT_RUNS):
        rb = ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0)
    rb.set_markers(*S.markers(10))
    o = orc.Oracle.from_runs(S.heads, S.lens, S.ssa, S.esa)
    o.set_markers(*S.markers(10))
    unit = S.L + S.pad

    def piece(m, nsub):
        h = int(rng.integers(S.H))
        s0 = h * unit + int(rng.integers(0, S.L - m + 1))
        r = bytearray(S.text[s0:s0 + m].tobytes())
        for _ in range(nsub):
            q = int(rng.integers(m))
            r[q] = int(rng.choice([c for c in b"ACGT" if c != r[q]]))
        return r
    reads = []
    for _ in range(2):
        reads += [bytes(piece(100, int(rng.integers(0, 3)))) for _ in range(64)]                 # a staged wave
        reads += [bytes(piece(300, int(rng.integers(0, 4)))) for _ in range(64)]                 # a wave of long reads: bytes
        w = [piece(int(rng.choice([40, 100, 256])), int(rng.integers(0, 3))) for _ in range(64)]   # a staged wave but for one N
        w[int(rng.integers(64))][7] = ord("N")
        reads += [bytes(x) for x in w]
        w = [piece(100, 1) for _ in range(63)] + [piece(257, 2)]                                  # 63 short reads and one just over the cap
        reads += [bytes(x) for x in w]
    reads += [b"", b"A", b"N", bytes(piece(256, 0)), bytes(piece(19, 0)), bytes(piece(20, 1))]
    nseed, nmk = _check_marker_seeds(rb, o, reads, 19, 1000)
    assert nseed >= len(reads) and nmk > 100
    nseed10, _ = _check_marker_seeds(rb, o, reads[:200], 10, 1000)
    assert nseed10 >= 200
    seqs, off = ra.pack_reads(reads)
    loc_off, locs = rb.find_locs_greedy_seeding(seqs, off, 20)
    got = split(loc_off, locs)
    for i, q in enumerate(reads):
        assert got[i] == o.greedy_locate(q, 20)[0], (i, len(q))
    rb.close()
    o.close()
