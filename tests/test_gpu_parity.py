"""Parity tests proper: the HIP path (through the C-ABI) against the oracle and the reference's
golden values.  Bit-exact: everything on this path is unsigned 64-bit integer work.  Need an MI355X."""
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

pytestmark = pytest.mark.gpu
MAXU = G.MAXU
ALL = ra.LoadRbwtFlag.SA | ra.LoadRbwtFlag.MA


@pytest.fixture(scope="module")
def small(data_dir):
    rb = ra.load_rowbowt(os.path.join(data_dir, "small.fa"), ALL, device=0)
    o = orc.Oracle.load(os.path.join(data_dir, "small.fa"), orc.SA | orc.MA)
    yield rb, o
    rb.close()
    o.close()


@pytest.fixture(scope="module")
def simple_reads(data_dir):
    return orc.read_fastx(os.path.join(data_dir, "simple_query.fq"))[1]


@pytest.fixture(scope="module")
def error_reads(data_dir):
    return orc.read_fastx(os.path.join(data_dir, "error_query.fq"))[1]


def split(off, vals):
    return [vals[int(off[i]):int(off[i + 1])].tolist() for i in range(len(off) - 1)]


# ---- the reference's own golden vectors, through the HIP path ------------------------------------
def test_golden_count(small, simple_reads):
    rb, _ = small
    seqs, off = ra.pack_reads(simple_reads)
    lo, hi = rb.find_range(seqs, off)
    assert list(zip(lo.tolist(), hi.tolist())) == G.SIMPLE_RANGES  # rb_tests.cpp:115-120
    assert rb.count(seqs, off).tolist() == [h - l + 1 for l, h in G.SIMPLE_RANGES]


def test_golden_kmers(small):
    rb, _ = small
    qs = list(G.KMER_RANGES)
    lo, hi = rb.find_range(*ra.pack_reads(qs))
    assert list(zip(lo.tolist(), hi.tolist())) == [G.KMER_RANGES[q] for q in qs]  # rb_tests.cpp:147-173


def test_golden_locate(small, simple_reads):
    rb, _ = small
    seqs, off = ra.pack_reads(simple_reads)
    lo, hi, k = rb.find_range_w_toehold(seqs, off)
    assert list(zip(lo.tolist(), hi.tolist())) == G.SIMPLE_RANGES
    loc_off, locs = rb.locs_at(lo, hi, k, MAXU)
    assert locs.tolist() == G.SIMPLE_ALL_LOCS  # rb_tests.cpp:47-58
    assert split(loc_off, locs) == G.SIMPLE_LOCS_PER_READ


def test_golden_markers(small, simple_reads):
    rb, _ = small
    seqs, off = ra.pack_reads(simple_reads)
    lo, hi, mk_off, mk = rb.find_range_w_markers(seqs, off, 10, MAXU)  # rb_tests.cpp:126
    assert list(zip(lo.tolist(), hi.tolist())) == G.SIMPLE_RANGES
    for got, want in zip(split(mk_off, mk), G.SIMPLE_FIRST_MARKER):  # rb_tests.cpp:131-140
        if want is None:
            assert got == []
        else:
            assert (G.get_pos(got[0]), G.get_allele(got[0])) == want


def test_golden_files_through_the_hip_path(small, simple_reads, error_reads, data_dir, tmp_path):
    """tests/golden/ (oracle-made, committed) against the HIP path, no oracle in the loop"""
    import json
    rb, _ = small
    gd = G.GOLDEN_DIR
    # get_markers_greedy_seeding records
    reads = simple_reads + error_reads
    seqs, off = ra.pack_reads(reads)
    for case in json.load(open(os.path.join(gd, "toy_marker_seeds.json")))["cases"]:
        seed_off, seeds, mk = rb.get_markers_greedy_seeding(seqs, off, case["wsize"], case["max_range"], case["ftab_k"])
        assert len(case["reads"]) == len(reads)
        for i, want in enumerate(case["reads"]):
            got = seeds[int(seed_off[i]):int(seed_off[i + 1])]
            assert [[int(g[0]), int(g[1]), int(g[2]), int(g[3]), mk[int(g[4]):int(g[5])].tolist()] for g in got] == want["seeds"], (case, i)
    # locate on error_query.fq
    lo, hi, k = rb.find_range_w_toehold(*ra.pack_reads(error_reads))
    loc_off, locs = rb.locs_at(lo, hi, k)
    for i, want in enumerate(json.load(open(os.path.join(gd, "toy_error_query_locate.json")))["reads"]):
        assert (int(lo[i]), int(hi[i]), int(k[i])) == (want["lo"], want["hi"], want["toehold"])
        assert locs[int(loc_off[i]):int(loc_off[i + 1])].tolist() == want["locs"]
    # the reference-format ftab
    rb.write_ftab(4, str(tmp_path / "k4.ftab"))
    assert (tmp_path / "k4.ftab").read_bytes() == open(os.path.join(gd, "toy_k4.ftab"), "rb").read()
    assert rb.check_ftab(os.path.join(gd, "toy_k4.ftab")) == 4
    # rb_markers stdout
    both = tmp_path / "both.fq"
    both.write_bytes(open(os.path.join(data_dir, "simple_query.fq"), "rb").read() + open(os.path.join(data_dir, "error_query.fq"), "rb").read())
    rc, out, err = _run_rb_markers([os.path.join(data_dir, "small.fa"), str(both)])
    assert rc == 0 and out == open(os.path.join(gd, "toy_rb_markers_default.txt")).read(), err
    rc, out, err = _run_rb_markers(["--heuristic", "--best-strand-only", "-y", "5", "-l", "20", "-w", "8", os.path.join(data_dir, "small.fa"), str(both)])
    assert rc == 0 and out == open(os.path.join(gd, "toy_rb_markers_heuristic.txt")).read(), err


def test_error_reads(small, error_reads):
    rb, o = small
    seqs, off = ra.pack_reads(error_reads)
    lo, hi, k = rb.find_range_w_toehold(seqs, off)
    want = [o.find_range_w_toehold(q) for q in error_reads]
    assert list(zip(lo.tolist(), hi.tolist(), k.tolist())) == want
    assert want[0] == (1, 0, 0)


# ---- config 1 of BASELINE.json: toy index, 10k synthetic 100 bp reads ---------------------------
def test_config1_toy_10k(small):
    import naive
    from test_oracle_vs_naive import sample_reads
    rb, o = small
    heads, lens = o.runs()
    text = naive.invert_bwt(naive.expand_bwt(heads, lens))
    rng = np.random.default_rng(20240231)
    reads = sample_reads(text, 10000, 100, rng, spans=[(0, 10000), (10010, 20010), (20020, 30020)])
    seqs, off = ra.pack_reads(reads)
    rb.counters_reset()
    lo, hi, k = rb.find_range_w_toehold(seqs, off)
    wlo, whi, wk = o.find_range_w_toehold_batch(seqs, off, nthreads=4)
    assert (lo == wlo).all() and (hi == whi).all() and (k == wk).all()
    lo2, hi2 = rb.find_range(seqs, off)
    assert (lo2 == wlo).all() and (hi2 == whi).all()
    loc_off, locs = rb.locs_at(lo, hi, k)
    woff, wlocs = o.locs_at_batch(wlo, whi, wk, nthreads=4)
    assert (loc_off == woff).all() and (locs == wlocs).all()
    # counters: {reads, matched, sum occ, sum locs}; two find_range passes + one locate
    occ = np.where(whi >= wlo, whi - wlo + 1, 0)
    c = rb.counters()
    assert c.tolist() == [20000, 2 * int((whi >= wlo).sum()), 2 * int(occ.sum()), int(occ.sum())]


# ---- synthetic pangenomes: both position widths, several bucket shifts, ragged reads ------------
@pytest.fixture(scope="module")
def synth():
    return SynthIndex(L=4000, H=8, n_sites=60, seed=11)


@pytest.mark.parametrize("pos_bytes,rshift,pshift,ksteps",
                         [(0, -1, -1, 5), (8, 3, 3, 5), (0, 9, -1, 5), (0, 12, 4, 4), (8, 10, -1, 2), (4, 11, 8, 1), (0, -1, -1, 4), (0, -1, -1, 3), (0, -1, -1, 2), (0, -1, -1, 1), (8, -1, -1, 4), (4, 0, 0, 2),
                          (8, 3, 2, 1), (4, 8, 8, 3), (8, 8, 7, 4), (4, 5, 6, 1), (4, 2, 2, 4)])
@pytest.mark.parametrize("packed", [0, 2])
def test_synth_all_paths(synth, pos_bytes, rshift, pshift, ksteps, packed, request):
    S = synth
    ra.set_default_option(capi.OPT_PACKED_READS, packed)   # byte kernels / 2-bit packed reads: same answers
    request.addfinalizer(lambda: ra.set_default_option(capi.OPT_PACKED_READS, 1))
    ra.set_default_option(capi.OPT_POS_BYTES, pos_bytes)
    ra.set_default_option(capi.OPT_RANK_BUCKET_SHIFT, rshift)
    ra.set_default_option(capi.OPT_PHI_BUCKET_SHIFT, pshift)
    ra.set_default_option(capi.OPT_KMER_STEPS, ksteps)
    try:
        rb = ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0)
    finally:
        for o_ in (capi.OPT_POS_BYTES, capi.OPT_RANK_BUCKET_SHIFT, capi.OPT_PHI_BUCKET_SHIFT):
            ra.set_default_option(o_, 0 if o_ == capi.OPT_POS_BYTES else -1)
        ra.set_default_option(capi.OPT_KMER_STEPS, 5)
    o = orc.Oracle.from_runs(S.heads, S.lens, S.ssa, S.esa)
    assert rb.info().pos_bytes == (pos_bytes or 4)
    assert rb.info().kmer_steps == ksteps and rb.info().kmer_symbols == (4 if ksteps > 1 else 0)
    assert (rb.info().quad_runs > 0) == (ksteps >= 4) and (rb.info().quint_runs > 0) == (ksteps == 5)
    reads = S.sample_reads(3000, 60, seed=5, sub_rate=0.15, ragged=True)
    reads += [b"", b"A", b"N", b"ACGTN", b"NACGT", b"ACNGT", b"AC", b"ACG", b"acgt", bytes([1]), bytes([255]) * 3, bytes([0]),
              S.text[:500].tobytes(), S.text[:501].tobytes(), b"A" + bytes([1]), bytes([1]) + b"A",
              S.text[-30:].tobytes(), S.text[-31:-1].tobytes(), S.text[-2:].tobytes()]
    seqs, off = ra.pack_reads(reads)
    lo, hi, k = rb.find_range_w_toehold(seqs, off)
    wlo, whi, wk = o.find_range_w_toehold_batch(seqs, off)
    assert (lo == wlo).all() and (hi == whi).all() and (k == wk).all()
    # independent check against the explicit-text FM index too
    for i in range(0, len(reads), 37):
        assert (int(lo[i]), int(hi[i])) == S.fm.find_range(reads[i])
    for max_hits in (MAXU, 1, 3, 0):
        loc_off, locs = rb.locs_at(lo, hi, k, max_hits)
        woff, wlocs = o.locs_at_batch(wlo, whi, wk, max_hits)
        assert (loc_off == woff).all() and (locs == wlocs).all()
    rb.close()
    o.close()


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
    rb = _with_layout(layout, 48, lambda: ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0))
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


@pytest.mark.parametrize("pos_bytes", [4, 8])
def test_device_compose_matches_host_compose(synth, pos_bytes):
    """The k-mer tables of DESIGN.md 2b are composed on the device at load time (k_compose.hip: merges, sorts and scans
    over the run lists); rbg_host.cpp compose() is the same statement as serial host code (RBG_HOST_COMPOSE=1).  Same
    tables -- runs per level, replica size -- and the same answers, toeholds of k-mer steps included (nested LF_w_loc,
    rowbowt.hpp:555-573), on both layouts; a reference-built index (tests/data) likewise."""
    S = synth
    reads = S.sample_reads(4000, 90, seed=77, sub_rate=0.12, ragged=True) + [b"", b"ACGTN", S.text[:400].tobytes()]
    seqs, off = ra.pack_reads(reads)
    o = orc.Oracle.from_runs(S.heads, S.lens, S.ssa, S.esa)
    wlo, whi, wk = o.find_range_w_toehold_batch(seqs, off)
    woff, wlocs = o.locs_at_batch(wlo, whi, wk)
    for layout in (capi.LAYOUT_SLOTS, capi.LAYOUT_RUNS):
        infos = []
        for host in ("1", None):
            if host:
                os.environ["RBG_HOST_COMPOSE"] = host
            try:
                with capi.default_option(capi.OPT_POS_BYTES, pos_bytes):
                    rb = _with_layout(layout, 48, lambda: ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0))
            finally:
                os.environ.pop("RBG_HOST_COMPOSE", None)
            i = rb.info()
            infos.append((i.kmer_steps, i.pair_runs, i.triple_runs, i.quad_runs, i.quint_runs, i.rank_slots_overflow))
            assert i.kmer_steps == 5 and i.pos_bytes == pos_bytes
            lo, hi, k = rb.find_range_w_toehold(seqs, off)
            assert (lo == wlo).all() and (hi == whi).all() and (k == wk).all()
            loc_off, locs = rb.locs_at(lo, hi, k)
            assert (loc_off == woff).all() and (locs == wlocs).all()
            for ks in (2, 3, 4):   # fewer levels asked for: only those are composed
                with capi.default_option(capi.OPT_KMER_STEPS, ks), capi.default_option(capi.OPT_POS_BYTES, pos_bytes):
                    if host:
                        os.environ["RBG_HOST_COMPOSE"] = host
                    try:
                        rb2 = _with_layout(layout, 48, lambda: ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0))
                    finally:
                        os.environ.pop("RBG_HOST_COMPOSE", None)
                assert rb2.info().kmer_steps == ks
                lo2, hi2, k2 = rb2.find_range_w_toehold(seqs, off)
                assert (lo2 == wlo).all() and (hi2 == whi).all() and (k2 == wk).all()
                rb2.close()
            rb.close()
        assert infos[0] == infos[1], infos
    o.close()


@pytest.mark.parametrize("pos_bytes,rshift,ksteps,fk", [(4, -1, 5, -1), (8, -1, 5, -1), (4, 8, 3, 0), (4, 4, 5, 3), (8, 2, 1, -1), (4, 0, 2, -1)])
def test_slots_of_64_bytes(synth, pos_bytes, rshift, ksteps, fk):
    """RBG_OPT_SLOT_BYTES = 64 (rbg_dev.h RankSlot64, k_search64.hip): one 64-byte slot per 4 x 2^shift rows fetched by a
    quad of lanes, fourteen inline runs, ordinal and predecessor sample inline, 4 KB dense tables for crowded buckets (small
    shifts force them) -- same ranges, toeholds (LF_w_loc, rowbowt.hpp:555-573), locations and the same answers from every
    other kernel (they read the 64 bytes lane by lane) as the 16-byte slots, i.e. as the oracle."""
    S = synth
    o = orc.Oracle.from_runs(S.heads, S.lens, S.ssa, S.esa)
    with capi.default_option(capi.OPT_SLOT_BYTES, 64), capi.default_option(capi.OPT_POS_BYTES, pos_bytes), \
            capi.default_option(capi.OPT_RANK_BUCKET_SHIFT, rshift), capi.default_option(capi.OPT_KMER_STEPS, ksteps), capi.default_option(capi.OPT_FTAB_K, fk):
        rb = _with_layout(capi.LAYOUT_SLOTS, 48, lambda: ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0))
    i = rb.info()
    assert i.slot_bytes == 64 and i.pos_bytes == pos_bytes and i.kmer_steps == ksteps
    if rshift == 8:
        assert i.rank_slots_overflow > 0          # 1024-row buckets of the single symbols hold more than 14 run starts: dense tables
    reads = S.sample_reads(3000, 70, seed=15, sub_rate=0.15, ragged=True)
    reads += [b"", b"A", b"N", b"ACGTN", b"NACGT", b"AC", b"acgt", bytes([1]), bytes([255]) * 3, S.text[:600].tobytes(), bytes([1]) + b"A",
              S.text[-30:].tobytes(), S.text[-31:-1].tobytes(), S.text[-2:].tobytes()]
    seqs, off = ra.pack_reads(reads)
    wlo, whi, wk = o.find_range_w_toehold_batch(seqs, off)
    for packed in (0, 2):
        with capi.default_option(capi.OPT_PACKED_READS, packed):
            lo, hi, k = rb.find_range_w_toehold(seqs, off)
            lo1, hi1 = rb.find_range(seqs, off)
        assert (lo == wlo).all() and (hi == whi).all() and (k == wk).all() and (lo1 == wlo).all() and (hi1 == whi).all()
    loc_off, locs = rb.locs_at(lo, hi, k)
    woff, wlocs = o.locs_at_batch(wlo, whi, wk)
    assert (loc_off == woff).all() and (locs == wlocs).all()
    for cnt in (1, 3, 63, 65):     # batches that do not fill a quad / a wave
        s2, o2 = ra.pack_reads(reads[:cnt])
        l2, h2, k2 = rb.find_range_w_toehold(s2, o2)
        assert (l2 == wlo[:cnt]).all() and (h2 == whi[:cnt]).all() and (k2 == wk[:cnt]).all()
    rng = np.random.default_rng(3)
    rows = rng.integers(0, S.n, 500).astype(np.uint64)
    his = np.minimum(rows + rng.integers(0, 2000, 500).astype(np.uint64), np.uint64(S.n - 1))
    cs = rng.choice(np.frombuffer(b"ACGT\x01N", dtype=np.uint8), 500)
    nlo, nhi = rb.LF(rows, his, cs)
    for j in range(500):
        assert (int(nlo[j]), int(nhi[j])) == o.LF(int(rows[j]), int(his[j]), int(cs[j]))
    ms, me, mo, mv = S.markers(wsize=10)
    rb.set_markers(ms, me, mo, mv)
    o.set_markers(ms, me, mo, mv)
    _check_marker_seeds(rb, o, reads[:200], 10, 1000)
    goff, glocs = rb.find_locs_greedy_seeding(*ra.pack_reads(reads[:150]), 10)
    for j in range(150):
        assert glocs[int(goff[j]):int(goff[j + 1])].tolist() == o.greedy_locate(reads[j], 10)[0]
    rb.close()
    o.close()


def test_single_LF_steps(small, synth):
    """RowBowt::LF (rowbowt.hpp:74-88) one step at a time, against the oracle's LF."""
    rb, o = small
    rng = np.random.default_rng(5)
    n = 30031
    lo = rng.integers(0, n, 4000).astype(np.uint64)
    hi = np.minimum(lo + rng.integers(0, 200, 4000).astype(np.uint64), np.uint64(n - 1))
    lo[:50] = 0
    hi[:50] = n - 1
    sym = rng.choice(np.frombuffer(b"ACGT\x01N", dtype=np.uint8), 4000)
    nlo, nhi = rb.LF(lo, hi, sym)
    for i in range(4000):
        assert (int(nlo[i]), int(nhi[i])) == o.LF(int(lo[i]), int(hi[i]), int(sym[i]))
    # chaining LF reproduces find_range (rowbowt.hpp:127-129)
    q = b"TATCTCCGCGATCTCCAACT"
    l, h = np.array([0], np.uint64), np.array([n - 1], np.uint64)
    for c in reversed(q):
        l, h = rb.LF(l, h, np.array([c], np.uint8))
    assert (int(l[0]), int(h[0])) == (24279, 24280)


def test_device_resident_api(synth):
    """HBM in / HBM out entry points on torch's current stream (what bench.py times)."""
    import ctypes as C
    import torch
    S = synth
    rb = ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0)
    o = orc.Oracle.from_runs(S.heads, S.lens, S.ssa, S.esa)
    reads = S.sample_reads(5000, 100, seed=3, sub_rate=0.1)
    seqs, off = ra.pack_reads(reads)
    N = len(reads)
    dev = torch.device("cuda:0")
    pad = (-len(seqs)) % 16
    d_seqs = torch.from_numpy(np.concatenate([seqs, np.zeros(pad, np.uint8)])).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    d_lo, d_hi, d_k = (torch.empty(N, dtype=torch.int64, device=dev) for _ in range(3))
    st = torch.cuda.current_stream().cuda_stream
    L = ra.lib()
    assert L.rbg_find_range_w_toehold_dev(rb.h, d_seqs.data_ptr(), d_off.data_ptr(), N, d_lo.data_ptr(),
                                          d_hi.data_ptr(), d_k.data_ptr(), st) == 0
    d_loc_off = torch.empty(N + 1, dtype=torch.int64, device=dev)
    tmp_bytes = L.rbg_locate_plan_tmp_bytes(N)
    d_tmp = torch.empty(tmp_bytes, dtype=torch.uint8, device=dev)
    assert L.rbg_locate_plan_dev(rb.h, d_lo.data_ptr(), d_hi.data_ptr(), N, MAXU, d_loc_off.data_ptr(),
                                 d_tmp.data_ptr(), tmp_bytes, st) == 0
    total = int(d_loc_off[-1].item())
    d_locs = torch.empty(max(total, 1), dtype=torch.int64, device=dev)
    assert L.rbg_locate_fill_dev(rb.h, d_lo.data_ptr(), d_hi.data_ptr(), d_k.data_ptr(), N, MAXU,
                                 d_loc_off.data_ptr(), d_locs.data_ptr(), None, st) == 0
    # same walk with the chains ordered by toehold (locality only: identical output)
    ws_bytes = L.rbg_locate_order_ws_bytes(N)
    d_ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    d_locs2 = torch.full_like(d_locs, -1)
    assert L.rbg_locate_order_dev(rb.h, d_k.data_ptr(), N, d_ws.data_ptr(), ws_bytes, st) == 0
    assert L.rbg_locate_fill_dev(rb.h, d_lo.data_ptr(), d_hi.data_ptr(), d_k.data_ptr(), N, MAXU,
                                 d_loc_off.data_ptr(), d_locs2.data_ptr(), d_ws.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert bool((d_locs[:total] == d_locs2[:total]).all().item())
    assert L.rbg_locate_order_dev(rb.h, d_k.data_ptr(), N, d_ws.data_ptr(), ws_bytes - 1024, st) == -4  # workspace too small
    wlo, whi, wk = o.find_range_w_toehold_batch(seqs, off, nthreads=4)
    woff, wlocs = o.locs_at_batch(wlo, whi, wk, nthreads=4)
    assert (d_lo.cpu().numpy().view(np.uint64) == wlo).all()
    assert (d_hi.cpu().numpy().view(np.uint64) == whi).all()
    assert (d_k.cpu().numpy().view(np.uint64) == wk).all()
    assert (d_loc_off.cpu().numpy().view(np.uint64) == woff).all()
    assert (d_locs.cpu().numpy().view(np.uint64)[:total] == wlocs).all()
    # unaligned read buffer is rejected, not mis-read
    assert L.rbg_find_range_dev(rb.h, d_seqs.data_ptr() + 1, d_off.data_ptr(), N, d_lo.data_ptr(), d_hi.data_ptr(), st) == -4
    rb.close()
    o.close()


@pytest.mark.parametrize("layout", [capi.LAYOUT_SLOTS, capi.LAYOUT_RUNS])
def test_instrumented_kernels_and_32_bit_locations(synth, layout):
    """rbg_find_range_stats_dev / rbg_locate_fill_stats_dev (the instrumented instantiations bench.py prices the kernels
    with): same outputs as the plain kernels on both layouts, sums that add up; rbg_locate_fill_dev32: the low 32 bits of
    rbg_locate_fill_dev's locations (toehold_sa.hpp:37-49 fills 64-bit ones), refused at 8-byte positions; the packed
    (2-bit) search on the run-indexed layout: the cooperative kernel, same answers."""
    import torch
    S = synth
    rb = _with_layout(layout, 48, lambda: ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0))
    assert rb.info().rank_layout == layout
    reads = S.sample_reads(4000, 80, seed=31, sub_rate=0.1, ragged=True) + [b"", b"ACGTN", b"A"]
    seqs, off = ra.pack_reads(reads)
    N = len(reads)
    dev = torch.device("cuda:0")
    d_seqs = torch.from_numpy(np.concatenate([seqs, np.zeros(16 + (-len(seqs)) % 16, np.uint8)])).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    L = ra.lib()
    lo, hi, k = rb.find_range_w_toehold(seqs, off)
    d_lo, d_hi, d_k = (torch.full((N,), -3, dtype=torch.int64, device=dev) for _ in range(3))
    d_stats = torch.zeros(16, dtype=torch.int64, device=dev)
    for toe in (True, False):
        d_stats.zero_(); d_lo.fill_(-3); d_hi.fill_(-3)
        assert L.rbg_find_range_stats_dev(rb.h, d_seqs.data_ptr(), d_off.data_ptr(), N, d_lo.data_ptr(), d_hi.data_ptr(),
                                          d_k.data_ptr() if toe else None, d_stats.data_ptr(), st) == 0
        torch.cuda.synchronize()
        assert (d_lo.cpu().numpy().view(np.uint64) == lo).all() and (d_hi.cpu().numpy().view(np.uint64) == hi).all()
        assert not toe or (d_k.cpu().numpy().view(np.uint64) == k).all()
        sv = d_stats.cpu().numpy().tolist()
        steps, slots, dense, search, ftab, resamp, chunks, symbols = sv[:8]
        total_syms = int(off[-1])
        assert 0 < steps <= symbols <= total_syms and slots >= steps - ftab and 0 < chunks <= total_syms // 16 + 2 * N and ftab <= N
        # every read that matched consumed all of its symbols
        matched = hi >= lo
        lens = (off[1:] - off[:-1]).astype(np.int64)
        assert symbols >= int(lens[matched].sum())
        assert (resamp > 0) == toe or resamp == 0
        if layout == capi.LAYOUT_RUNS and sum(rb.layout_info().rec_bytes) == 0:
            assert dense >= 2 * steps - N            # at least two entries per probe (one probe per step when lo and hi + 1 share it)
        elif layout == capi.LAYOUT_RUNS:             # bucket records (the library's choice on an index this small): one or two records per step, entries only for crowded buckets
            assert steps <= slots <= 2 * steps
    # locations: u64, instrumented u64, u32 -- ordered walk
    d_loc_off = torch.empty(N + 1, dtype=torch.int64, device=dev)
    tmp_bytes = L.rbg_locate_plan_tmp_bytes(N)
    d_tmp = torch.empty(tmp_bytes, dtype=torch.uint8, device=dev)
    d_lo.copy_(torch.from_numpy(lo.view(np.int64))); d_hi.copy_(torch.from_numpy(hi.view(np.int64))); d_k.copy_(torch.from_numpy(k.view(np.int64)))
    assert L.rbg_locate_plan_dev(rb.h, d_lo.data_ptr(), d_hi.data_ptr(), N, MAXU, d_loc_off.data_ptr(), d_tmp.data_ptr(), tmp_bytes, st) == 0
    total = int(d_loc_off[-1].item())
    ws_bytes = L.rbg_locate_order_ws_bytes(N)
    d_ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    assert L.rbg_locate_order_dev(rb.h, d_k.data_ptr(), N, d_ws.data_ptr(), ws_bytes, st) == 0
    d_locs, d_locs_s = (torch.full((total + 1,), -1, dtype=torch.int64, device=dev) for _ in range(2))
    d_locs32 = torch.full((total + 3,), -1, dtype=torch.int32, device=dev)
    assert L.rbg_locate_fill_dev(rb.h, d_lo.data_ptr(), d_hi.data_ptr(), d_k.data_ptr(), N, MAXU, d_loc_off.data_ptr(), d_locs.data_ptr(), d_ws.data_ptr(), st) == 0
    d_stats.zero_()
    assert L.rbg_locate_fill_stats_dev(rb.h, d_lo.data_ptr(), d_hi.data_ptr(), d_k.data_ptr(), N, MAXU, d_loc_off.data_ptr(), d_locs_s.data_ptr(),
                                       d_ws.data_ptr(), d_stats.data_ptr(), st) == 0
    for order in (d_ws.data_ptr(), None):
        d_locs32.fill_(-1)
        assert L.rbg_locate_fill_dev32(rb.h, d_lo.data_ptr(), d_hi.data_ptr(), d_k.data_ptr(), N, MAXU, d_loc_off.data_ptr(), d_locs32.data_ptr(), order, st) == 0
        torch.cuda.synchronize()
        assert bool((d_locs32[:total].to(torch.int64) & 0xFFFFFFFF == d_locs[:total] & 0xFFFFFFFF).all().item())
        assert d_locs32[total:].tolist() == [-1, -1, -1]           # nothing written past the end
    assert bool((d_locs_s[:total] == d_locs[:total]).all().item())
    woff, wlocs = rb.locs_at(lo, hi, k)
    assert (d_locs[:total].cpu().numpy().view(np.uint64) == wlocs).all()
    phi_steps, phi_search, chains, nlocs = d_stats.cpu().numpy().tolist()[:4]
    assert nlocs == total and chains == int((hi >= lo).sum()) and phi_steps == total - chains
    # the packed search on this layout (on the run-indexed one: the cooperative kernel with a bit-stream cursor)
    wsb = L.rbg_pack_ws_bytes(N, int(off[-1]))
    d_pws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    assert L.rbg_pack_reads_dev(rb.h, d_seqs.data_ptr(), d_off.data_ptr(), N, int(off[-1]), d_pws.data_ptr(), wsb, st) == 0
    d_lo.fill_(-3); d_hi.fill_(-3); d_k.fill_(-3)
    assert L.rbg_find_range_w_toehold_packed_dev(rb.h, d_pws.data_ptr(), d_seqs.data_ptr(), d_off.data_ptr(), N, int(off[-1]),
                                                 d_lo.data_ptr(), d_hi.data_ptr(), d_k.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert (d_lo.cpu().numpy().view(np.uint64) == lo).all() and (d_hi.cpu().numpy().view(np.uint64) == hi).all() and (d_k.cpu().numpy().view(np.uint64) == k).all()
    d_lo.fill_(-3); d_hi.fill_(-3)
    assert L.rbg_find_range_packed_dev(rb.h, d_pws.data_ptr(), d_seqs.data_ptr(), d_off.data_ptr(), N, int(off[-1]), d_lo.data_ptr(), d_hi.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert (d_lo.cpu().numpy().view(np.uint64) == lo).all() and (d_hi.cpu().numpy().view(np.uint64) == hi).all()
    rb.close()
    # 8-byte positions: 32-bit locations are refused
    with capi.default_option(capi.OPT_POS_BYTES, 8):
        rb8 = _with_layout(layout, 48, lambda: ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0))
    assert L.rbg_locate_fill_dev32(rb8.h, d_lo.data_ptr(), d_hi.data_ptr(), d_k.data_ptr(), N, MAXU, d_loc_off.data_ptr(), d_locs32.data_ptr(), None, st) == -4
    rb8.close()


def test_device_pipeline_is_graph_capturable(synth):
    """The *_dev entry points neither allocate nor synchronise: the whole count+locate step is captured
    into one HIP graph and replayed on new reads in the same buffers."""
    import torch
    S = synth
    rb = ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0)
    o = orc.Oracle.from_runs(S.heads, S.lens, S.ssa, S.esa)
    N, m = 4096, 64
    dev = torch.device("cuda:0")
    L = ra.lib()
    batches = [S.sample_reads(N, m, seed=s_, sub_rate=0.1) for s_ in (31, 32, 33)]
    d_seqs = torch.zeros(N * m + 16, dtype=torch.uint8, device=dev)
    d_off = torch.arange(N + 1, dtype=torch.int64, device=dev) * m
    d_lo, d_hi, d_k = (torch.empty(N, dtype=torch.int64, device=dev) for _ in range(3))
    d_loc_off = torch.empty(N + 1, dtype=torch.int64, device=dev)
    tmp_bytes, ws_bytes = L.rbg_locate_plan_tmp_bytes(N), L.rbg_locate_order_ws_bytes(N)
    d_tmp = torch.empty(tmp_bytes, dtype=torch.uint8, device=dev)
    d_ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    cap = N * 8 * 64   # at most H = 8 haplotype copies (+ chance hits) per read; checked below
    d_locs = torch.empty(cap, dtype=torch.int64, device=dev)

    def step(st):
        assert L.rbg_find_range_w_toehold_dev(rb.h, d_seqs.data_ptr(), d_off.data_ptr(), N, d_lo.data_ptr(), d_hi.data_ptr(), d_k.data_ptr(), st) == 0
        assert L.rbg_locate_plan_dev(rb.h, d_lo.data_ptr(), d_hi.data_ptr(), N, MAXU, d_loc_off.data_ptr(), d_tmp.data_ptr(), tmp_bytes, st) == 0
        assert L.rbg_locate_order_dev(rb.h, d_k.data_ptr(), N, d_ws.data_ptr(), ws_bytes, st) == 0
        assert L.rbg_locate_fill_dev(rb.h, d_lo.data_ptr(), d_hi.data_ptr(), d_k.data_ptr(), N, MAXU, d_loc_off.data_ptr(),
                                     d_locs.data_ptr(), d_ws.data_ptr(), st) == 0

    def load(reads):
        seqs, _ = ra.pack_reads(reads)
        d_seqs[:N * m].copy_(torch.from_numpy(seqs).to(dev))

    load(batches[0])
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):      # warm-up outside capture
        step(side.cuda_stream)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step(torch.cuda.current_stream().cuda_stream)
    for reads in batches[1:]:
        load(reads)
        g.replay()
        torch.cuda.synchronize()
        seqs, off = ra.pack_reads(reads)
        wlo, whi, wk = o.find_range_w_toehold_batch(seqs, off, nthreads=4)
        woff, wlocs = o.locs_at_batch(wlo, whi, wk, nthreads=4)
        assert int(woff[-1]) <= cap
        assert (d_lo.cpu().numpy().view(np.uint64) == wlo).all() and (d_k.cpu().numpy().view(np.uint64) == wk).all()
        assert (d_loc_off.cpu().numpy().view(np.uint64) == woff).all()
        assert (d_locs.cpu().numpy().view(np.uint64)[:int(woff[-1])] == wlocs).all()
    rb.close()
    o.close()


@pytest.mark.parametrize("sigma,skew", [(2, 1.0), (3, 0.5), (4, 2.0), (6, 1.0), (12, 1.5), (40, 1.2), (200, 1.0)])
def test_random_alphabets(sigma, skew):
    """Nothing in the engine is DNA-specific: random repetitive texts over 2..200 symbols (fewer than 4
    'major' symbols, more symbols than the LDS keeps records for, skewed frequencies), every query path
    against the oracle."""
    import naive
    rng = np.random.default_rng(1000 + sigma)
    alphabet = np.sort(rng.choice(np.arange(2, 256), size=sigma, replace=False)).astype(np.uint8)
    p = 1.0 / np.arange(1, sigma + 1) ** skew
    block = rng.choice(alphabet, size=700, p=p / p.sum())
    pieces = []
    for c in range(6):                       # six mutated copies: a repetitive collection
        b = block.copy()
        pos = rng.choice(len(b), size=12, replace=False)
        b[pos] = rng.choice(alphabet, size=12)
        pieces.append(b)
    text = np.concatenate(pieces + [np.array([1], np.uint8)])   # terminator = smallest symbol, unique
    sa = naive.suffix_array(text)
    heads, lens, brk = naive.rle(naive.bwt_from_sa(text, sa))
    ssa, esa = naive.run_samples(sa, brk, len(text))
    rb = ra.RowBowt.from_runs(heads, lens, ssa, esa, device=0)
    o = orc.Oracle.from_runs(heads, lens, ssa, esa)
    assert rb.info().sigma == len(np.unique(text))
    tb = text.tobytes()
    reads = []
    for _ in range(1500):
        a = int(rng.integers(0, len(tb) - 2))
        q = bytearray(tb[a:a + int(rng.integers(1, 60))])
        if rng.random() < 0.3 and q:
            q[int(rng.integers(0, len(q)))] = int(rng.integers(0, 256))   # any byte, present in the text or not
        reads.append(bytes(q))
    reads += [b"", bytes([1]), bytes([0]), bytes([255]), tb[-5:], tb[:80]]
    seqs, off = ra.pack_reads(reads)
    for packed in (0, 2):
        ra.set_default_option(capi.OPT_PACKED_READS, packed)
        try:
            lo, hi, k = rb.find_range_w_toehold(seqs, off)
            lo2, hi2 = rb.find_range(seqs, off)
        finally:
            ra.set_default_option(capi.OPT_PACKED_READS, 1)
        wlo, whi, wk = o.find_range_w_toehold_batch(seqs, off)
        assert (lo == wlo).all() and (hi == whi).all() and (k == wk).all() and (lo2 == wlo).all() and (hi2 == whi).all()
    assert int((hi >= lo).sum()) > 800
    loc_off, locs = rb.locs_at(lo, hi, k)
    woff, wlocs = o.locs_at_batch(wlo, whi, wk)
    assert (loc_off == woff).all() and (locs == wlocs).all()
    goff, glocs = rb.find_locs_greedy_seeding(seqs, off, 8)
    for i in range(0, len(reads), 7):
        assert glocs[int(goff[i]):int(goff[i + 1])].tolist() == o.greedy_locate(reads[i], 8)[0]
    _check_marker_seeds(rb, o, reads[::5], 6, 1000)
    _check_marker_seeds(rb, o, reads[::9], 6, 1000, ftab_k=3)
    rb.close()
    o.close()


@pytest.mark.parametrize("seed", list(range(10)))
def test_random_small_texts_every_layout(seed):
    """Random texts of 2 to 400 symbols over alphabets of 1 to 6 letters (repeats, long runs, no runs at all), indexed
    from a naive suffix array: ranges, toeholds and locations of random reads and of every kind of substring, through the
    slot tables, the run-indexed layout at two k-mer depths and a 16-key top level, and 8-byte positions -- all equal to
    the oracle's.  (Table sizing, clamped searches at slice boundaries, directories with empty buckets, one-run tables.)"""
    import naive
    rng = np.random.default_rng(1000 + seed)
    letters = [b"A", b"AC", b"ACG", b"ACGT", b"ACGTN", b"ACGTNB"][seed % 6]
    n_body = int(rng.integers(1, 400))
    if seed % 3 == 0:   # repetitive: copies of a short unit with a few substitutions
        unit = rng.choice(list(letters), size=int(rng.integers(1, 12))).astype(np.uint8)
        body = np.tile(unit, n_body // len(unit) + 1)[:n_body].copy()
        for _ in range(n_body // 25):
            body[int(rng.integers(0, n_body))] = letters[int(rng.integers(0, len(letters)))]
    else:
        body = rng.choice(list(letters), size=n_body).astype(np.uint8)
    body = body.tobytes()
    text = np.frombuffer(body + bytes([1]), dtype=np.uint8)
    sa = naive.suffix_array(text)
    heads, lens, brk = naive.rle(naive.bwt_from_sa(text, sa))
    ssa, esa = naive.run_samples(sa, brk, len(text))
    o = orc.Oracle.from_runs(heads, lens, ssa, esa)
    alphabet = sorted(set(body)) + [ord("N"), 1]
    reads = [bytes(rng.choice(alphabet, size=int(rng.integers(0, 12))).astype(np.uint8)) for _ in range(200)]
    for _ in range(200):
        a = int(rng.integers(0, len(body)))
        reads.append(body[a:a + int(rng.integers(1, 40))])
    reads += [body, body[:1], body[-1:], body + body[:1], b"", body[1:], body[:-1]]
    seqs, off = ra.pack_reads(reads)
    wlo, whi, wk = o.find_range_w_toehold_batch(seqs, off)
    woff, wlocs = o.locs_at_batch(wlo, whi, wk)
    for layout, ks, top_kb, pb, rec in ((capi.LAYOUT_SLOTS, 5, 48, 0, None), (capi.LAYOUT_RUNS, 5, 48, 0, None), (capi.LAYOUT_RUNS, 2, 0, 8, None),
                                        (capi.LAYOUT_SLOTS, 3, 48, 8, None), (capi.LAYOUT_RUNS, 5, 48, 0, "2"), (capi.LAYOUT_RUNS, 3, 48, 8, "1")):
        ra.set_default_option(capi.OPT_KMER_STEPS, ks)
        ra.set_default_option(capi.OPT_POS_BYTES, pb)
        if rec is not None:
            os.environ["RBG_RANK_REC"] = rec   # bucket records (rbg_dev.h RunRec)
        try:
            rb = _with_layout(layout, top_kb, lambda: ra.RowBowt.from_runs(heads, lens, ssa, esa, device=0))
        finally:
            ra.set_default_option(capi.OPT_KMER_STEPS, 5)
            ra.set_default_option(capi.OPT_POS_BYTES, 0)
            os.environ.pop("RBG_RANK_REC", None)
        lo, hi, k = rb.find_range_w_toehold(seqs, off)
        assert (lo == wlo).all() and (hi == whi).all() and (k == wk).all(), (layout, ks, body[:40])
        lo2, hi2 = rb.find_range(seqs, off)
        assert (lo2 == wlo).all() and (hi2 == whi).all()
        loc_off, locs = rb.locs_at(lo, hi, k)
        assert (loc_off == woff).all() and (locs == wlocs).all(), (layout, ks, body[:40])
        rb.close()
    o.close()


@pytest.mark.parametrize("body", [b"A", b"AAAAAAAAAAAA", b"ACGT", b"ABABABABAB", b"TTTTTTTTCTTTTTTTT", b"ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT" * 8])
def test_tiny_indexes(body):
    """Degenerate texts (two symbols long, one long run, pure repeats): table sizing, the automatic
    shifts and the ftab word length must not assume anything about n"""
    import naive
    text = np.frombuffer(body + bytes([1]), dtype=np.uint8)
    sa = naive.suffix_array(text)
    heads, lens, brk = naive.rle(naive.bwt_from_sa(text, sa))
    ssa, esa = naive.run_samples(sa, brk, len(text))
    rb = ra.RowBowt.from_runs(heads, lens, ssa, esa, device=0)
    o = orc.Oracle.from_runs(heads, lens, ssa, esa)
    alphabet = sorted(set(body)) + [ord("N"), 1]
    rng = np.random.default_rng(len(body))
    reads = [bytes(rng.choice(alphabet, size=int(rng.integers(0, 9))).astype(np.uint8)) for _ in range(300)]
    reads += [body, body[:1], body[-1:], body + body, b"", body[1:], body[:-1]]
    seqs, off = ra.pack_reads(reads)
    for packed in (0, 2):
        ra.set_default_option(capi.OPT_PACKED_READS, packed)
        try:
            lo, hi, k = rb.find_range_w_toehold(seqs, off)
        finally:
            ra.set_default_option(capi.OPT_PACKED_READS, 1)
        wlo, whi, wk = o.find_range_w_toehold_batch(seqs, off)
        assert (lo == wlo).all() and (hi == whi).all() and (k == wk).all()
    loc_off, locs = rb.locs_at(lo, hi, k)
    woff, wlocs = o.locs_at_batch(wlo, whi, wk)
    assert (loc_off == woff).all() and (locs == wlocs).all()
    goff, glocs = rb.find_locs_greedy_seeding(seqs, off, 2)
    for i in range(len(reads)):
        assert glocs[int(goff[i]):int(goff[i + 1])].tolist() == o.greedy_locate(reads[i], 2)[0]
    _check_marker_seeds(rb, o, reads, 2, 1000)
    _check_marker_seeds(rb, o, reads, 3, 1000, ftab_k=2)
    nlo, nhi = rb.LF(np.zeros(len(alphabet), np.uint64), np.full(len(alphabet), len(text) - 1, np.uint64), np.array(alphabet, np.uint8))
    for j, c in enumerate(alphabet):
        assert (int(nlo[j]), int(nhi[j])) == o.LF(0, len(text) - 1, c)
    rb.close()
    rbr = _with_layout(capi.LAYOUT_RUNS, 0, lambda: ra.RowBowt.from_runs(heads, lens, ssa, esa, device=0))
    lo, hi, k = rbr.find_range_w_toehold(seqs, off)
    assert (lo == wlo).all() and (hi == whi).all() and (k == wk).all()
    loc_off, locs = rbr.locs_at(lo, hi, k)
    assert (loc_off == woff).all() and (locs == wlocs).all()
    rbr.close()
    o.close()


def test_packed_reads_device_api(synth):
    """rbg_pack_reads_dev + *_packed_dev against the byte kernels on the same batch: ranges, toeholds and
    the device counters; reads with symbols outside the major alphabet go through the sel list."""
    import torch
    S = synth
    rb = ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0)
    reads = S.sample_reads(6000, 100, seed=13, sub_rate=0.1, ragged=True)
    rng = np.random.default_rng(3)
    for i in range(0, len(reads), 17):   # sprinkle non-major symbols: absent (N), present but minor (terminator 1)
        q = bytearray(reads[i])
        if q:
            q[int(rng.integers(0, len(q)))] = b"N\x01n"[i % 3]
        reads[i] = bytes(q)
    reads += [b"", b"A", b"ACGT" * 40, S.text[:3000].tobytes(), b"", S.text[100:165].tobytes(), S.text[100:164].tobytes(), S.text[100:163].tobytes()]
    seqs, off = ra.pack_reads(reads)
    N, total = len(reads), int(off[-1])
    dev = torch.device("cuda:0")
    d_seqs = torch.from_numpy(np.concatenate([seqs, np.zeros(16 + (-len(seqs)) % 16, np.uint8)])).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    L = ra.lib()
    outs = {}
    for name in ("bytes", "packed"):
        d_lo, d_hi, d_k, d_lo2, d_hi2 = (torch.full((N,), -7, dtype=torch.int64, device=dev) for _ in range(5))
        L.rbg_counters_reset(rb.h)
        if name == "bytes":
            assert L.rbg_find_range_w_toehold_dev(rb.h, d_seqs.data_ptr(), d_off.data_ptr(), N, d_lo.data_ptr(), d_hi.data_ptr(), d_k.data_ptr(), st) == 0
            assert L.rbg_find_range_dev(rb.h, d_seqs.data_ptr(), d_off.data_ptr(), N, d_lo2.data_ptr(), d_hi2.data_ptr(), st) == 0
        else:
            wsb = L.rbg_pack_ws_bytes(N, total)
            d_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            assert L.rbg_pack_reads_dev(rb.h, d_seqs.data_ptr(), d_off.data_ptr(), N, total, d_ws.data_ptr(), wsb - 1, st) == -4
            assert L.rbg_pack_reads_dev(rb.h, d_seqs.data_ptr(), d_off.data_ptr(), N, total, d_ws.data_ptr(), wsb, st) == 0
            assert L.rbg_find_range_w_toehold_packed_dev(rb.h, d_ws.data_ptr(), d_seqs.data_ptr(), d_off.data_ptr(), N, total,
                                                         d_lo.data_ptr(), d_hi.data_ptr(), d_k.data_ptr(), st) == 0
            assert L.rbg_find_range_packed_dev(rb.h, d_ws.data_ptr(), d_seqs.data_ptr(), d_off.data_ptr(), N, total,
                                               d_lo2.data_ptr(), d_hi2.data_ptr(), st) == 0
        torch.cuda.synchronize()
        outs[name] = [t.cpu().numpy() for t in (d_lo, d_hi, d_k, d_lo2, d_hi2)] + [rb.counters()]
    for a, b in zip(outs["bytes"], outs["packed"]):
        assert (a == b).all()
    assert (outs["packed"][0] != -7).all() and int(outs["packed"][5][0]) == 2 * N   # every read answered exactly once per call
    rb.close()


def test_size_independent_properties(synth):
    """Properties that hold at any size (used again at BASELINE sizes by bench.py --check):
    every located position really is an occurrence; occ == number of distinct locations;
    count of a read == count of its range; appending context never widens a range."""
    S = synth
    rb = ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0)
    reads = S.sample_reads(2000, 80, seed=21, sub_rate=0.0)
    seqs, off = ra.pack_reads(reads)
    lo, hi, k = rb.find_range_w_toehold(seqs, off)
    assert (hi >= lo).all()
    loc_off, locs = rb.locs_at(lo, hi, k)
    tb = S.text.tobytes()
    for i, q in enumerate(reads):
        mine = locs[int(loc_off[i]):int(loc_off[i + 1])].tolist()
        assert len(mine) == int(hi[i] - lo[i] + 1) == len(set(mine))
        for p in mine:
            assert tb[p:p + len(q)] == q
    suff = [q[20:] for q in reads]
    slo, shi = rb.find_range(*ra.pack_reads(suff))
    assert ((shi - slo) >= (hi - lo)).all()
    rb.close()


# ---- the rb_align-compatible CLI: byte-exact stdout (reference src/rb_align.cpp:118-145) ---------
def _run_cli(args, env=None):
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rowbowt_amd", "rb_align")
    p = subprocess.run([exe] + args, capture_output=True, timeout=120, env=dict(os.environ, **env) if env else None)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


def test_cli_count_stdout(data_dir, simple_reads):
    rc, out, err = _run_cli([os.path.join(data_dir, "small.fa"), os.path.join(data_dir, "simple_query.fq")])
    assert rc == 0, err
    names = ["r1.ref", "r1.sample0.0", "r2.ref", "r2.sample0.0", "r3.ref", "r3.sample0.0"]
    want = "".join(f"{n} ({lo},{hi}), count={hi - lo + 1}\n" for n, (lo, hi) in zip(names, G.SIMPLE_RANGES))
    assert out == want
    assert len(err.strip().splitlines()[-1].split()) == 2  # "<load_s> <query_s>", rb_align.cpp:192
    # empty ranges print the unsigned wrap of 0-1+1 (rb_align.cpp:122)
    rc, out, _ = _run_cli([os.path.join(data_dir, "small.fa"), os.path.join(data_dir, "error_query.fq")])
    lines = out.splitlines()
    assert rc == 0 and lines[0] == "r1.ref (1,0), count=0" and lines[2] == "r2.ref (27430,27432), count=3"


def test_cli_layout_from_the_environment(data_dir, tmp_path):
    """rb_align keeps the reference's flags; the library's load-time knobs reach it by environment (include/rbg.h):
    RBG_LAYOUT=runs answers from the run-indexed layout, all depths or depths 1 and 4 only -- the same text."""
    import shutil
    for suf in (".rbwt", ".tsa"):
        shutil.copy(os.path.join(data_dir, "small.fa" + suf), tmp_path / ("idx" + suf))
    (tmp_path / "idx.docs").write_text("ref 0\nhap1 10010\nhap2 20020\n")
    args = ["-s", str(tmp_path / "idx"), os.path.join(data_dir, "simple_query.fq")]
    rc0, out0, err0 = _run_cli(args, env={"RBG_VERBOSE": "1"})
    assert rc0 == 0 and "run-indexed layout" not in err0, err0
    for env in ({"RBG_LAYOUT": "runs"}, {"RBG_LAYOUT": "runs", "RBG_RUN_DEPTHS": "0x1F"}, {"RBG_LAYOUT": "runs", "RBG_RUN_DEPTHS": "9", "RBG_FTAB_K": "0"}):
        rc, out, err = _run_cli(args, env=dict(env, RBG_VERBOSE="1"))
        assert rc == 0 and out == out0 and "run-indexed layout" in err, err
        mask = {None: "0x15", "0x1F": "0x1f", "9": "0x9"}[env.get("RBG_RUN_DEPTHS")]
        assert f"k-mer depths with run lists: mask {mask}" in err, err


def test_cli_locs_and_markers_stdout(data_dir, tmp_path, small, simple_reads):
    import gzip
    import shutil
    rb, o = small
    for suf in (".rbwt", ".tsa", ".mab"):
        shutil.copy(os.path.join(data_dir, "small.fa" + suf), tmp_path / ("idx" + suf))
    (tmp_path / "idx.docs").write_text("ref 0\nhap1 10010\nhap2 20020\n")  # no .docs is shipped
    o.set_docs(["ref", "hap1", "hap2"], [0, 10010, 20020])
    fq = tmp_path / "q.fq.gz"  # gz + true FASTQ syntax + descriptions after the name
    with gzip.open(fq, "wt") as f:
        for i, q in enumerate(simple_reads):
            f.write(f"@read{i} some description\n{q.decode()}\n+\n{'~' * len(q)}\n")
    rc, out, err = _run_cli(["-s", "-m", str(tmp_path / "idx"), str(fq)])
    assert rc == 0, err
    want = ""
    for i, q in enumerate(simple_reads):
        lo, hi, k = o.find_range_w_toehold(q)
        want += f"read{i} ({lo},{hi}), count={hi - lo + 1}\n\tlocs: "
        for l in o.locs_at(lo, hi, k):
            name, off = o.resolve_offset(l)
            want += f"{l}/{name}:{off} "
        want += "\n\tmarkers: "
        mk = o.markers_at(lo, hi)
        if not mk:
            want += "no markers (consider building the marker array with a larger window size)"
        for m_ in mk:
            want += f"{G.get_pos(m_)}/{G.get_allele(m_)} "
        want += "\n"
    assert out == want
    assert "20306/hap2:286 286/ref:286" in out
    # (-s -m is made on the device too, markers line included; the host formatter gives the same bytes)
    rc, out_h, _ = _run_cli(["-s", "-m", str(tmp_path / "idx"), str(fq)], env={"RB_ALIGN_HOST_TEXT": "1"})
    assert rc == 0 and out_h == want
    # batching is invisible: one read per GPU batch gives the same bytes
    rc, out1, _ = _run_cli(["-s", "-m", "--batch", "1", str(tmp_path / "idx"), str(fq)])
    assert rc == 0 and out1 == want
    # replicas are invisible too: every batch sharded over three replicas (on this box's one GPU the same device
    # three times; `--gpus G` puts them on G devices), with batches smaller than, equal to and larger than the shards
    for extra in (["--devices", "0,0,0"], ["--devices", "0,0", "--batch", "3"], ["--gpus", "1", "--batch", "2"]):
        rc, outg, err = _run_cli(["-s", "-m"] + extra + [str(tmp_path / "idx"), str(fq)])
        assert rc == 0 and outg == want, err
    # a truncated record ends the run like kseq's -2 without being reported; the reads before it are
    # (rb_align.cpp:176-185: the loop stops at the failing kseq_read)
    part = tmp_path / "part.fq"
    part.write_text("".join(f"@read{i}\n{q.decode()}\n+\n{'~' * len(q)}\n" for i, q in enumerate(simple_reads[:3])) + "@bad\nACGT\n+\n~~\n")
    rc, outp, err = _run_cli([str(tmp_path / "idx"), str(part)])
    assert rc == 1 and "truncated quality string" in err
    assert outp.count("\n") == 3 and "bad" not in outp and outp.startswith("read0 ")
    # missing index -> "bad file", exit(1) (rowbowt_io.hpp:166-169)
    rc, _, err = _run_cli([str(tmp_path / "nope"), str(fq)])
    assert rc == 1 and "bad file" in err
    # truncated quality string -> error like kseq's -2 (rb_align.cpp:183-185)
    bad = tmp_path / "bad.fq"
    bad.write_text("@r\nACGT\n+\n~~\n")
    rc, _, err = _run_cli([str(tmp_path / "idx"), str(bad)])
    assert rc == 1 and "truncated quality string" in err


def test_cli_locs_text_made_on_the_device(data_dir, tmp_path, small, simple_reads, error_reads, synth):
    """`rb_align -s` (no -m): the text comes from rbg_align_text -- locs_at, resolve_offset and the decimals on the device
    (k_text.hip) -- and is byte-identical to the oracle's rendering of rb_report (rb_align.cpp:118-139) and to the host
    formatter (RB_ALIGN_HOST_TEXT=1): reads without a match, names of 1 and of 700 characters (beyond what a workgroup
    stages in LDS), descriptions, one read per batch, three replicas, and a synthetic pangenome whose reads have tens of
    locations in 50 documents, in batches that do not divide the input."""
    import shutil
    rb, o = small
    for suf in (".rbwt", ".tsa"):
        shutil.copy(os.path.join(data_dir, "small.fa" + suf), tmp_path / ("idx" + suf))
    long_doc = "hap2_" + "x" * 90
    (tmp_path / "idx.docs").write_text(f"ref 0\nhap1 10010\n{long_doc} 20020\n")
    o.set_docs(["ref", "hap1", long_doc], [0, 10010, 20020])
    reads = list(simple_reads) + list(error_reads) + [b"ACGT", b"A", simple_reads[0][:30]]
    names = [f"read{i}" for i in range(len(reads))]
    names[1] = "r"
    names[2] = "n" * 700
    names[4] = "q" * 300
    fq = tmp_path / "q.fq"
    fq.write_text("".join(f"@{n} desc {i}\n{q.decode()}\n+\n{'~' * len(q)}\n" for i, (n, q) in enumerate(zip(names, reads))))
    want = ""
    for n, q in zip(names, reads):
        lo, hi, k = o.find_range_w_toehold(q)
        want += f"{n} ({lo},{hi}), count={(hi - lo + 1) % 2**64}\n\tlocs: "
        if lo <= hi:
            for l in o.locs_at(lo, hi, k):
                dn, off = o.resolve_offset(l)
                want += f"{l}/{dn}:{off} "
        want += "\n"
    for extra in ([], ["--batch", "1"], ["--devices", "0,0,0"], ["--devices", "0,0", "--batch", "3"]):
        rc, out, err = _run_cli(["-s"] + extra + [str(tmp_path / "idx"), str(fq)])
        assert rc == 0 and out == want, err
    rc, out_h, err = _run_cli(["-s", str(tmp_path / "idx"), str(fq)], env={"RB_ALIGN_HOST_TEXT": "1"})
    assert rc == 0 and out_h == want, err
    # a pangenome: many locations per read, 50 documents
    S = synth
    unit = len(S.text) // 50
    docs = "".join(f"hap{h} {h * unit}\n" for h in range(50))
    capi.convert_runs(S.heads, S.lens, S.ssa, S.esa, out_path=str(tmp_path / "pg.rbgpu"), docs_text=docs)
    rs = S.sample_reads(5000, 70, seed=31, sub_rate=0.1, ragged=True)
    fq2 = tmp_path / "pg.fq"
    fq2.write_text("".join(f"@pg.{i}/{i % 7}\n{q.decode()}\n+\n{'I' * len(q)}\n" for i, q in enumerate(rs) if len(q)))
    outs = []
    for extra, env in (([], None), (["--batch", "700", "--devices", "0,0"], None), ([], {"RB_ALIGN_HOST_TEXT": "1"})):
        rc, out, err = _run_cli(["-s"] + extra + [str(tmp_path / "pg"), str(fq2)], env=env)
        assert rc == 0, err
        outs.append(out)
    assert outs[0] == outs[2] and outs[1] == outs[2] and outs[0].count("\n") == 2 * sum(1 for q in rs if len(q))
    assert len(outs[0]) > 200 * len(rs)


def test_bench_two_ranks_rehearsal():
    """`bench.py --gpus 2 --rehearse-ranks`: the whole multi-rank path on this box's one GPU -- the GPU-free parent starts two
    ranks, rank 0 derives the BWT and writes the cache file, both load their replica from it, each searches ITS block of the
    global batch, the timing is the max over ranks, the counters are summed (gloo stands in for RCCL, and the line says it is
    no measurement).  n_gpus = the group's size; reads and matches of both ranks arrive in the counters."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-ranks", "--L", "1500000", "--H", "8", "--reads", "150000",
                        "--steps", "2", "--warmup", "1", "--no-space-speed", "--no-markers", "--no-cpu-baseline", "--check-reads", "2000",
                        "--property-reads", "20000"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]   # (gloo announces its connections on stdout; RCCL does not)
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "rehearsal" in d["config"] and d["vs_baseline"] is None
    assert d["config"]["reads_per_gpu"] == 150000
    assert d["parity"]["bit_exact_vs_oracle"] and d["counters"]["reads"] == 2 * 150000 * 2   # (two ranks x the two timed steps)


def test_align_text_markers_line(small, simple_reads, error_reads):
    """RBG_TEXT_MARKERS through the ABI on the reference's fixture: with and without the locations' line, reads with and
    without markers and without a match (rb_align.cpp:118-145)"""
    rb, o = small
    o.set_docs(["ref", "hap1", "hap2"], [0, 10010, 20020])
    rb.set_docs(["ref", "hap1", "hap2"], [0, 10010, 20020])
    reads = list(simple_reads) + list(error_reads) + [b"ACGT", simple_reads[0][:25]]
    seqs, off = ra.pack_reads(reads)
    lo, hi, k = rb.find_range_w_toehold(seqs, off)
    names = [f"q{i}".encode() for i in range(len(reads))]
    for with_locs in (True, False):
        got = rb.align_text(lo, hi, k if with_locs else None, names, markers=True).decode()
        want = ""
        for n, a, b, kk in zip(names, lo, hi, k):
            a, b, kk = int(a), int(b), int(kk)
            want += f"{n.decode()} ({a},{b}), count={(b - a + 1) % 2**64}\n"
            if with_locs:
                want += "\tlocs: "
                if a <= b:
                    for l in o.locs_at(a, b, kk):
                        dn, offs = o.resolve_offset(l)
                        want += f"{l}/{dn}:{offs} "
                want += "\n"
            want += "\tmarkers: "
            mk = o.markers_at(a, b) if a <= b else []
            if not mk:
                want += "no markers (consider building the marker array with a larger window size)"
            for m_ in mk:
                want += f"{G.get_pos(m_)}/{G.get_allele(m_)} "
            want += "\n"
        assert got == want


def test_align_text_through_the_abi(synth):
    """rbg_align_text called directly: max_hits caps the locations per read like locs_at's (rowbowt.hpp:613-621); without the
    document list the call says RBG_ENOTLOADED; texts of several calls may be out at once"""
    S = synth
    rb = ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0)
    o = orc.Oracle.from_runs(S.heads, S.lens, S.ssa, S.esa)
    reads = S.sample_reads(300, 50, seed=3, sub_rate=0.1)
    seqs, off = ra.pack_reads(reads)
    lo, hi, k = rb.find_range_w_toehold(seqs, off)
    names = [f"r{i}".encode() for i in range(len(reads))]
    with pytest.raises(ra.RbgError):
        rb.align_text(lo, hi, k, names)
    # k = NULL: the report without -s (rb_align.cpp:120-122), no document list needed; empty ranges print count=0
    got = rb.align_text(lo, hi, None, names)
    assert got.decode() == "".join(f"{n.decode()} ({int(a)},{int(b)}), count={(int(b) - int(a) + 1) % 2**64}\n" for n, a, b in zip(names, lo, hi))
    unit = len(S.text) // 4
    starts = [0, unit, 2 * unit, 3 * unit]
    rb.set_docs([f"d{j}" for j in range(4)], starts)
    o.set_docs([f"d{j}" for j in range(4)], starts)
    for max_hits in (MAXU, 3, 1, 0):
        got = rb.align_text(lo, hi, k, names, max_hits)
        want = ""
        for n, a, b, kk in zip(names, lo, hi, k):
            a, b, kk = int(a), int(b), int(kk)
            want += f"{n.decode()} ({a},{b}), count={(b - a + 1) % 2**64}\n\tlocs: "
            if a <= b:
                for l in o.locs_at(a, b, kk, max_hits):
                    dn, offs = o.resolve_offset(l)
                    want += f"{l}/{dn}:{offs} "
            want += "\n"
        assert got.decode() == want
    rb.close()
    o.close()


# ---- the rb_markers-compatible CLI (reference src/rb_markers.cpp, default seeding mode) ----------
def _run_rb_markers(args):
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rowbowt_amd", "rb_markers")
    p = subprocess.run([exe] + args, capture_output=True, timeout=120)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


def test_cli_rb_markers_stdout(data_dir, tmp_path, small):
    import rb_markers_model as RM
    rb, o = small
    idx = os.path.join(data_dir, "small.fa")
    text = open(idx, "rb").read().split(b"\n", 1)[1].replace(b"\n", b"")
    rng = np.random.default_rng(77)
    recs = []
    for fn in ("simple_query.fq", "error_query.fq"):
        names, seqs = orc.read_fastx(os.path.join(data_dir, fn))
        recs += list(zip(names, seqs))
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    for i in range(300):   # 101 bp reads from either strand, some with errors, lower case and Ns
        p = int(rng.integers(0, len(text) - 101))
        q = bytearray(text[p:p + 101])
        if i % 2:
            q = bytearray(bytes(q).translate(comp)[::-1])
        for _ in range(int(rng.integers(0, 3))):
            q[int(rng.integers(0, 101))] = b"ACGTN"[int(rng.integers(0, 5))]
        if i % 7 == 0:
            q = bytearray(bytes(q).lower())
        recs.append((f"syn{i}".encode(), bytes(q)))
    recs.append((b"short", b"ACG"))
    recs.append((b"empty", b""))
    fq = tmp_path / "reads.fq"
    with open(fq, "wb") as f:
        for name, seq in recs:
            f.write(b"@" + name + b" x\n" + seq + b"\n+\n" + b"I" * len(seq) + b"\n")
    rc, out, err = _run_rb_markers([idx, str(fq)])
    assert rc == 0, err
    want = RM.expected_stdout(o, recs)
    assert out == want
    assert " + 0 20 0/289/0\n" in out and out.count("\n") > 2 * len(recs) - 10
    assert "loading rowbowt + markers took" in err and "counting markers took" in err
    for args, kw in ((["--wsize", "10", "--max-range", "3", "--min-range", "2"], dict(wsize=10, max_range=3, min_range=2)),
                     (["-w", "5", "--batch", "7", "--threads", "3"], dict(wsize=5)),
                     (["--heuristic"], dict(heuristic=True)),
                     (["--heuristic", "--best-strand-only", "--min-seed-length", "30", "--read-len", "101"],
                      dict(heuristic=True, best_strand=True, min_seed_len=30, read_len=101)),
                     (["--heuristic", "-y", "25", "--clear-conflicting", "--clear-identical", "-l", "50", "-w", "8"],
                      dict(heuristic=True, min_seed_len=25, clear_conflicting=True, clear_identical=True, read_len=50, wsize=8))):
        rc, out, err = _run_rb_markers(args + [idx, str(fq)])
        assert rc == 0, err
        assert out == RM.expected_stdout(o, recs, **kw), args
    # --ftab: the index prefix needs its .ftab (rb_build -f); seeds then go through search_ftab
    import shutil
    for suf in (".rbwt", ".mab"):
        shutil.copy(idx + suf, tmp_path / ("fx" + suf))
    rc, _, err = _run_rb_markers(["--ftab", str(tmp_path / "fx"), str(fq)])
    assert rc == 1 and "bad file" in err                      # no .ftab yet (rowbowt_io.hpp:166-169)
    rb.write_ftab(6, str(tmp_path / "fx.ftab"))
    long_recs = [r for r in recs if len(r[1]) >= 6]
    fq2 = tmp_path / "long.fq"
    with open(fq2, "wb") as f:
        for name, seq in long_recs:
            f.write(b"@" + name + b"\n" + seq + b"\n+\n" + b"I" * len(seq) + b"\n")
    rc, out, err = _run_rb_markers(["--ftab", "-w", "8", str(tmp_path / "fx"), str(fq2)])
    assert rc == 0, err
    assert out == RM.expected_stdout(o, long_recs, wsize=8, ftab_k=6)
    assert out != RM.expected_stdout(o, long_recs, wsize=8)
    rc, out, err = _run_rb_markers(["-f", "--heuristic", "--best-strand-only", "-y", "20", str(tmp_path / "fx"), str(fq2)])
    assert rc == 0 and out == RM.expected_stdout(o, long_recs, heuristic=True, best_strand=True, min_seed_len=20, ftab_k=6)
    rc, _, err = _run_rb_markers(["--ftab", "-w", "4", str(tmp_path / "fx"), str(fq2)])
    assert rc == 1 and "wsize cannot be greater" in err       # rowbowt.hpp:423-426 (k - 1 > wsize)
    rc, _, err = _run_rb_markers(["--ftab", str(tmp_path / "fx"), str(fq)])
    assert rc == 1 and "shorter than the ftab" in err         # the reference dies in substr (rowbowt.hpp:431)
    text_ftab = (tmp_path / "fx.ftab").read_text().splitlines()
    (tmp_path / "fx.ftab").write_text("\n".join(text_ftab[:-1] + [text_ftab[-1].rsplit(" ", 1)[0] + " 0"]) + "\n")
    rc, _, err = _run_rb_markers(["--ftab", str(tmp_path / "fx"), str(fq2)])
    assert rc == 1 and "ftab" in err                          # a table that is not this index's is refused
    # modes the reference itself refuses or that this engine does not build
    for flag in ("--overlap", "--lmem", "--fbb"):
        rc, _, err = _run_rb_markers([flag, idx, str(fq)])
        assert rc == 1 and err
    rc, _, err = _run_rb_markers([idx])
    assert rc == 1 and "no argument provided" in err
    rc, _, err = _run_rb_markers([str(tmp_path / "nope"), str(fq)])
    assert rc == 1 and "bad file" in err


# ---- next-row f1: rb_build outputs (native cache, the reference's text .ftab) ----------------------
def test_ftab_file_and_cache_only_prefix(data_dir, tmp_path, small, simple_reads):
    import itertools
    import subprocess
    rb, o = small
    # RowBowt::build_ftab(k) + FTab::serialize (rowbowt.hpp:726-744, ftab.hpp:29-34) against the oracle
    for k in (1, 3, 5):
        rb.write_ftab(k, str(tmp_path / f"k{k}.ftab"))
        want = ""
        for kmer in sorted("".join(t) for t in itertools.product("ACGT", repeat=k)):   # std::map order
            lo, hi = o.find_range(kmer.encode())
            if lo <= hi:
                want += f"{kmer} {lo} {hi}\n"
        assert (tmp_path / f"k{k}.ftab").read_text() == want
    # rb_build --from-index -f -k 10: cache + .ftab holding the FTab tests' answers (rb_tests.cpp:147-173)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "rowbowt_amd", "rb_build")
    out = tmp_path / "built" / "small"
    out.parent.mkdir()
    p = subprocess.run([exe, "--from-index", "-s", "-m", "-f", "-k", "10", "-o", str(out), os.path.join(data_dir, "small.fa")],
                       capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()
    lines = (tmp_path / "built" / "small.ftab").read_text().splitlines()
    assert lines == sorted(lines) and all(len(l.split()[0]) == 10 for l in lines)
    table = {l.split()[0]: (int(l.split()[1]), int(l.split()[2])) for l in lines}
    assert table["TTCGTCGTAA"] == G.KMER_RANGES[b"TTCGTCGTAA"] == (28942, 28944)
    n_kmers = 0
    for kmer, (lo, hi) in list(table.items())[::97]:
        assert o.find_range(kmer.encode()) == (lo, hi)
        n_kmers += 1
    assert n_kmers > 100 and len(table) <= 30031
    # -a / --ftab-only (rb_build.cpp:108-109): only the table, from the index already at the output prefix
    p = subprocess.run([exe, "-a", "-k", "3", "-o", str(out), os.path.join(data_dir, "small.fa")], capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()
    want3 = ""
    for kmer in sorted("".join(t) for t in itertools.product("ACGT", repeat=3)):
        lo, hi = o.find_range(kmer.encode())
        if lo <= hi:
            want3 += f"{kmer} {lo} {hi}\n"
    assert (tmp_path / "built" / "small.ftab").read_text() == want3
    # the CLIs run from a prefix that only has the cache (no .rbwt/.tsa/.mab): same bytes as from the reference's files
    (tmp_path / "built" / "small.docs").write_text("ref 0\nhap1 10010\nhap2 20020\n")
    import shutil
    for suf in (".rbwt", ".tsa", ".mab"):
        shutil.copy(os.path.join(data_dir, "small.fa" + suf), tmp_path / ("full" + suf))
    (tmp_path / "full.docs").write_text("ref 0\nhap1 10010\nhap2 20020\n")
    fq = os.path.join(data_dir, "simple_query.fq")
    a = _run_cli(["-s", "-m", str(out), fq])
    b = _run_cli(["-s", "-m", str(tmp_path / "full"), fq])
    assert a[0] == b[0] == 0 and a[1] == b[1] and "20306/hap2:286" in a[1]
    a = _run_rb_markers([str(out), fq])
    b = _run_rb_markers([os.path.join(data_dir, "small.fa"), fq])
    assert a[0] == b[0] == 0 and a[1] == b[1] and a[1]


# ---- mid-scale (n ~ 4e6) synthetic pangenome built with the bench's own generator on the GPU:
# count, toehold, locate, markers (BASELINE configs 2, 3, 5 in miniature) against the oracle -----
def test_midscale_pangenome_all_queries():
    import torch
    from rowbowt_amd.tools import synth_pangenome as sp
    dev = torch.device("cuda:0")
    text, info = sp.make_text(200_000, 20, 0.01, 77, dev)
    sa = sp.suffix_array(text)
    inp = sp.index_inputs(text, sa)
    n, unit, H, L = info["n"], info["unit"], info["H"], info["L"]
    rb = ra.RowBowt.from_runs(inp["heads"], inp["lens"], inp["ssa"], inp["esa"], device=0)
    o = orc.Oracle.from_runs(inp["heads"], inp["lens"], inp["ssa"], inp["esa"])
    assert rb.info().kmer_steps == 5 and rb.info().kmer_symbols == 4
    # marker array like small.fa.mab: rows whose suffix starts within w bases before a variant site
    w = 10
    tcpu = text.cpu().numpy()
    isa = np.empty(n, dtype=np.int64)
    isa[sa.cpu().numpy()] = np.arange(n)
    base = tcpu[:L]
    site_pos = np.flatnonzero((tcpu[: H * unit].reshape(H, unit)[:, :L] != base[None, :]).any(axis=0))
    tags = {}
    for h in range(H):
        hap = tcpu[h * unit:h * unit + L]
        for s in site_pos:
            allele = int(hap[s] != base[s])
            for d in range(w):
                p = s - d
                if p >= 0:
                    tags.setdefault(int(isa[h * unit + p]), set()).add(int(s) | (allele << 60))
    rows = sorted(tags)
    ms, me, mo, mv = [], [], [0], []
    for r in rows:
        vals = sorted(tags[r])
        if ms and me[-1] == r - 1 and mv[mo[-2]:mo[-1]] == vals:
            me[-1] = r
        else:
            ms.append(r); me.append(r); mv += vals; mo.append(len(mv))
    rb.set_markers(ms, me, mo, mv)
    o.set_markers(ms, me, mo, mv)
    N, m = 200_000, 100
    reads, _ = sp.sample_reads(text, info, N, m, seed=5, sub_rate=0.1)
    seqs = reads.cpu().numpy().reshape(-1)
    off = (np.arange(N + 1, dtype=np.uint64) * m)
    rb.counters_reset()
    lo, hi, k = rb.find_range_w_toehold(seqs, off)
    wlo, whi, wk = o.find_range_w_toehold_batch(seqs, off, nthreads=8)
    assert (lo == wlo).all() and (hi == whi).all() and (k == wk).all()
    clo, chi = rb.find_range(seqs, off)
    assert (clo == wlo).all() and (chi == whi).all()
    loc_off, locs = rb.locs_at(lo, hi, k)
    woff, wlocs = o.locs_at_batch(wlo, whi, wk, nthreads=8)
    assert (loc_off == woff).all() and (locs == wlocs).all()
    occ = np.where(whi >= wlo, whi - wlo + 1, 0)
    assert rb.counters().tolist() == [2 * N, 2 * int((whi >= wlo).sum()), 2 * int(occ.sum()), int(occ.sum())]
    mk_off, mk = rb.markers_at(lo, hi)
    got = split(mk_off[:2001], mk)
    hits = 0
    for i in range(2000):
        want = o.markers_at(int(lo[i]), int(hi[i]))
        assert got[i] == want
        hits += bool(want)
    assert hits > 80  # ~ w * site_rate of the matched reads start within a marker window
    sub = slice(0, 3000 * m)
    lo2, hi2, mk_off2, mk2 = rb.find_range_w_markers(seqs[sub], off[:3001], 19, 1000)  # rb_markers defaults (rb_markers.cpp:29-30)
    got2 = split(mk_off2, mk2)
    for i in range(3000):
        (wl, wh), wm = o.find_range_w_markers(reads[i].cpu().numpy().tobytes(), 19, 1000)
        assert (int(lo2[i]), int(hi2[i])) == (wl, wh) and got2[i] == wm
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
        ra.set_default_option(capi.OPT_KMER_STEPS, 5)
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


def test_cpp_shim_reference_goldens(tmp_path, data_dir):
    """The reference's own test assertions, issued through the C++ shim with the reference's signatures."""
    import shutil
    import subprocess
    from test_capi_host import _compile_shim_test
    exe = _compile_shim_test(tmp_path)
    for suf in (".rbwt", ".tsa"):
        shutil.copy(os.path.join(data_dir, "small.fa" + suf), tmp_path / ("idx" + suf))
    (tmp_path / "idx.docs").write_text("ref 0\nhap1 10010\nhap2 20020\n")
    p = subprocess.run([str(exe), data_dir, str(tmp_path / "idx")], capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert b"shim goldens ok" in p.stdout


def test_cpp_threaded_one_read_caller(tmp_path, data_dir):
    """tests/cpp/shim_threads.cpp: a thread pool calling the reference's one-query methods through the shim, unmodified
    (rb_markers.cpp:318-535's shape); answers equal the batch forms, and the library's micro-batching queue serves the
    calls with fewer launches than calls (printed: rate with and without it)"""
    import subprocess
    from test_capi_host import ROOT
    exe = tmp_path / "shim_threads"
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "rowbowt_amd", "include"),
                           os.path.join(ROOT, "tests", "cpp", "shim_threads.cpp"), "-o", str(exe),
                           "-L", os.path.join(ROOT, "rowbowt_amd"), "-lrbg", "-Wl,-rpath," + os.path.join(ROOT, "rowbowt_amd")])
    prefix, fasta = os.path.join(data_dir, "small.fa"), os.path.join(data_dir, "small.fa")
    for combine, threads in (("1", 16), ("0", 16), ("1", 1)):
        env = dict(os.environ, RBG_HOST_COMBINE=combine)
        p = subprocess.run([str(exe), prefix, fasta, str(threads), "4000"], capture_output=True, timeout=600, env=env)
        assert p.returncode == 0 and b"shim threads ok" in p.stdout, p.stdout.decode()[-500:] + p.stderr.decode()[-1500:]
        print(f"RBG_HOST_COMBINE={combine}:", p.stdout.decode().splitlines()[0])
        if combine == "1" and threads == 16:
            m = re.search(r"(\d+) one-read calls in (\d+) launches", p.stdout.decode())
            assert m and int(m.group(1)) == 9000 and int(m.group(2)) < int(m.group(1))


def test_c_abi_example_program(tmp_path, data_dir):
    """tests/c/abi_usage.c (plain C11 over include/rbg.h): the reference's golden values through the C-ABI"""
    import subprocess
    from test_capi_host import _compile_c_example
    exe = _compile_c_example(tmp_path)
    p = subprocess.run([str(exe), os.path.join(data_dir, "small.fa")], capture_output=True, timeout=120)
    assert p.returncode == 0 and b"abi_usage ok" in p.stdout, p.stderr.decode()


@pytest.mark.parametrize("packed", [0, 2])
def test_long_and_ragged_reads(synth, packed, request):
    """Reads far longer than the 100 bp of the bench (whole haplotypes, the whole text, longer than the
    text), mixed with tiny ones in one batch (with packed reads: groups that do not fit the pack
    kernel's LDS staging take its direct path)."""
    S = synth
    ra.set_default_option(capi.OPT_PACKED_READS, packed)
    request.addfinalizer(lambda: ra.set_default_option(capi.OPT_PACKED_READS, 1))
    rb = ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0)
    o = orc.Oracle.from_runs(S.heads, S.lens, S.ssa, S.esa)
    unit = S.L + S.pad
    t = S.text.tobytes()
    reads = [t[:S.L], t[unit:unit + S.L], t[3 * unit + 17:4 * unit - 33], t[:-1], t, t + b"A", t[5:3000] * 3, b"T", b"",
             t[unit - 40:unit + 40], t[-200:-1], t[1:2 * unit]]
    seqs, off = ra.pack_reads(reads)
    lo, hi, k = rb.find_range_w_toehold(seqs, off)
    wlo, whi, wk = o.find_range_w_toehold_batch(seqs, off)
    assert (lo == wlo).all() and (hi == whi).all() and (k == wk).all()
    assert int(hi[0]) >= int(lo[0]) and (int(lo[4]), int(hi[4])) != (1, 0)  # (LF is cyclic: text+x may still match)
    loc_off, locs = rb.locs_at(lo, hi, k)
    woff, wlocs = o.locs_at_batch(wlo, whi, wk)
    assert (loc_off == woff).all() and (locs == wlocs).all()
    rb.close()
    o.close()


@pytest.mark.parametrize("phi_shift", [3, 5, 6, 7])
def test_packed_phi_slots_at_8_byte_positions(synth, phi_shift):
    """8-byte positions, n < 2^38, buckets of at most 64 positions: the phi slots are the 16-byte packed form
    (rbg_dev.h PhiSlotPacked; shift 7 keeps the 32-byte form) -- same locations as the oracle and as RBG_PHI_PACKED=0,
    buckets with 0, 1, 2 and more than 2 sampled positions, toeholds that wrapped below zero included"""
    S = synth
    o = orc.Oracle.from_runs(S.heads, S.lens, S.ssa, S.esa)
    reads = S.sample_reads(3000, 60, seed=23, sub_rate=0.1, ragged=True) + [S.text[:300].tobytes(), S.text[-40:-1].tobytes(), b"A", b""]
    seqs, off = ra.pack_reads(reads)
    wlo, whi, wk = o.find_range_w_toehold_batch(seqs, off)
    woff, wlocs = o.locs_at_batch(wlo, whi, wk)
    sizes = {}
    for packed in ("1", "0"):
        os.environ["RBG_PHI_PACKED"] = packed
        ra.set_default_option(capi.OPT_POS_BYTES, 8)
        ra.set_default_option(capi.OPT_PHI_BUCKET_SHIFT, phi_shift)
        try:
            rb = ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0)
        finally:
            ra.set_default_option(capi.OPT_POS_BYTES, 0)
            ra.set_default_option(capi.OPT_PHI_BUCKET_SHIFT, -1)
            del os.environ["RBG_PHI_PACKED"]
        assert rb.info().pos_bytes == 8 and rb.info().phi_bucket_shift == phi_shift
        sizes[packed] = int(rb.info().hbm_bytes)
        lo, hi, k = rb.find_range_w_toehold(seqs, off)
        assert (lo == wlo).all() and (hi == whi).all() and (k == wk).all()
        for mh in (MAXU, 3):
            loc_off, locs = rb.locs_at(lo, hi, k, max_hits=mh)
            w2off, w2locs = (woff, wlocs) if mh == MAXU else o.locs_at_batch(wlo, whi, wk, max_hits=mh)
            assert (loc_off == w2off).all() and (locs == w2locs).all()
        rb.close()
    # (the arena rounds every array to 64 KB: on this small index the halved slots show from 8-position buckets down)
    assert sizes["1"] <= sizes["0"] and (phi_shift != 3 or sizes["1"] < sizes["0"]) and (phi_shift <= 6 or sizes["1"] == sizes["0"])
    o.close()


@pytest.mark.parametrize("layout", [capi.LAYOUT_PREFER_SLOTS, capi.LAYOUT_AUTO])
def test_hbm_budget_drops_kmer_levels(synth, layout):
    """A tight memory budget keeps fewer k-mer levels (RBG_LAYOUT_PREFER_SLOTS) -- or, under RBG_LAYOUT_AUTO, switches to the
    run-indexed layout with all five symbols per step while that fits (about 110 bytes per run); answers do not change."""
    S = synth
    ra.set_default_option(capi.OPT_RANK_LAYOUT, layout)
    try:
        _hbm_budget_drops_kmer_levels(S, layout)
    finally:
        ra.set_default_option(capi.OPT_RANK_LAYOUT, capi.LAYOUT_AUTO)


def _hbm_budget_drops_kmer_levels(S, layout):
    reads = S.sample_reads(500, 70, seed=2, sub_rate=0.1)
    seqs, off = ra.pack_reads(reads)
    o = orc.Oracle.from_runs(S.heads, S.lens, S.ssa, S.esa)
    want = o.find_range_w_toehold_batch(seqs, off)
    rb = ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0)
    full = rb.info().hbm_bytes
    assert rb.info().kmer_steps == 5
    rb.close()
    seen = set()
    for frac in (4.0, 0.7, 0.3, 0.08, 0.02):   # (the budget rule prices the replica with the composition's own lists: about 2.7 x what stays)
        ra.set_default_option(capi.OPT_HBM_BUDGET_MB, max(1, int(full * frac) >> 20))
        try:
            try:
                rb = ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0)
            except ra.RbgError as e:
                assert e.code == -5  # RBG_ENOMEM: not even the single-symbol tables fit
                continue
        finally:
            ra.set_default_option(capi.OPT_HBM_BUDGET_MB, 0)
        runs = int(rb.info().rank_layout) == capi.LAYOUT_RUNS
        seen.add("runs" if runs and layout == capi.LAYOUT_AUTO else int(rb.info().kmer_steps))
        # levels that cannot fit are not even composed (options_for in rbg_capi.hip); what was asked for is still reported
        assert int(rb.info().kmer_steps_requested) == 5 and int(rb.info().hbm_budget) == max(1, int(full * frac) >> 20) << 20
        if runs and layout == capi.LAYOUT_AUTO and 110 * len(S.heads) <= int(rb.info().hbm_budget):
            assert int(rb.info().kmer_steps) == 5        # the switch was made to keep the symbols per step
        got = rb.find_range_w_toehold(seqs, off)
        assert all((g == w).all() for g, w in zip(got, want))
        rb.close()
    if layout == capi.LAYOUT_AUTO:
        assert "runs" in seen and 5 in seen, seen        # five symbols from slot tables while they fit, then the run-indexed layout
    else:
        assert len(seen) >= 2 and min(seen) < 5
    o.close()


@pytest.mark.parametrize("deep", [9, 10, 12])
def test_wide_buckets_on_the_deep_levels(synth, deep):
    """The 4-mer and deeper tables in the wide-bucket encoding (rbg_dev.h): same answers, smaller replica"""
    S = synth
    o = orc.Oracle.from_runs(S.heads, S.lens, S.ssa, S.esa)
    base = ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0)
    ra.set_default_option(capi.OPT_DEEP_BUCKET_SHIFT, deep)
    try:
        rb = ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0)
    finally:
        ra.set_default_option(capi.OPT_DEEP_BUCKET_SHIFT, -1)
    assert rb.info().kmer_steps == 5 and rb.info().hbm_bytes <= base.info().hbm_bytes
    reads = S.sample_reads(3000, 80, seed=17, sub_rate=0.2, ragged=True) + [b"", b"ACGTN", S.text[:400].tobytes()]
    seqs, off = ra.pack_reads(reads)
    lo, hi, k = rb.find_range_w_toehold(seqs, off)
    wlo, whi, wk = o.find_range_w_toehold_batch(seqs, off)
    assert (lo == wlo).all() and (hi == whi).all() and (k == wk).all()
    loc_off, locs = rb.locs_at(lo, hi, k)
    woff, wlocs = o.locs_at_batch(wlo, whi, wk)
    assert (loc_off == woff).all() and (locs == wlocs).all()
    _check_marker_seeds(rb, o, reads[::7], 10, 1000)
    goff, glocs = rb.find_locs_greedy_seeding(seqs, off, 10)
    for i in range(0, len(reads), 11):
        assert glocs[int(goff[i]):int(goff[i + 1])].tolist() == o.greedy_locate(reads[i], 10)[0]
    rb.close()
    base.close()
    o.close()


@pytest.mark.parametrize("rshift", [8, 5, 1])
@pytest.mark.parametrize("pos_bytes", [4, 8])
def test_dense_overflow_buckets(rshift, pos_bytes):
    """Buckets with more run starts than a slot holds (rbg_dev.h): with their dense tables (default) and with
    the run-list search (RBG_OPT_DENSE_OVERFLOW = 0) the answers are the oracle's.  A near-random text over
    five symbols puts almost every 256-row bucket of every table in that state."""
    import naive
    rng = np.random.default_rng(77)
    body = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=6000, p=[0.3, 0.25, 0.25, 0.19, 0.01])
    body[2000:2600] = body[100:700]          # some repetition, so that reads match more than once
    body[4000:4300] = ord("A")               # and one long run next to the busy rows
    text = np.concatenate([body, np.array([1], np.uint8)])
    sa = naive.suffix_array(text)
    heads, lens, brk = naive.rle(naive.bwt_from_sa(text, sa))
    ssa, esa = naive.run_samples(sa, brk, len(text))
    o = orc.Oracle.from_runs(heads, lens, ssa, esa)
    tb = text.tobytes()
    reads = [tb[a:a + int(rng.integers(1, 50))] for a in rng.integers(0, len(tb) - 1, size=3000)]
    reads += [b"", tb[:300], tb[-40:], b"A" * 200, b"AC" * 30]
    seqs, off = ra.pack_reads(reads)
    wlo, whi, wk = o.find_range_w_toehold_batch(seqs, off)
    woff, wlocs = o.locs_at_batch(wlo, whi, wk)
    sizes = {}
    for dense in (1, 0):
        ra.set_default_option(capi.OPT_DENSE_OVERFLOW, dense)
        ra.set_default_option(capi.OPT_RANK_BUCKET_SHIFT, rshift)
        ra.set_default_option(capi.OPT_POS_BYTES, pos_bytes)
        try:
            rb = ra.RowBowt.from_runs(heads, lens, ssa, esa, device=0)
        finally:
            ra.set_default_option(capi.OPT_DENSE_OVERFLOW, 1)
            ra.set_default_option(capi.OPT_RANK_BUCKET_SHIFT, -1)
            ra.set_default_option(capi.OPT_POS_BYTES, 0)
        i = rb.info()
        sizes[dense] = i.hbm_bytes
        if rshift == 8:   # (rank_slots counts the 1365 tables of all five k-mer levels, most of them nearly empty here)
            assert i.rank_slots_overflow >= 60
        lo, hi, k = rb.find_range_w_toehold(seqs, off)
        lo2, hi2 = rb.find_range(seqs, off)
        assert (lo == wlo).all() and (hi == whi).all() and (k == wk).all() and (lo2 == wlo).all() and (hi2 == whi).all()
        loc_off, locs = rb.locs_at(lo, hi, k)
        assert (loc_off == woff).all() and (locs == wlocs).all()
        goff, glocs = rb.find_locs_greedy_seeding(seqs, off, 6)
        for j in range(0, len(reads), 13):
            assert glocs[int(goff[j]):int(goff[j + 1])].tolist() == o.greedy_locate(reads[j], 6)[0]
        _check_marker_seeds(rb, o, reads[::11], 5, 1000)
        rb.close()
    assert sizes[1] >= sizes[0]
    o.close()


def test_positions_beyond_32_bits():
    """n > 2^32: 8-byte positions chosen automatically, rank values above 2^32 in the 16-byte slots, 64-bit phi
    slots, and the HBM-budget rule at work by itself (five levels in 256-row buckets would need 540 GB).
    No text of that size is needed: rank, LF, the toehold bookkeeping and phi are arithmetic on the run-length
    BWT and its run-boundary samples, so a synthetic run list (random heads and lengths, distinct random
    samples) defines them completely -- for the oracle and for the device alike.  Reads that match are read off
    LF walks: c0 = bwt[i0], i1 = LF(i0), c1 = bwt[i1], ... is matched by the pattern c_k ... c1 c0."""
    rng = np.random.default_rng(4242)
    r = 20_000_000
    sym = np.frombuffer(b"ACGT", dtype=np.uint8)
    step = rng.integers(1, 4, size=r, dtype=np.int64)
    step[0] = 0
    heads = sym[np.cumsum(step) % 4]                       # neighbouring runs differ
    lens = rng.integers(1, 500, size=r, dtype=np.int64).astype(np.uint64)
    heads[r // 3], lens[r // 3] = 1, 1                     # one terminator, as every BWT of a text has (the k-mer levels ask for it)
    n = int(lens.sum())
    assert n > (1 << 32) + (1 << 29)
    stride = n // (2 * r)
    vals = (np.arange(2 * r, dtype=np.uint64) * np.uint64(stride) + rng.integers(0, stride, size=2 * r).astype(np.uint64))
    rng.shuffle(vals)                                      # distinct sample values below n
    ssa, esa = vals[:r].copy(), vals[r:].copy()
    o = orc.Oracle.from_runs(heads, lens, ssa, esa)
    starts = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
    reads = []
    for row in rng.integers(0, n, size=1500):
        row, m, q = int(row), int(rng.integers(1, 120)), bytearray()
        for _ in range(m):
            c = int(heads[np.searchsorted(starts, row, side="right") - 1])
            q.append(c)
            row = o.LF(row, row, c)[0]
        reads.append(bytes(q[::-1]))
    reads += [bytes(rng.choice(sym, size=int(rng.integers(1, 40)))) for _ in range(500)] + [b"", b"ACGTN"]
    seqs, off = ra.pack_reads(reads)
    wlo, whi, wk = o.find_range_w_toehold_batch(seqs, off)
    woff, wlocs = o.locs_at_batch(wlo, whi, wk, max_hits=64)
    assert int((whi >= wlo).sum()) >= 1500 and int(wlo.max()) > (1 << 32) and int(wlocs.max()) > (1 << 32)
    # single-symbol steps: ranges, toeholds and locations
    ra.set_default_option(capi.OPT_KMER_STEPS, 1)
    try:
        rb1 = ra.RowBowt.from_runs(heads, lens, ssa, esa, device=0)
    finally:
        ra.set_default_option(capi.OPT_KMER_STEPS, 5)
    assert rb1.info().n == n and rb1.info().pos_bytes == 8 and rb1.info().kmer_steps == 1
    lo, hi, k = rb1.find_range_w_toehold(seqs, off)
    assert (lo == wlo).all() and (hi == whi).all() and (k == wk).all()
    loc_off, locs = rb1.locs_at(lo, hi, k, max_hits=64)
    assert (loc_off == woff).all() and (locs == wlocs).all()
    rb1.close()
    # k-mer levels (the deepest one dropped by the budget rule): ranges, and the phi walks from the oracle's
    # toeholds.  The toeholds of k-mer steps are not compared here: composing the run-end samples of a k-mer
    # table presumes samples that are the suffix array's, which random ones are not.
    rb = ra.RowBowt.from_runs(heads, lens, ssa, esa, device=0)
    i = rb.info()
    assert i.n == n and i.pos_bytes == 8 and 2 <= i.kmer_steps <= 5 and i.hbm_bytes < 235e9   # (the budget rule widens the deep levels' buckets, then drops levels)
    lo, hi, _ = rb.find_range_w_toehold(seqs, off)
    lo2, hi2 = rb.find_range(seqs, off)
    assert (lo == wlo).all() and (hi == whi).all() and (lo2 == wlo).all() and (hi2 == whi).all()
    loc_off, locs = rb.locs_at(wlo, whi, wk, max_hits=64)
    assert (loc_off == woff).all() and (locs == wlocs).all()
    # the run-indexed layout at this size: 3 GB instead of 140, sampled index three levels deep with a 16-key top
    # (single steps: toeholds compared; k-mer depths: ranges and the walks from the oracle's toeholds, as above)
    for top_kb, ks in ((48, 1), (0, 1), (48, 5), (0, 3)):
        ra.set_default_option(capi.OPT_KMER_STEPS, ks)
        ra.set_default_option(capi.OPT_RUN_PHI, 1 if top_kb else 2)   # phi over the list of sampled positions / through phi slots of about n / r rows
        try:
            rbr = _with_layout(capi.LAYOUT_RUNS, top_kb, lambda: ra.RowBowt.from_runs(heads, lens, ssa, esa, device=0))
        finally:
            ra.set_default_option(capi.OPT_KMER_STEPS, 5)
            ra.set_default_option(capi.OPT_RUN_PHI, 0)
        ir = rbr.info()
        assert ir.rank_layout == capi.LAYOUT_RUNS and ir.pos_bytes == 8 and ir.kmer_steps == ks and ir.hbm_bytes < (4e9 if ks == 1 else 12e9) + (0 if top_kb else 3e9)
        lo, hi, k = rbr.find_range_w_toehold(seqs, off)
        lo2, hi2 = rbr.find_range(seqs, off)
        assert (lo == wlo).all() and (hi == whi).all() and (lo2 == wlo).all() and (hi2 == whi).all() and (ks > 1 or (k == wk).all())
        loc_off, locs = rbr.locs_at(wlo, whi, wk, max_hits=64)
        assert (loc_off == woff).all() and (locs == wlocs).all()
        rbr.close()
    rows = rng.integers(0, n, size=4000).astype(np.uint64)
    width = rng.integers(0, 3000, size=4000).astype(np.uint64)
    his = np.minimum(rows + width, np.uint64(n - 1))
    cs = rng.choice(sym, size=4000)
    nlo, nhi = rb.LF(rows, his, cs)
    for j in range(0, 4000, 7):
        assert (int(nlo[j]), int(nhi[j])) == o.LF(int(rows[j]), int(his[j]), int(cs[j]))
    rb.close()
    o.close()


def _random_run_index(rng, r, max_len, term_at=None):
    """A synthetic run list (random heads over ACGT with neighbouring runs different, random lengths, distinct random
    samples below n) with one terminator run: rank, LF, the toehold bookkeeping and phi are arithmetic on these arrays
    alone, so they define the answers completely -- for the oracle and for the device alike (no text needed)."""
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


def test_width_limits_2_38_and_2_40():
    """The position widths the layouts are built around, each crossed by a test (the reference computes in plain
    uint64_t: toehold_sa.hpp:56-72, rowbowt.hpp:555-573, rle_string.hpp:131-161):
      * n >= 2^38: phi slots can no longer be packed into 16 bytes (PhiSlotPacked holds 38-bit values) -- the slot
        layout must fall back to the 32-byte PhiSlot<uint64_t>; ranks above 2^38 in the 48-bit RankSlot;
      * n just below 2^40: the wide-bucket slot encoding (40-bit ranks, rbg_dev.h) at its largest values;
      * n >= 2^40: wide buckets are refused when forced (RBG_EARG) and never chosen by the budget rule; the
        run-indexed layout (8-byte positions throughout) is what serves such an index;
      * n >= 2^48: refused at flatten (RankSlot carries 48-bit ranks).
    Run lists are synthetic (r = 2*10^7, mean run 1.4*10^4 .. 5.5*10^4): see _random_run_index."""
    sym = np.frombuffer(b"ACGT", dtype=np.uint8)
    rng = np.random.default_rng(3838)
    r = 20_000_000

    def case(max_len):
        heads, lens, ssa, esa, n = _random_run_index(rng, r, max_len)
        o = orc.Oracle.from_runs(heads, lens, ssa, esa)
        reads = _lf_walk_reads(o, heads, lens, n, rng, 1200, 100)
        reads += [bytes(rng.choice(sym, size=int(rng.integers(1, 30)))) for _ in range(400)] + [b"", b"ACGTN"]
        seqs, off = ra.pack_reads(reads)
        wlo, whi, wk = o.find_range_w_toehold_batch(seqs, off)
        woff, wlocs = o.locs_at_batch(wlo, whi, wk, max_hits=48)
        return heads, lens, ssa, esa, n, o, seqs, off, wlo, whi, wk, woff, wlocs

    def check(rb, c, toeholds=True):
        heads, lens, ssa, esa, n, o, seqs, off, wlo, whi, wk, woff, wlocs = c
        lo, hi, k = rb.find_range_w_toehold(seqs, off)
        lo2, hi2 = rb.find_range(seqs, off)
        assert (lo == wlo).all() and (hi == whi).all() and (lo2 == wlo).all() and (hi2 == whi).all()
        assert not toeholds or (k == wk).all()
        loc_off, locs = rb.locs_at(wlo, whi, wk, max_hits=48)
        assert (loc_off == woff).all() and (locs == wlocs).all()

    # ---- n in (2^38, 2^40): 2^38 = 2.75e11 ----------------------------------------------------------------------------
    c = case(30_000)
    n = c[4]
    assert (1 << 38) < n < (1 << 40) and int(c[8].max()) > (1 << 38) and int(c[12].max()) > (1 << 38)
    # run-indexed: single steps (toeholds compared) and k-mer depths (ranges + walks from the oracle's toeholds: the
    # k-mer tables' run-end samples presume a suffix array's samples)
    for ks in (1, 5):
        with capi.default_option(capi.OPT_KMER_STEPS, ks):
            rbr = _with_layout(capi.LAYOUT_RUNS, 48, lambda: ra.RowBowt.from_runs(*c[:4], device=0))
        ir = rbr.info()
        assert ir.rank_layout == capi.LAYOUT_RUNS and ir.pos_bytes == 8 and ir.n == n and ir.hbm_bytes < 16e9
        check(rbr, c, toeholds=ks == 1)
        rbr.close()
    # slot tables, single-symbol level, 256-row rank buckets and 256-position phi buckets: n/256 x (5 x 20 + 36) bytes
    with capi.default_option(capi.OPT_KMER_STEPS, 1), capi.default_option(capi.OPT_PHI_BUCKET_SHIFT, 8):
        rbs = _with_layout(capi.LAYOUT_SLOTS, 48, lambda: ra.RowBowt.from_runs(*c[:4], device=0))
    i = rbs.info()
    assert i.rank_layout == capi.LAYOUT_SLOTS and i.pos_bytes == 8 and i.phi_bucket_shift == 8 and i.kmer_steps == 1
    # 32-byte phi slots: (n >> 8) x 32 bytes alone exceed what packed 16-byte slots would take for the whole table
    assert i.hbm_bytes > (n >> 8) * (5 * 20 + 36)
    check(rbs, c)
    rows = rng.integers(0, n, size=3000).astype(np.uint64)
    his = np.minimum(rows + rng.integers(0, 100_000, size=3000).astype(np.uint64), np.uint64(n - 1))
    cs = rng.choice(sym, size=3000)
    nlo, nhi = rbs.LF(rows, his, cs)
    for j in range(0, 3000, 5):
        assert (int(nlo[j]), int(nhi[j])) == c[5].LF(int(rows[j]), int(his[j]), int(cs[j]))
    rbs.close()
    c[5].close()
    del c

    # ---- n just below 2^40: the wide-bucket encoding (4096-row buckets, 40-bit ranks) at the top of its range --------
    c = case(109_000)
    n = c[4]
    assert (1 << 40) - (1 << 36) < n < (1 << 40), n
    with capi.default_option(capi.OPT_KMER_STEPS, 1), capi.default_option(capi.OPT_RANK_BUCKET_SHIFT, 12), \
            capi.default_option(capi.OPT_PHI_BUCKET_SHIFT, 8), capi.default_option(capi.OPT_HBM_BUDGET_MB, 240_000):
        rbw = _with_layout(capi.LAYOUT_SLOTS, 48, lambda: ra.RowBowt.from_runs(*c[:4], device=0))
    i = rbw.info()
    assert i.rank_layout == capi.LAYOUT_SLOTS and i.rank_bucket_shift == 12 and i.pos_bytes == 8
    assert int(c[8].max()) > (1 << 39)                     # ranks in the top half of the 40-bit range
    check(rbw, c)
    rbw.close()
    c[5].close()
    del c

    # ---- n just above 2^40: wide buckets refused, run-indexed layout serves ---------------------------------------------
    c = case(112_000)
    n = c[4]
    assert (1 << 40) < n < (1 << 40) + (1 << 37), n
    with capi.default_option(capi.OPT_KMER_STEPS, 1), capi.default_option(capi.OPT_RANK_BUCKET_SHIFT, 12), \
            capi.default_option(capi.OPT_PHI_BUCKET_SHIFT, 8):
        with pytest.raises(ra.RbgError) as ei:
            _with_layout(capi.LAYOUT_SLOTS, 48, lambda: ra.RowBowt.from_runs(*c[:4], device=0))
        assert ei.value.code == -4                          # RBG_EARG: 40-bit ranks cannot hold this index
    rba = ra.RowBowt.from_runs(*c[:4], device=0)            # AUTO: the single-symbol slot level (n/256 x 100 B = 430 GB) does not fit
    ia = rba.info()
    assert ia.rank_layout == capi.LAYOUT_RUNS and ia.pos_bytes == 8 and ia.hbm_bytes < 16e9
    check(rba, c, toeholds=False)                           # (k-mer depths: see above)
    rba.close()
    with capi.default_option(capi.OPT_KMER_STEPS, 1):
        rb1 = ra.RowBowt.from_runs(*c[:4], device=0)
    assert rb1.info().rank_layout == capi.LAYOUT_RUNS and int(c[12].max()) > (1 << 40)
    check(rb1, c)
    rb1.close()
    c[5].close()
    del c

    # ---- n >= 2^48: refused (48-bit ranks in RankSlot; the run-indexed tables share flatten()) ---------------------------
    heads, lens, ssa, esa, n = _random_run_index(rng, 2_000_000, 300_000_000)
    assert n > (1 << 48)
    for layout in (capi.LAYOUT_SLOTS, capi.LAYOUT_RUNS):
        with pytest.raises(ra.RbgError) as ei:
            _with_layout(layout, 48, lambda: ra.RowBowt.from_runs(heads, lens, ssa, esa, device=0))
        assert ei.value.code == -4


def _with_layout(layout, top_kb, build):
    ra.set_default_option(capi.OPT_RANK_LAYOUT, layout)
    ra.set_default_option(capi.OPT_TREE_TOP_KB, top_kb)
    try:
        return build()
    finally:
        ra.set_default_option(capi.OPT_RANK_LAYOUT, capi.LAYOUT_AUTO)
        ra.set_default_option(capi.OPT_TREE_TOP_KB, 48)


@pytest.mark.parametrize("pos_bytes,top_kb,fk,ks,rec,fmt", [(0, 48, -1, 5, None, 2), (8, 48, -1, 5, None, 2), (0, 48, -1, 5, None, 1), (8, 48, -1, 5, None, 1),
                                                            (0, 0, -1, 5, "200", 1), (8, 0, 3, 4, None, 1),
                                                            (0, 1, 0, 3, None, 1), (0, 48, -1, 1, "1", 1), (8, 1, -1, 2, "40", 1), (0, 0, 0, 1, "8", 1),
                                                            (8, 48, 0, 5, "3", 1), (0, 48, 0, 5, "8", 1),
                                                            (8, 48, 0, 1, None, 2), (0, 48, 3, 3, None, 2), (8, 48, 3, 2, None, 2), (0, 48, 0, 4, None, 2),
                                                            (8, 48, -1, 4, None, 2), (0, 48, -1, 5, None, 22), (8, 48, -1, 5, None, 22), (8, 48, 0, 3, None, 22),
                                                            (0, 48, -1, 5, None, 23), (8, 48, -1, 5, None, 23), (8, 48, 0, 4, None, 23), (0, 48, 3, 1, None, 23),
                                                            (0, 48, -1, 5, None, 24), (8, 48, -1, 5, None, 24), (8, 48, 3, 2, None, 24)])
def test_run_indexed_layout(synth, pos_bytes, top_kb, fk, ks, rec, fmt):
    """RBG_LAYOUT_RUNS (k_runs.hip): space proportional to r, rank and phi as wave-cooperative predecessor searches
    over the run lists (rle_string.hpp:131-161, toehold_sa.hpp:56-72), k-mer steps through one clamped search per
    depth (ks = symbols per step) -- same answers as the slot tables, i.e. as the oracle, on every read shape of
    test_synth_all_paths; top_kb = 0 forces the deepest sampled index.  rec: RBG_RANK_REC, the average number of runs
    per bucket record (None: no records, directories and run lists only; a large value: most buckets overflow their
    record and go through the run list; a small one: narrow buckets, clipped predecessors everywhere)."""
    S = synth
    # fmt 22 = format 2 with phi SLOTS (RBG_OPT_RUN_PHI = 2: the slot layout's direct-addressed phi records at buckets of about n / r rows);
    # plain 2 pins phi to the list of sampled positions and its directory
    # 23 / 24 = the same two with BUCKET RECORDS instead of the rank directories (RBG_OPT_RUN_REC = 2: one aligned 64-byte record per bucket)
    phi_slots, recs = fmt in (22, 24), fmt in (23, 24)
    fmt = 2 if fmt in (22, 23, 24) else fmt
    ra.set_default_option(capi.OPT_RUN_PHI, 2 if phi_slots else 1)
    ra.set_default_option(capi.OPT_RUN_REC, 2 if recs else 1)
    ra.set_default_option(capi.OPT_POS_BYTES, pos_bytes)
    ra.set_default_option(capi.OPT_FTAB_K, fk)
    ra.set_default_option(capi.OPT_KMER_STEPS, ks)
    ra.set_default_option(capi.OPT_RUN_FMT, fmt)   # 2: one lane per query (rbg_runs2_device.hpp); 1: quads of lanes / the descent / bucket records
    if fmt == 1 and ks in (2, 4):
        os.environ["RBG_PHI_DIR"] = "0"   # phi by the descent through the sampled levels only (no directory)
    if fmt == 1 and ks in (3, 4):
        os.environ["RBG_RANK_DIR"] = "0"  # ranks likewise
    if rec is not None:
        os.environ["RBG_RANK_REC"] = rec
    # (compact records hold eleven entries: at the default 2.5 per bucket none of this index's buckets overflows; the 8-byte variants
    #  take buckets of about nine entries so that overflowing records -- pivots, then the run list -- are met here too)
    rec_per = "9" if recs and pos_bytes == 8 else None
    if rec_per:
        os.environ["RBG_RUN_REC_PER"] = rec_per
    if rec is not None or ks == 4:   # run lists at every depth (otherwise the default: every other depth from the deepest down)
        ra.set_default_option(capi.OPT_RUN_DEPTHS, (1 << ks) - 1)
    try:
        rb = _with_layout(capi.LAYOUT_RUNS, top_kb, lambda: ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0))
    finally:
        ra.set_default_option(capi.OPT_RUN_FMT, 2)
        ra.set_default_option(capi.OPT_RUN_PHI, 0)
        ra.set_default_option(capi.OPT_RUN_REC, 0)
        ra.set_default_option(capi.OPT_RUN_DEPTHS, 0)
        ra.set_default_option(capi.OPT_POS_BYTES, 0)
        ra.set_default_option(capi.OPT_FTAB_K, -1)
        ra.set_default_option(capi.OPT_KMER_STEPS, 5)
        os.environ.pop("RBG_PHI_DIR", None)
        os.environ.pop("RBG_RANK_DIR", None)
        os.environ.pop("RBG_RANK_REC", None)
        os.environ.pop("RBG_RUN_REC_PER", None)
    info = rb.info()
    assert info.rank_layout == capi.LAYOUT_RUNS and info.kmer_steps == ks and info.pos_bytes == (pos_bytes or 4)
    assert info.rank_slots == 0 and (info.phi_slots == 0) != phi_slots
    lists = [d for d, x in ((2, info.pair_runs), (3, info.triple_runs), (4, info.quad_runs), (5, info.quint_runs)) if x]
    assert lists == ([d for d in range(2, ks + 1)] if rec is not None or ks == 4 else [d for d in range(2, ks + 1) if (ks - d) % 2 == 0])
    li = rb.layout_info()
    assert li.run_fmt == fmt and li.depths_dropped_budget == 0 and li.depths_dropped_limit == 0 and li.phi_directory_dropped == 0
    assert [d + 1 for d in range(5) if li.depth_mask_kept >> d & 1] == [1] + [d for d in lists]
    if fmt == 2:
        assert li.rank_directories == (0 if recs else 1) and li.phi_entries == len(S.heads) and sum(li.fillers) == 0
        assert all((li.rec_bytes[d] > 0) == (recs and bool(li.depth_mask_kept >> d & 1)) for d in range(5)) and (not rec_per or sum(li.rec_overflow) > 0)
        assert (li.phi_slots > 0 and li.phi_directory == 0 and rb.info().phi_slots == li.phi_slots) if phi_slots else (li.phi_slots == 0 and li.phi_directory == 1)
        assert all((li.entries[d] > 0) == bool(li.depth_mask_kept >> d & 1) for d in range(5))
    _run_indexed_checks(S, rb)


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


@pytest.mark.parametrize("recs", [False, True])
@pytest.mark.parametrize("fill_shift,super_shift,ks,depths,dir_runs,phi_per", [(6, 2, 5, 0, None, None), (4, 1, 3, 0x7, "1", "0.5"), (9, 5, 1, 0, "16", "4"),
                                                                                (5, 3, 5, 0x1F, "40", "9")])
def test_run_indexed_format2_fillers_and_super_counts(synth, fill_shift, super_shift, ks, depths, dir_runs, phi_per, recs):
    """Format 2 at 8-byte positions stores the LOW WORDS of {start, cum} and of the sampled positions; what makes that exact
    (rbg_dev.h DevRunTab2) is (a) filler entries wherever two entries of a table lie 2^fill_shift rows or more apart, (b)
    directory buckets no wider than that, (c) the rank's high part in the directory, (d) 64-bit super counts under the phi
    directory's 32-bit ones.  On a real index the distance is 2^30 rows and the super blocks 2^16 buckets: never met by a
    test-sized text.  RBG_RUN_FILL_SHIFT / RBG_PHI_SUPER_SHIFT shrink both so that this index is FULL of fillers (runs
    longer than the distance are cut into continuation pieces, gaps get empty runs, phi entries get shifted bases) and
    spans many super blocks -- and every query must still equal the oracle's.  dir_runs / phi_per: coarse directories on
    top (narrowing rounds over fillers), or fine ones (most buckets empty)."""
    S = synth
    ra.set_default_option(capi.OPT_POS_BYTES, 8)
    ra.set_default_option(capi.OPT_KMER_STEPS, ks)
    ra.set_default_option(capi.OPT_RUN_DEPTHS, depths)
    ra.set_default_option(capi.OPT_RUN_PHI, 1)    # (phi over the list of sampled positions: the structure that has fillers and super counts)
    ra.set_default_option(capi.OPT_RUN_REC, 2 if recs else 1)   # bucket records over the same filler-laden lists (dir_runs then sets THEIR bucket width)
    if recs and dir_runs:
        os.environ["RBG_RUN_REC_PER"] = dir_runs
    os.environ["RBG_RUN_FILL_SHIFT"] = str(fill_shift)
    os.environ["RBG_PHI_SUPER_SHIFT"] = str(super_shift)
    if dir_runs:
        os.environ["RBG_RANK_DIR_RUNS"] = dir_runs
    if phi_per:
        os.environ["RBG_PHI_DIR_PER"] = phi_per
    try:
        rb = _with_layout(capi.LAYOUT_RUNS, 48, lambda: ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0))
    finally:
        ra.set_default_option(capi.OPT_POS_BYTES, 0)
        ra.set_default_option(capi.OPT_KMER_STEPS, 5)
        ra.set_default_option(capi.OPT_RUN_DEPTHS, 0)
        ra.set_default_option(capi.OPT_RUN_PHI, 0)
        ra.set_default_option(capi.OPT_RUN_REC, 0)
        for k in ("RBG_RUN_FILL_SHIFT", "RBG_PHI_SUPER_SHIFT", "RBG_RANK_DIR_RUNS", "RBG_PHI_DIR_PER", "RBG_RUN_REC_PER"):
            os.environ.pop(k, None)
    li = rb.layout_info()
    assert li.run_fmt == 2 and li.fill_shift == fill_shift and rb.info().pos_bytes == 8
    kept = [d for d in range(5) if li.depth_mask_kept >> d & 1]
    assert sum(li.fillers) > 0 and (fill_shift > 6 or all(li.fillers[d] > 0 for d in kept)), list(li.fillers)   # tables with gaps beyond the distance
    assert li.phi_fillers > 0 and li.phi_entries == len(S.heads) + li.phi_fillers
    assert (S.n >> li.phi_dir_shift) >> super_shift > 2                   # several super blocks
    _run_indexed_checks(S, rb)


@pytest.mark.parametrize("depths", [0, 0x15, 0x1f])
def test_composition_spills_kept_depths_to_host_and_back(synth, depths, capfd):
    """k_compose.hip: when a depth's sweeps do not fit beside the kept depths made so far (r = 1e9 with 5 symbols per step),
    those wait in host memory and come back at the end.  RBG_COMPOSE_SPILL takes that path at test size: the index must be
    the same one (every query equal to the oracle's), with the same depth set."""
    S = synth
    ra.set_default_option(capi.OPT_RUN_DEPTHS, depths)
    os.environ["RBG_COMPOSE_SPILL"] = "1"
    os.environ["RBG_VERBOSE"] = "1"
    try:
        rb = _with_layout(capi.LAYOUT_RUNS, 48, lambda: ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0))
    finally:
        ra.set_default_option(capi.OPT_RUN_DEPTHS, 0)
        os.environ.pop("RBG_COMPOSE_SPILL", None)
        os.environ.pop("RBG_VERBOSE", None)
    assert "waits in host memory" in capfd.readouterr().err
    li = rb.layout_info()
    assert li.depths_composed == 5 and (not depths or li.depth_mask_kept == depths), (li.depths_composed, li.depth_mask_kept)
    _run_indexed_checks(S, rb)


def test_run_indexed_format1_limits_are_loud(synth, capfd):
    """Format 1 (rounds 2-3) has two width limits that used to bite without a word: 32-bit entry indices (a depth with 2^32
    entries was left out, and the depth forced in its place could already have been released) and a phi directory that
    vanished beyond 2 GiB / r >= 2^31 (phi then descends the sampled levels, several times slower).  Format 2 has
    neither.  Here both thresholds are lowered (RBG_RUN1_MAX_ENTRIES, RBG_PHI1_DIR_MAX_BYTES): the first is now an error
    that names the way out, the second is reported on stderr and by rbg_layout_info -- and the answers stay the oracle's."""
    S = synth
    ra.set_default_option(capi.OPT_RUN_FMT, 1)
    os.environ["RBG_RUN1_MAX_ENTRIES"] = "5000"
    try:
        with pytest.raises(ra.RbgError) as ei:
            _with_layout(capi.LAYOUT_RUNS, 48, lambda: ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0))
        assert ei.value.code == -4
        assert "format 2" in capfd.readouterr().err
        os.environ.pop("RBG_RUN1_MAX_ENTRIES")
        os.environ["RBG_PHI1_DIR_MAX_BYTES"] = "64"
        rb = _with_layout(capi.LAYOUT_RUNS, 48, lambda: ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0))
        assert "leaves the phi directory out" in capfd.readouterr().err
    finally:
        ra.set_default_option(capi.OPT_RUN_FMT, 2)
        os.environ.pop("RBG_RUN1_MAX_ENTRIES", None)
        os.environ.pop("RBG_PHI1_DIR_MAX_BYTES", None)
    li = rb.layout_info()
    assert li.run_fmt == 1 and li.phi_directory == 0 and li.phi_directory_dropped == 1 and li.rank_directories == 1
    o = orc.Oracle.from_runs(S.heads, S.lens, S.ssa, S.esa)
    seqs, off = ra.pack_reads(S.sample_reads(500, 50, seed=9, sub_rate=0.1, ragged=True))
    lo, hi, k = rb.find_range_w_toehold(seqs, off)
    wlo, whi, wk = o.find_range_w_toehold_batch(seqs, off)
    assert (lo == wlo).all() and (hi == whi).all() and (k == wk).all()
    loc_off, locs = rb.locs_at(lo, hi, k, MAXU)
    woff, wlocs = o.locs_at_batch(wlo, whi, wk, MAXU)
    assert (loc_off == woff).all() and (locs == wlocs).all()
    # the same index in format 2: nothing dropped, nothing to report
    with capi.default_option(capi.OPT_RUN_PHI, 1):
        rb2 = _with_layout(capi.LAYOUT_RUNS, 48, lambda: ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0))
    l2 = rb2.layout_info()
    assert l2.run_fmt == 2 and l2.phi_directory == 1 and l2.phi_directory_dropped == 0 and l2.depths_dropped_limit == 0
    rb.close(); rb2.close(); o.close()


@pytest.mark.parametrize("pos_bytes,rec,dir_runs", [(0, None, None), (8, None, "64"), (0, None, "64"), (0, "6", None), (8, "3", None), (0, "fmt2", None), (8, "fmt2", "64")])
def test_run_indexed_crowded_buckets(pos_bytes, rec, dir_runs):
    """Directory buckets with a hundred and more runs (k_runs.hip: narrowing rounds, one after the other when the
    directory is coarse -- dir_runs = RBG_RANK_DIR_RUNS; probes whose sixteen candidates all lie below the position;
    bucket records that overflow) beside buckets with none: a text that is 2 000 bases repeated 300 times, then
    1 500 x (one of A,C,G,T + the same 14-mer + 10 random bases) -- the rows of the suffixes that start with the 14-mer
    are consecutive and their BWT symbols change at nearly every row, while the average run is 39 rows long and sets
    the bucket width.  Quads of lanes at both position widths."""
    import naive
    rng = np.random.default_rng(7)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    block, x = acgt[rng.integers(0, 4, 2000)], acgt[rng.integers(0, 4, 14)]
    parts = [block] * 300 + [np.concatenate([acgt[[i % 4]], x, acgt[rng.integers(0, 4, 10)]]) for i in range(1500)]
    text = np.concatenate(parts + [np.array([1], np.uint8)])
    n, dense_at = len(text), 2000 * 300
    sa = naive.suffix_array(text)
    heads, lens, brk = naive.rle(naive.bwt_from_sa(text, sa))
    ssa, esa = naive.run_samples(sa, brk, n)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
    crowded = 0
    for c in b"ACGT":   # the directory's rule (rbg_capi.hip upload_tables_runs): at most dir_runs (4) runs per bucket on average
        st = starts[heads == c]
        sh = 0
        while len(st) * (2 << sh) <= float(dir_runs or 4) * n:
            sh += 1
        crowded = max(crowded, int(np.bincount(st >> sh).max()))
    assert crowded > (256 if dir_runs else 64), crowded   # one narrowing round at least; two with the coarse directory
    ra.set_default_option(capi.OPT_POS_BYTES, pos_bytes)
    if rec == "fmt2":      # format 2's bucket records: the crowded buckets overflow them and go through the run list (narrowed by the lane)
        ra.set_default_option(capi.OPT_RUN_REC, 2)
        if dir_runs is not None:
            os.environ["RBG_RUN_REC_PER"] = dir_runs
    elif rec is not None:
        os.environ["RBG_RANK_REC"] = rec
    else:
        ra.set_default_option(capi.OPT_RUN_REC, 1)
    if dir_runs is not None:
        os.environ["RBG_RANK_DIR_RUNS"] = dir_runs
    try:
        rb = _with_layout(capi.LAYOUT_RUNS, 48, lambda: ra.RowBowt.from_runs(heads, lens, ssa, esa, device=0))
    finally:
        ra.set_default_option(capi.OPT_POS_BYTES, 0)
        ra.set_default_option(capi.OPT_RUN_REC, 0)
        os.environ.pop("RBG_RANK_REC", None)
        os.environ.pop("RBG_RANK_DIR_RUNS", None)
        os.environ.pop("RBG_RUN_REC_PER", None)
    o = orc.Oracle.from_runs(heads, lens, ssa, esa)
    assert rb.info().rank_layout == capi.LAYOUT_RUNS and rb.info().pos_bytes == (pos_bytes or 4)
    reads = []
    for _ in range(6000):   # reads across the 14-mer's occurrences (every length, so that ranges end inside the crowded rows) ...
        a = dense_at + int(rng.integers(0, 1500 * 25 - 30))
        reads.append(text[a:a + int(rng.integers(1, 31))].tobytes())
    for _ in range(2000):   # ... and from the repeats (rows in buckets without a run)
        a = int(rng.integers(0, dense_at - 80))
        reads.append(text[a:a + int(rng.integers(1, 80))].tobytes())
    reads += [x.tobytes(), x[1:].tobytes(), x[:-1].tobytes(), b"A" + x.tobytes(), x.tobytes() + b"C"]
    seqs, off = ra.pack_reads(reads)
    lo, hi, k = rb.find_range_w_toehold(seqs, off)
    lo1, hi1 = rb.find_range(seqs, off)
    wlo, whi, wk = o.find_range_w_toehold_batch(seqs, off)
    assert (lo == wlo).all() and (hi == whi).all() and (k == wk).all()
    assert (lo1 == wlo).all() and (hi1 == whi).all()
    loc_off, locs = rb.locs_at(lo, hi, k, 20)
    woff, wlocs = o.locs_at_batch(wlo, whi, wk, 20)
    assert (loc_off == woff).all() and (locs == wlocs).all()
    rb.close()
    o.close()


def test_run_indexed_layout_goldens_and_budget_rule(small, simple_reads, error_reads, data_dir):
    """the reference's fixture through the run-indexed layout (goldens rb_tests.cpp:47-58,115-120), and the
    automatic choice: a budget below the single-symbol slot tables selects it by itself"""
    _rb, o = small
    rb = _with_layout(capi.LAYOUT_RUNS, 48, lambda: ra.load_rowbowt(os.path.join(data_dir, "small.fa"), ra.LoadRbwtFlag.SA | ra.LoadRbwtFlag.MA, device=0))
    assert rb.info().rank_layout == capi.LAYOUT_RUNS
    seqs, off = ra.pack_reads(simple_reads + error_reads)
    lo, hi, k = rb.find_range_w_toehold(seqs, off)
    wlo, whi, wk = o.find_range_w_toehold_batch(seqs, off)
    assert (lo == wlo).all() and (hi == whi).all() and (k == wk).all()
    assert (int(lo[0]), int(hi[0])) == (24279, 24280)              # rb_tests.cpp:115
    loc_off, locs = rb.locs_at(lo, hi, k)
    woff, wlocs = o.locs_at_batch(wlo, whi, wk)
    assert (loc_off == woff).all() and (locs == wlocs).all()
    assert locs[:2].tolist() == [20306, 286]                       # rb_tests.cpp:47-48
    hbm_runs = rb.info().hbm_bytes
    rb.close()
    ra.set_default_option(capi.OPT_HBM_BUDGET_MB, 1)               # small.fa's slot tables need more than 1 MB
    ra.set_default_option(capi.OPT_RANK_BUCKET_SHIFT, 0)
    try:
        rb2 = ra.load_rowbowt(os.path.join(data_dir, "small.fa"), ra.LoadRbwtFlag.SA, device=0)
    finally:
        ra.set_default_option(capi.OPT_HBM_BUDGET_MB, 0)
        ra.set_default_option(capi.OPT_RANK_BUCKET_SHIFT, -1)
    assert rb2.info().rank_layout == capi.LAYOUT_RUNS
    l2, h2, k2 = rb2.find_range_w_toehold(seqs, off)
    assert (l2 == wlo).all() and (h2 == whi).all() and (k2 == wk).all()
    rb2.close()
    assert hbm_runs < 3_000_000   # (run lists, samples, sampled levels and directories of five depths, each array rounded to 64 KB)


@pytest.mark.parametrize("pos_bytes", [4, 8])
def test_run_indexed_layout_built_on_the_device_equals_the_host_build(synth, pos_bytes):
    """The run-indexed layout adopts the k-mer levels where k_compose.hip left them and makes directories, sampled levels
    and 6-byte samples with kernels (k_build.hip: k_run_dirs, k_sample_keys, k_pack_samp48); RBG_RUNS_HOST_BUILD=1 is the
    round-2 way (levels copied out, everything built on the host, uploaded).  Same replica size, same answers."""
    S = synth
    ra.set_default_option(capi.OPT_POS_BYTES, pos_bytes)
    built = []
    try:
        for host in (False, True):
            if host:
                os.environ["RBG_RUNS_HOST_BUILD"] = "1"
            try:
                built.append(_with_layout(capi.LAYOUT_RUNS, 48, lambda: ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0)))
            finally:
                os.environ.pop("RBG_RUNS_HOST_BUILD", None)
    finally:
        ra.set_default_option(capi.OPT_POS_BYTES, 0)
    dev, host = built
    assert dev.info().rank_layout == host.info().rank_layout == capi.LAYOUT_RUNS and dev.info().kmer_steps == host.info().kmer_steps == 5
    assert abs(int(dev.info().hbm_bytes) - int(host.info().hbm_bytes)) <= 64 * 65536   # (the same arrays; allocations are rounded to 64 KB)
    reads = S.sample_reads(4000, 80, seed=23, sub_rate=0.1, ragged=True) + [b"", b"A", b"N", S.text[:300].tobytes(), S.text[-40:].tobytes()]
    seqs, off = ra.pack_reads(reads)
    a, b = dev.find_range_w_toehold(seqs, off), host.find_range_w_toehold(seqs, off)
    assert all((x == y).all() for x, y in zip(a, b))
    la, lb = dev.locs_at(*a, 50), host.locs_at(*b, 50)
    assert (la[0] == lb[0]).all() and (la[1] == lb[1]).all()
    o = orc.Oracle.from_runs(S.heads, S.lens, S.ssa, S.esa)
    w = o.find_range_w_toehold_batch(seqs, off)
    assert all((x == y).all() for x, y in zip(a, w))
    o.close()
    dev.close()
    host.close()


@pytest.mark.parametrize("pos_bytes,mask,kept,host_build", [(0, 0, 0x15, False), (0, 0x15, 0x15, False), (8, 0x11, 0x11, False), (0, 0x13, 0x13, True),
                                                             (8, 0x0A, 0x0B, False), (0, 0x1E, 0x1F, False), (8, 0x15, 0x15, True),
                                                             (0, 0x15, 0x15, "rec"), (8, 0x09, 0x09, "rec")])
def test_run_indexed_layout_sparse_depths(synth, pos_bytes, mask, kept, host_build):
    """RBG_OPT_RUN_DEPTHS: run lists for some of the k-mer depths only (bit d - 1; depth 1 always, nothing above the
    highest bit).  A step takes the longest stretch a kept depth covers (k_runs.hip, k_runs_seeds.hip pick_step), so the
    answers are those of every other layout -- the oracle's -- in less space.  Both ways of building the layout, and with
    bucket records ("rec")."""
    S = synth
    ra.set_default_option(capi.OPT_POS_BYTES, pos_bytes)
    if host_build == "rec":
        os.environ["RBG_RANK_REC"] = "8"          # bucket records (built on the host) beside the depth set
    elif host_build:
        os.environ["RBG_RUNS_HOST_BUILD"] = "1"
    try:
        with capi.default_option(capi.OPT_RUN_DEPTHS, 0x1F):
            full = _with_layout(capi.LAYOUT_RUNS, 48, lambda: ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0))
        with capi.default_option(capi.OPT_RUN_DEPTHS, mask):
            rb = _with_layout(capi.LAYOUT_RUNS, 48, lambda: ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0))
    finally:
        ra.set_default_option(capi.OPT_POS_BYTES, 0)
        os.environ.pop("RBG_RUNS_HOST_BUILD", None)
        os.environ.pop("RBG_RANK_REC", None)
    fi, info = full.info(), rb.info()
    runs_full = [fi.r, fi.pair_runs, fi.triple_runs, fi.quad_runs, fi.quint_runs]
    runs_kept = [info.r, info.pair_runs, info.triple_runs, info.quad_runs, info.quint_runs]
    assert info.rank_layout == capi.LAYOUT_RUNS and info.kmer_steps == kept.bit_length() and fi.kmer_steps == 5
    assert runs_kept == [x if kept >> d & 1 else 0 for d, x in enumerate(runs_full)]   # (rbg_info: the depths left out report no runs)
    if kept != 0x1F:
        assert info.hbm_bytes < fi.hbm_bytes
    o = orc.Oracle.from_runs(S.heads, S.lens, S.ssa, S.esa)
    reads = S.sample_reads(3000, 60, seed=9, sub_rate=0.12, ragged=True)
    reads += [b"", b"A", b"N", b"ACGTN", b"NACGT", b"ACNGT", b"AC", b"ACG", b"ACGT", b"ACGTA", b"ACGTAC", b"acgt", bytes([1]), bytes([0]),
              S.text[:500].tobytes(), S.text[:501].tobytes(), S.text[:502].tobytes(), S.text[:503].tobytes(), S.text[:504].tobytes(),
              S.text[-30:].tobytes(), S.text[-31:-1].tobytes(), S.text[-2:].tobytes()]
    seqs, off = ra.pack_reads(reads)
    wlo, whi, wk = o.find_range_w_toehold_batch(seqs, off)
    for b in (rb, full):
        lo, hi, k = b.find_range_w_toehold(seqs, off)
        lo1, hi1 = b.find_range(seqs, off)
        assert (lo == wlo).all() and (hi == whi).all() and (k == wk).all()
        assert (lo1 == wlo).all() and (hi1 == whi).all()
    with capi.default_option(capi.OPT_PACKED_READS, 0):   # the byte form of the kernels (reads cross as bytes)
        lo, hi, k = rb.find_range_w_toehold(seqs, off)
        assert (lo == wlo).all() and (hi == whi).all() and (k == wk).all()
    loc_off, locs = rb.locs_at(lo, hi, k, 30)
    woff, wlocs = o.locs_at_batch(wlo, whi, wk, 30)
    assert (loc_off == woff).all() and (locs == wlocs).all()
    ms, me, mo, mv = S.markers(wsize=10)
    rb.set_markers(ms, me, mo, mv)
    o.set_markers(ms, me, mo, mv)
    nseed, nmk = _check_marker_seeds(rb, o, reads[:300] + reads[-22:], 10, 1000)
    assert nseed > 330 and nmk > 20
    goff, glocs = rb.find_locs_greedy_seeding(*ra.pack_reads(reads[:200] + reads[-22:]), 10)
    for i, q in enumerate(reads[:200] + reads[-22:]):
        assert glocs[int(goff[i]):int(goff[i + 1])].tolist() == o.greedy_locate(q, 10)[0]
    sub = reads[:300] + reads[-22:]
    s3, o3 = ra.pack_reads(sub)
    for wsize, max_range in ((10, MAXU), (7, 4)):
        lo3, hi3, mk_off3, mk3 = rb.find_range_w_markers(s3, o3, wsize, max_range)
        got3 = split(mk_off3, mk3)
        for i, q in enumerate(sub):
            (wl, wh), wm = o.find_range_w_markers(q, wsize, max_range)
            assert (int(lo3[i]), int(hi3[i])) == (wl, wh) and got3[i] == wm, (i, q, wsize)
    # the copy made for another handle (rbg_replicate) carries the same depths
    rep = rb.replicate(0)
    ri = rep.info()
    assert ri.kmer_steps == info.kmer_steps and [ri.pair_runs, ri.triple_runs, ri.quad_runs, ri.quint_runs] == runs_kept[1:]
    lo, hi, k = rep.find_range_w_toehold(seqs, off)
    assert (lo == wlo).all() and (hi == whi).all() and (k == wk).all()
    rep.close()
    rb.close()
    full.close()
    o.close()


def test_run_indexed_layout_budget_leaves_middle_depths_out():
    """Over budget the run-indexed layout gives up the depths between the first and the deepest before the deepest
    itself (rbg_capi.hip upload): the step length stays, the space goes down, the answers stay.  (A synthetic run list of
    a million runs: the budget option counts MB.)"""
    rng = np.random.default_rng(41)
    heads, lens, ssa, esa, n = _random_run_index(rng, 1_000_000, 200)
    def build():   # (without the device ftab: the budget is about the run lists; phi over the list: phi slots are the budget's to give, too)
        with capi.default_option(capi.OPT_FTAB_K, 0), capi.default_option(capi.OPT_RUN_DEPTHS, depths[0]), capi.default_option(capi.OPT_RUN_PHI, 1), capi.default_option(capi.OPT_RUN_REC, 1):
            return _with_layout(capi.LAYOUT_RUNS, 48, lambda: ra.RowBowt.from_runs(heads, lens, ssa, esa, device=0))
    depths = [0x1F]   # asked for: all five, unless stated
    full = build()
    hb = {}
    for mask in (0x11, 0x17, 0x09):   # depths {1,5}, {1,2,3,5}, {1,4}
        depths[0] = mask
        x = build()
        hb[mask] = int(x.info().hbm_bytes)
        x.close()
    depths[0] = 0x1F
    fi = full.info()
    hb_full, d4 = int(fi.hbm_bytes), int(fi.hbm_bytes) - hb[0x17]
    assert d4 > (12 << 20) and hb[0x09] + (8 << 20) < hb[0x11] < hb[0x17], (hb, hb_full)   # (the estimate rounds up by a few MB)
    # half of depth 4 too much: depth 4 alone pays for it
    with capi.default_option(capi.OPT_HBM_BUDGET_MB, (hb_full - d4 // 2) >> 20):
        rb4 = build()
    i4 = rb4.info()
    assert i4.kmer_steps == 5 and i4.quad_runs == 0 and i4.triple_runs == fi.triple_runs > 0 and i4.pair_runs == fi.pair_runs > 0
    assert abs(int(i4.hbm_bytes) - hb[0x17]) < (1 << 20)
    # room for the first and the deepest and half of depth 4: depths 4, 3 and 2 go, in that order, and the deepest stays
    with capi.default_option(capi.OPT_HBM_BUDGET_MB, (hb[0x11] + d4 // 2) >> 20):
        rb = build()
    info = rb.info()
    assert info.rank_layout == capi.LAYOUT_RUNS and info.kmer_steps == 5 and info.quint_runs == fi.quint_runs > 0
    assert info.quad_runs == info.triple_runs == info.pair_runs == 0 and abs(int(info.hbm_bytes) - hb[0x11]) < (1 << 20)
    # less than the first and the deepest need: the deepest goes, and the one below it is stepped by again
    with capi.default_option(capi.OPT_HBM_BUDGET_MB, (hb[0x11] + hb[0x09]) // 2 >> 20):
        rb3 = build()
    i3 = rb3.info()
    assert i3.kmer_steps in (3, 4) and i3.quint_runs == 0 and i3.pair_runs == 0 and int(i3.hbm_bytes) <= hb[0x09] + (1 << 20)
    assert (i3.quad_runs == fi.quad_runs and i3.triple_runs == 0) if i3.kmer_steps == 4 else (i3.quad_runs == 0 and i3.triple_runs == fi.triple_runs)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    reads = [acgt[rng.integers(0, 4, int(rng.integers(1, 16)))].tobytes() for _ in range(20000)]
    seqs, off = ra.pack_reads(reads)
    want = full.find_range(seqs, off)
    assert int((want[1] >= want[0]).sum()) > 5000
    for b in (rb, rb4, rb3):   # (ranges only: the samples of a synthetic run list are not those of a text, so a toehold taken
        got = b.find_range(seqs, off)   #  through other depths is another number; test_run_indexed_layout_sparse_depths has the toeholds)
        assert all((x == y).all() for x, y in zip(got, want))
        got = b.find_range_w_toehold(seqs, off)
        assert (got[0] == want[0]).all() and (got[1] == want[1]).all()
        b.close()
    full.close()


@pytest.mark.parametrize("packed", [0, 1, 2])
def test_host_pointer_pipeline(small, packed, request):
    """the host-pointer calls as rbg_hostpath.hpp runs them: several double-buffered chunks (2.2 M short reads), reads
    crossing PCIe as bytes (0) or as 2-bit codes packed on the CPU (1 = default, 2 = always) with the reads that
    hold other symbols searched from their bytes afterwards, spans of one buffer instead of the packed layout,
    and calls too small to wake the worker threads -- same answers as the oracle every way"""
    rb, o = small
    ra.set_default_option(capi.OPT_PACKED_READS, packed)
    request.addfinalizer(lambda: ra.set_default_option(capi.OPT_PACKED_READS, 1))
    rng = np.random.default_rng(11 + packed)
    text = np.frombuffer(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "small.fa"), "rb").read().split(b"\n", 1)[1].replace(b"\n", b""), dtype=np.uint8)
    N = 2_200_000
    starts = rng.integers(0, len(text) - 40, N)
    lens = rng.integers(0, 33, N).astype(np.uint32)
    lens[rng.integers(0, N, 2000)] = 0                                     # empty reads
    begin = starts.astype(np.uint64)
    buf = text.copy()
    dirty = rng.integers(0, len(buf), 300)
    buf[dirty] = rng.choice(np.frombuffer(b"Nacgt\x01", dtype=np.uint8), len(dirty))   # some reads hold other symbols
    lo, hi, k = rb.find_range_spans(buf, begin, lens, toehold=True)
    # the same reads in the packed layout, through the oracle and through the packed-layout entry points
    idx = begin[:, None] + np.arange(32, dtype=np.uint64)[None, :]
    mask = np.arange(32)[None, :] < lens[:, None]
    seqs = buf[np.minimum(idx, len(buf) - 1)][mask]
    off = np.concatenate([[0], np.cumsum(lens.astype(np.uint64))]).astype(np.uint64)
    wlo, whi, wk = o.find_range_w_toehold_batch(seqs, off, nthreads=min(os.cpu_count() or 1, 64))
    assert (lo == wlo).all() and (hi == whi).all() and (k == wk).all()
    assert int(((whi < wlo)).sum()) > 100 and int((lens == 0).sum()) > 1000
    lo2, hi2 = rb.find_range_spans(buf, begin, lens)
    assert (lo2 == wlo).all() and (hi2 == whi).all()
    lo3, hi3, k3 = rb.find_range_w_toehold(seqs, off)
    assert (lo3 == wlo).all() and (hi3 == whi).all() and (k3 == wk).all()
    cnt = rb.count(seqs, off)
    assert (cnt == np.where(whi >= wlo, whi - wlo + 1, 0)).all()
    for n_small in (1, 2, 100, 5000):
        l4, h4 = rb.find_range(seqs[:int(off[n_small])], off[:n_small + 1])
        assert (l4 == wlo[:n_small]).all() and (h4 == whi[:n_small]).all()
    # many more chunks than staging buffers (every buffer reused several times, chunks handed back out of lockstep)
    os.environ["RBG_HOST_CHUNK_READS"] = "70001"
    try:
        lo5, hi5, k5 = rb.find_range_w_toehold(seqs, off)
        lo6, hi6 = rb.find_range_spans(buf, begin, lens)
    finally:
        del os.environ["RBG_HOST_CHUNK_READS"]
    assert (lo5 == wlo).all() and (hi5 == whi).all() and (k5 == wk).all() and (lo6 == wlo).all() and (hi6 == whi).all()
    # offsets that do not ascend are refused (the staging passes check them chunk by chunk before reading any byte)
    for where in (1, 4000, N // 2 + 12345, N):
        bad = off.copy()
        bad[where] = bad[where - 1] - 1 if bad[where - 1] else np.uint64(2**63)
        if where < N and bad[where + 1] >= bad[where] and bad[where] >= bad[where - 1]:
            continue
        with pytest.raises(ra.RbgError) as ei:
            rb.find_range(seqs, bad)
        assert ei.value.code == -4
    bad = off.copy()
    bad[0] = 1
    with pytest.raises(ra.RbgError):
        rb.count(seqs, bad)


@pytest.mark.parametrize("layout", [capi.LAYOUT_AUTO, capi.LAYOUT_RUNS])
def test_replicas_sharded_queries_and_rccl_counters(synth, layout):
    """More than one replica in one process (include/rbg.h "several GPUs"): rbg_replicate copies the device index
    peer to peer and re-points it -- onto the SAME device here when the box has one GPU, which exercises every
    relocation -- rbg_find_range_sharded splits a batch by rbg_shard_bounds, and the counters are reduced by
    RCCL (a one-rank clique on a single GPU; one rank per device when there are more)."""
    import torch
    S = synth
    rb = _with_layout(layout, 48, lambda: ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0))
    ms, me, mo, mv = S.markers(wsize=10)
    rb.set_markers(ms, me, mo, mv)
    o = orc.Oracle.from_runs(S.heads, S.lens, S.ssa, S.esa)
    o.set_markers(ms, me, mo, mv)
    ndev = torch.cuda.device_count()
    rep = rb.replicate(1 if ndev > 1 else 0)
    assert rep.info().hbm_bytes == rb.info().hbm_bytes and rep.info().rank_layout == rb.info().rank_layout
    with pytest.raises(ra.RbgError):
        rep.replicate(0)                       # replicas are made from the primary
    with pytest.raises(ra.RbgError):
        rep.set_markers(ms, me, mo, mv)        # ... and everything is attached before replicating
    reads = S.sample_reads(2001, 70, seed=77, sub_rate=0.2, ragged=True) + [b"", b"ACGTN"]
    seqs, off = ra.pack_reads(reads)
    wlo, whi, wk = o.find_range_w_toehold_batch(seqs, off)
    woff, wlocs = o.locs_at_batch(wlo, whi, wk)
    # the replica answers everything the primary does
    lo, hi, k = rep.find_range_w_toehold(seqs, off)
    assert (lo == wlo).all() and (hi == whi).all() and (k == wk).all()
    loc_off, locs = rep.locs_at(lo, hi, k)
    assert (loc_off == woff).all() and (locs == wlocs).all()
    mk_off, mk = rep.markers_at(lo, hi)
    got = split(mk_off, mk)
    for i in range(0, len(reads), 11):
        assert got[i] == o.markers_at(int(lo[i]), int(hi[i]))
    _check_marker_seeds(rep, o, reads[:200], 10, 1000)
    # sharded over both replicas (and over one: the degenerate G = 1 path)
    for reps in ([rb, rep], [rb], [rep, rb, rep]):
        rb.counters_reset(); rep.counters_reset()
        lo2, hi2, k2 = capi.find_range_sharded(reps, seqs, off, toehold=True)
        assert (lo2 == wlo).all() and (hi2 == whi).all() and (k2 == wk).all()
        lo3, hi3 = capi.find_range_sharded(reps, seqs, off)
        assert (lo3 == wlo).all() and (hi3 == whi).all()
        tot = rb.counters().astype(np.int64) + rep.counters().astype(np.int64)
        assert tot[0] == 2 * len(reads) and tot[1] == 2 * int((whi >= wlo).sum())
    for g, G in ((0, 1), (0, 3), (2, 3), (6, 7)):
        assert capi.shard_bounds(len(reads), g, G) == shard_bounds(len(reads), g, G)
    # RCCL: one clique per process over distinct devices
    rb.counters_reset(); rep.counters_reset()
    rb.find_range(seqs, off)
    want0 = rb.counters()
    if ndev > 1:
        rep.find_range(seqs, off)
        red = capi.counters_allreduce_local([rb, rep])
        assert (red == 2 * want0).all()
    red1 = capi.counters_allreduce_local([rb])
    assert (red1 == want0).all() and int(red1[0]) == len(reads)
    # the clique of a device set is made once and kept (rbg_comm_cache_clear drops it; the next call makes a new one)
    import time
    t0 = time.perf_counter(); capi.counters_allreduce_local([rb]); t_again = time.perf_counter() - t0
    assert (capi.counters_allreduce_local([rb]) == want0).all()
    assert ra.lib().rbg_comm_cache_clear() == 0
    t0 = time.perf_counter(); red2 = capi.counters_allreduce_local([rb]); t_fresh = time.perf_counter() - t0
    assert (red2 == want0).all()
    print(f"counters all-reduce: {t_again * 1e3:.2f} ms with the kept clique, {t_fresh * 1e3:.2f} ms making one")
    with pytest.raises(ra.RbgError):
        capi.counters_allreduce_local([rb, rb])   # the same device twice is not a clique
    # several replicas at once (rbg_replicate_many: the peer copies of all targets are in flight together)
    many = rb.replicate_many([1 if ndev > 1 else 0, 0, (2 if ndev > 2 else 0)])
    assert len(many) == 3 and all(r.info().hbm_bytes == rb.info().hbm_bytes for r in many)
    for r in many:
        lo4, hi4, k4 = r.find_range_w_toehold(seqs, off)
        assert (lo4 == wlo).all() and (hi4 == whi).all() and (k4 == wk).all()
        o4, l4 = r.locs_at(lo4, hi4, k4)
        assert (o4 == woff).all() and (l4 == wlocs).all()
    lo5, hi5, k5 = capi.find_range_sharded(many, seqs, off, toehold=True)
    assert (lo5 == wlo).all() and (hi5 == whi).all() and (k5 == wk).all()
    with pytest.raises(ra.RbgError):
        rb.replicate_many([0, 4096])              # all or nothing: a bad device leaves no replica behind
    for r in many:
        r.close()
    rep.close()
    rb.close()
    o.close()


@pytest.mark.parametrize("fk", [0, -1, 1, 3, 7])
def test_ftab_is_result_neutral(synth, fk):
    """The device ftab (rowbowt.hpp:124-125, :726-758) changes no answer, whatever its word length; reads
    shorter than the word, reads with non-ACGT symbols inside the word, and absent words included."""
    S = synth
    ra.set_default_option(capi.OPT_FTAB_K, fk)
    try:
        rb = ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0)
    finally:
        ra.set_default_option(capi.OPT_FTAB_K, -1)
    assert rb.info().ftab_k == (fk if fk >= 0 else 5)  # automatic: 4^k <= n/16 for n = 32 081
    o = orc.Oracle.from_runs(S.heads, S.lens, S.ssa, S.esa)
    reads = S.sample_reads(2000, 40, seed=8, sub_rate=0.3, ragged=True)
    reads += [b"", b"A", b"AC", b"ACG", b"ACGTACG", b"ACGTACGN", b"NACGTACG", b"ACGNACGT", b"TTTTTTTTTTTT", bytes([1]) + b"ACGTACG",
              S.text[:7].tobytes(), S.text[:8].tobytes(), S.text[-9:-1].tobytes(), S.text[-8:].tobytes()]
    seqs, off = ra.pack_reads(reads)
    rb.counters_reset()
    lo, hi, k = rb.find_range_w_toehold(seqs, off)
    wlo, whi, wk = o.find_range_w_toehold_batch(seqs, off)
    assert (lo == wlo).all() and (hi == whi).all() and (k == wk).all()
    clo, chi = rb.find_range(seqs, off)
    assert (clo == wlo).all() and (chi == whi).all()
    assert int(rb.counters()[0]) == 2 * len(reads)  # building the table left no trace in the counters
    rb.close()
    o.close()


def test_concurrent_queries_one_index(synth):
    """The reference calls const query methods concurrently on one RowBowt (rb_markers.cpp:321-326);
    concurrent host-pointer calls on one rbg_index must be independent (per-thread streams)."""
    import threading
    S = synth
    rb = ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0)
    o = orc.Oracle.from_runs(S.heads, S.lens, S.ssa, S.esa)
    jobs = []
    for t in range(6):
        reads = S.sample_reads(3000 + 500 * t, 64, seed=100 + t, sub_rate=0.1, ragged=bool(t % 2))
        seqs, off = ra.pack_reads(reads)
        wlo, whi, wk = o.find_range_w_toehold_batch(seqs, off, nthreads=4)
        woff, wlocs = o.locs_at_batch(wlo, whi, wk, nthreads=4)
        jobs.append((seqs, off, wlo, whi, wk, woff, wlocs))
    errors = []

    def worker(job):
        seqs, off, wlo, whi, wk, woff, wlocs = job
        try:
            for _ in range(5):
                lo, hi, k = rb.find_range_w_toehold(seqs, off)
                loc_off, locs = rb.locs_at(lo, hi, k)
                clo, chi = rb.find_range(seqs, off)
                if not ((lo == wlo).all() and (hi == whi).all() and (k == wk).all() and (loc_off == woff).all()
                        and (locs == wlocs).all() and (clo == wlo).all() and (chi == whi).all()):
                    errors.append("mismatch")
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    rb.counters_reset()
    threads = [threading.Thread(target=worker, args=(j,)) for j in jobs]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]
    # the device counters saw every call exactly once: 5 rounds x (toehold search + count search), 5 x locate
    c = rb.counters()
    assert int(c[0]) == 10 * sum(len(j[1]) - 1 for j in jobs) and int(c[3]) == 5 * sum(int(j[5][-1]) for j in jobs)
    rb.close()
    o.close()


def test_one_read_calls_from_threads_are_combined(synth):
    """An unmodified threaded caller of the reference's one-query methods (rb_markers.cpp:318-535): twelve threads each
    asking ONE read per call -- find_range, count, find_range_w_toehold, get_markers_greedy_seeding with two different
    parameter sets -- get the oracle's answers, and the calls are served by fewer, batched launches (rbg_combine_stats)."""
    import threading
    S = synth
    rb = ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0)
    ms, me, mo, mv = S.markers(wsize=10)
    rb.set_markers(ms, me, mo, mv)
    o = orc.Oracle.from_runs(S.heads, S.lens, S.ssa, S.esa)
    o.set_markers(ms, me, mo, mv)
    T, per = 12, 120
    reads = S.sample_reads(T * per, 70, seed=4242, sub_rate=0.2, ragged=True)
    reads[5] = b""
    reads[17] = b"ACGTNACGT"
    seqs, off = ra.pack_reads(reads)
    wlo, whi, wk = o.find_range_w_toehold_batch(seqs, off)
    want_seeds = {}
    for ws, mr in ((10, 1000), (7, 50)):
        so, sd, mk = rb.get_markers_greedy_seeding(seqs, off, ws, mr)     # the batched call (itself checked against the oracle elsewhere)
        want_seeds[(ws, mr)] = (so, sd, mk)
    _check_marker_seeds(rb, o, reads[:60], 10, 1000)
    l0, r0 = rb.combine_stats()
    errors = []

    def worker(t):
        try:
            for i in range(t * per, (t + 1) * per):
                q = np.frombuffer(reads[i], dtype=np.uint8)
                o1 = np.array([0, len(q)], dtype=np.uint64)
                kind = (i + t) % 4
                if kind == 0:
                    lo, hi = rb.find_range(q, o1)
                    ok = (int(lo[0]), int(hi[0])) == (int(wlo[i]), int(whi[i]))
                elif kind == 1:
                    c = rb.count(q, o1)
                    ok = int(c[0]) == (int(whi[i]) - int(wlo[i]) + 1 if whi[i] >= wlo[i] else 0)
                elif kind == 2:
                    lo, hi, k = rb.find_range_w_toehold(q, o1)
                    ok = (int(lo[0]), int(hi[0]), int(k[0])) == (int(wlo[i]), int(whi[i]), int(wk[i]))
                else:
                    ws, mr = ((10, 1000), (7, 50))[t % 2]
                    so, sd, mk = rb.get_markers_greedy_seeding(q, o1, ws, mr)
                    wso, wsd, wmk = want_seeds[(ws, mr)]
                    a, b = int(wso[i]), int(wso[i + 1])
                    ok = int(so[1]) == b - a and len(sd) == b - a
                    if ok and b > a:
                        m0 = int(wsd[a, 4])
                        ok = (sd[:, :4] == wsd[a:b, :4]).all() and (sd[:, 4:] == wsd[a:b, 4:] - np.uint64(m0)).all() \
                            and (mk == wmk[m0:int(wsd[b - 1, 5])]).all()
                if not ok:
                    errors.append((t, i, kind))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:5]
    l1, r1 = rb.combine_stats()
    assert r1 - r0 == T * per and 0 < l1 - l0 <= r1 - r0
    print(f"combined: {r1 - r0} one-read calls in {l1 - l0} launches ({(r1 - r0) / (l1 - l0):.1f} per launch)")
    rb.close()
    o.close()


from kseq_model import kseq_model as _kseq_model  # noqa: E402


def test_cli_parser_matches_kseq(data_dir, tmp_path, small):
    rb, o = small
    t = open(os.path.join(data_dir, "small.fa"), "rb").read().split(b"\n", 1)[1].replace(b"\n", b"")
    s1, s2, s3 = t[100:160], t[500:530], t[900:1000]
    blob = (b"junk before the first header\n>multi line\tcomment here\n" + s1[:20] + b"\n" + s1[20:45] + b"\n\n" + s1[45:] + b"\n"
            b"@fq1 desc\r\n" + s2 + b"\r\n+\r\n" + b"I" * len(s2) + b"\r\n"
            b">with space in seq\n" + s3[:10] + b" " + s3[10:] + b"\n"
            b"@fq2\n" + s3[:50] + b"\n" + s3[50:] + b"\n+fq2\n" + b">" * 50 + b"\n" + b"@" * 50 + b"\n"
            b">empty\n>last_no_newline\n" + s1)
    fq = tmp_path / "weird.fx"
    fq.write_bytes(blob)
    recs, err = _kseq_model(blob)
    assert err == -1 and [r[0] for r in recs] == [b"multi", b"fq1", b"with", b"fq2", b"empty", b"last_no_newline"]
    assert recs[0][1] == s1 and recs[1][1] == s2 and recs[3][1] == s3 and recs[4][1] == b""
    rc, out, errtxt = _run_cli([os.path.join(data_dir, "small.fa"), str(fq)])
    assert rc == 0, errtxt
    want = ""
    for name, seq in recs:
        lo, hi = o.find_range(seq)
        want += f"{name.decode()} ({lo},{hi}), count={(hi - lo + 1) % 2**64}\n"
    assert out == want
    assert "(1,0), count=0" in out.splitlines()[2]  # the blank inside the sequence is kept, as kseq does


@pytest.mark.parametrize("fmt", ["fasta", "fastq"])
@pytest.mark.parametrize("gz", [False, True])
def test_cli_many_windows(data_dir, tmp_path, small, fmt, gz):
    """an input several windows long (rb_align --window-mb 1; more than 3 MB of records), plain (memory-mapped) and gzip (zlib,
    with the unfinished record carried from window to window): every record answered once, in order, as
    rb_align.cpp:176-191 prints it.  FASTA is the case where a window ends right after the next record's '>' has been
    consumed (kseq.h:195-199), which the zlib path once mishandled."""
    import gzip
    rb, o = small
    t = open(os.path.join(data_dir, "small.fa"), "rb").read().split(b"\n", 1)[1].replace(b"\n", b"")
    rng = np.random.default_rng(12)
    recs, blob = [], bytearray()
    for i in range(36000):
        a, m = int(rng.integers(0, len(t) - 160)), int(rng.integers(20, 150))
        seq = bytearray(t[a:a + m])
        if rng.random() < 0.2:
            seq[int(rng.integers(m))] = ord("ACGT"[int(rng.integers(4))])
        seq = bytes(seq)
        recs.append((b"r%d" % i, seq))
        if fmt == "fasta":
            blob += b">r%d some text\n" % i + seq[:60] + b"\n" + (seq[60:] + b"\n" if len(seq) > 60 else b"")
        else:
            blob += b"@r%d\n" % i + seq + b"\n+\n" + b"I" * len(seq) + b"\n"
    assert len(blob) > (3 << 20)
    path = tmp_path / ("reads." + fmt + (".gz" if gz else ""))
    if gz:
        with gzip.open(path, "wb") as f:
            f.write(bytes(blob))
    else:
        path.write_bytes(bytes(blob))
    rc, out, errtxt = _run_cli(["--window-mb", "1", os.path.join(data_dir, "small.fa"), str(path)])
    assert rc == 0, errtxt
    seqs, off = ra.pack_reads([r[1] for r in recs])
    wlo, whi = o.find_range_batch(seqs, off) if hasattr(o, "find_range_batch") else o.find_range_w_toehold_batch(seqs, off)[:2]
    lines = out.splitlines()
    assert len(lines) == len(recs)
    for i in (0, 1, 5000, 11000, 35999):
        assert lines[i] == f"r{i} ({int(wlo[i])},{int(whi[i])}), count={(int(whi[i]) - int(wlo[i]) + 1) % 2**64}"
    want = "".join(f"r{i} ({int(wlo[i])},{int(whi[i])}), count={(int(whi[i]) - int(wlo[i]) + 1) % 2**64}\n" for i in range(len(recs)))
    assert out == want


def test_full_size_properties_and_parity_sample():
    """BASELINE.json's size (n = 2.0e9, r = 3.7e7, 10 M x 100 bp reads) through the size-independent
    properties and an oracle sample: one bench.py step in a subprocess (about a minute and a half: the
    index synthesis dominates)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                        "--check-reads", "5000", "--property-reads", "300000", "--markers"],
                       capture_output=True, timeout=1500, cwd=root)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    d = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert d["config"]["index"]["n"] > 2_000_000_000 and d["config"]["index"]["r"] > 30_000_000
    props = d["properties_full_size"]
    assert props["reads"] == 300000 and props["locations"] > 5_000_000
    assert all(props[k] for k in ("every_location_is_an_occurrence", "locations_distinct", "occ_equals_range_width", "empty_is_{1,0}"))
    assert d["parity"]["bit_exact_vs_oracle"] and d["parity"]["locs_checked"] > 100_000
    assert d["markers"]["parity"]["bit_exact_vs_oracle"] and d["markers"]["marker_seeds"]["parity"]["bit_exact_vs_oracle"]
    c = d["counters"]
    assert c["reads"] == 10_000_000 and c["sum_occ"] == c["sum_locs"] > 300_000_000   # sum of range widths == locations written


@pytest.mark.parametrize("layout", ["slots", "runs"])
def test_pangenome_stream_true_bwt_beyond_32_bits(layout):
    """BASELINE.json configs[3]'s single-GPU shape under the driver's own test run: a TRUE BWT with n = 4.4e9 > 2^32
    (rowbowt_amd/tools/pangenome_bwt.py: run heads, lengths and both samples of every run derived from the text's
    structure), 150 bp reads generated on the device, streamed count+locate in batches, through
    tools/pangenome_stream.py in a subprocess -- both layouts.  Ranges, toeholds (the k-mer steps' re-sampled ones
    included, some of them above 2^32: rowbowt.hpp:555-573 over toehold_sa.hpp:56-72), locations and the count-only
    kernel bit-exact against the oracle on 5 000 reads; the size-independent properties on 100 000."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (the run-indexed leg samples its reads and checks its properties from the pangenome's STRUCTURE, as the n = 3e11 run of
    #  profiles/r04_pangenome_stream_r1e9.json has to: rbg_sample_reads_pangenome_dev, pangenome_bwt.TextView)
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "pangenome_stream.py"), "--L", "44000000", "--H", "100",
                        "--total-reads", "20000000", "--reads", "5000000", "--check-reads", "5000", "--property-reads", "100000",
                        "--layout", layout, "--gpus", "1", "--implicit-text", "on" if layout == "runs" else "off"],
                       capture_output=True, timeout=1200, cwd=root)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    d = json.loads(p.stdout.decode().strip().splitlines()[-1])
    ix = d["config"]["index"]
    assert ix["n"] > (1 << 32) and ix["true_bwt"] and ix["pos_bytes"] == 8 and ix["rank_layout"] == {"slots": 1, "runs": 2}[layout]
    assert ix["symbols_per_gather"] >= 4 and d["n_gpus"] == 1
    par = d["parity"]
    assert par["reads_checked"] == 5000 and par["bit_exact_vs_oracle"] and par["count_only_kernel_bit_exact"]
    assert par["toeholds_above_2^32"] > 20 and par["locs_checked"] > 100_000
    props = d["properties"]
    assert props["reads"] == 100000 and props["locations"] > 2_000_000
    assert all(props[k] for k in ("unmutated_reads_all_found", "empty_is_{1,0}", "every_location_is_an_occurrence", "locations_distinct",
                                  "occ_equals_range_width", "own_position_reported"))
    c = d["counters"]
    assert c["reads"] == 20_000_000 and c["sum_occ"] == c["sum_locs"] > 500_000_000
    if layout == "runs":
        li = ix["layout_info"]
        assert li["run_fmt"] == 2 and li["depths_dropped_budget"] == 0 and li["depths_dropped_limit"] == 0
        assert (li["rank_directories"] == 1) != (sum(li["rec_bytes"]) > 0)   # ranks: directories over the run lists, or bucket records
        assert (li["phi_directory"] == 1) != (li["phi_slots"] > 0)      # phi: the list of sampled positions with its directory, or slots of about n / r rows
    print(f"n = {ix['n']:.3e}, {layout}: {d['value']:.3e} reads/s streamed, {ix['hbm_bytes'] / 1e9:.1f} GB replica")


@pytest.mark.gpu
def test_reads_sampled_from_the_structure_equal_reads_sampled_from_the_text():
    """rbg_sample_reads_pangenome_dev (reads from base sequence + sites + allele matrix) == rbg_sample_reads_dev (reads from the
    materialised text), byte for byte and start for start, for the same seed -- so the n = 3e11 stream, whose text fits no
    GPU, streams the reads the smaller runs stream.  More than 255 haplotypes as well (16-bit haplotype ranks in the builder)."""
    import torch
    from rowbowt_amd.tools import pangenome_bwt as pb
    dev = torch.device("cuda", 0)
    Lb = ra.lib()
    for L, H, rate, m in ((20000, 7, 0.03, 150), (5000, 300, 0.05, 100), (4000, 3, 0.0, 64)):
        pg = pb.make_pangenome(L, H, rate, 5, dev)
        text = pb.materialize_text(pg)
        tv = pb.TextView(pg)
        N = 20000
        a, b = (torch.zeros(N * m + 32, dtype=torch.uint8, device=dev) for _ in range(2))
        oa, ob = (torch.empty(N + 1, dtype=torch.int64, device=dev) for _ in range(2))
        sa, sb = (torch.empty(N, dtype=torch.int64, device=dev) for _ in range(2))
        st = torch.cuda.current_stream().cuda_stream
        assert Lb.rbg_sample_reads_dev(text.data_ptr(), pg["unit"], H, L, m, 77, 12345, N, 200000, a.data_ptr(), oa.data_ptr(), sa.data_ptr(), st) == 0
        assert Lb.rbg_sample_reads_pangenome_dev(tv.base_b.data_ptr(), tv.sites.data_ptr() if tv.S else None, tv.alt_b.data_ptr() if tv.S else None,
                                                 tv.G.data_ptr() if tv.S else None, tv.S, tv.site_dir.data_ptr() if tv.S else None, tv.site_dir_shift,
                                                 pg["unit"], H, L, m, 77, 12345, N, 200000, b.data_ptr(), ob.data_ptr(), sb.data_ptr(), st) == 0
        torch.cuda.synchronize()
        assert torch.equal(a, b) and torch.equal(oa, ob) and torch.equal(sa, sb)
        b.zero_()   # without the site directory: the search over all sites
        assert Lb.rbg_sample_reads_pangenome_dev(tv.base_b.data_ptr(), tv.sites.data_ptr() if tv.S else None, tv.alt_b.data_ptr() if tv.S else None,
                                                 tv.G.data_ptr() if tv.S else None, tv.S, None, 0, pg["unit"], H, L, m, 77, 12345, N, 200000, b.data_ptr(),
                                                 ob.data_ptr(), sb.data_ptr(), st) == 0
        torch.cuda.synchronize()
        assert torch.equal(a, b)
        pos = torch.randint(0, pg["n"], (100000,), device=dev)
        assert torch.equal(tv.at(pos), text[pos])
