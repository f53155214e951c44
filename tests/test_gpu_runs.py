"""GPU parity (through the C-ABI, bit-exact against the oracle; needs an MI355X): the run-indexed layout (space proportional to r): depth sets, records and directories, fillers, crowded buckets, budget rule."""
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


def default_depth_mask(K):
    """rbg_capi.hip default_depth_mask: the deepest depth, half of it, a quarter of it ... and 1"""
    m, d = 1, K
    while d >= 1:
        m |= 1 << (d - 1)
        d //= 2
    return m


@pytest.mark.parametrize("pos_bytes,fk,ks,mode,all_depths",
                         [(0, -1, 5, "dir", False), (8, -1, 5, "dir", False), (8, 0, 1, "dir", False), (0, 3, 3, "dir", False), (8, 3, 2, "dir", False), (0, 0, 4, "dir", True),
                          (8, -1, 4, "dir", True), (0, -1, 5, "phislots", False), (8, -1, 5, "phislots", False), (8, 0, 3, "phislots", False),
                          (0, -1, 5, "rec", False), (8, -1, 5, "rec", False), (8, 0, 4, "rec", True), (0, 3, 1, "rec", False),
                          (0, -1, 5, "rec+phislots", False), (8, -1, 5, "rec+phislots", False), (8, 3, 2, "rec+phislots", False),
                          # depths 6-8: the tables' records come from the global array (rbg_dev.h kLdsRunDepth), not from LDS
                          (0, -1, 8, "rec", False), (8, -1, 8, "rec", False), (0, -1, 8, "dir", False), (8, 0, 8, "dir", True), (0, 0, 8, "rec", True),
                          (0, -1, 6, "rec", False), (8, 3, 6, "dir", True), (0, -1, 7, "dir", False), (8, -1, 7, "rec+phislots", True), (0, 3, 8, "rec+phislots", False),
                          (8, -1, 8, "phislots", False),
                          # records for the deepest depth only (RBG_OPT_RUN_REC_DEPTHS: what an r = 1e9 index has room for), directories below it
                          (0, -1, 5, "rec@deepest", False), (8, -1, 8, "rec@deepest", False), (8, 0, 4, "rec@deepest+phislots", True), (0, 3, 7, "rec@deepest", False)])
def test_run_indexed_layout(synth, pos_bytes, fk, ks, mode, all_depths):
    """RBG_LAYOUT_RUNS (k_runs.hip): space proportional to r, rank and phi as predecessor searches over the run lists
    (rle_string.hpp:131-161, toehold_sa.hpp:56-72) by the lane that owns the query, k-mer steps of up to ks symbols through the
    depths' tables -- same answers as the slot tables, i.e. as the oracle, on every read shape of test_synth_all_paths.
    mode: "dir" = per-table directories over the run lists, phi over the list of sampled positions; "phislots" = phi SLOTS
    (RBG_OPT_RUN_PHI = 2: the slot layout's direct-addressed phi records at buckets of about n / r rows); "rec" = BUCKET RECORDS
    instead of the rank directories (RBG_OPT_RUN_REC = 2: one aligned 64-byte record per bucket, fetched by quads of lanes)."""
    S = synth
    phi_slots, recs, deepest_only = "phislots" in mode, "rec" in mode, "rec@deepest" in mode
    ra.set_default_option(capi.OPT_RUN_PHI, 2 if phi_slots else 1)
    ra.set_default_option(capi.OPT_RUN_REC, 2 if recs else 1)
    ra.set_default_option(capi.OPT_RUN_REC_DEPTHS, 1 << (ks - 1) if deepest_only else 0)
    ra.set_default_option(capi.OPT_POS_BYTES, pos_bytes)
    ra.set_default_option(capi.OPT_FTAB_K, fk)
    ra.set_default_option(capi.OPT_KMER_STEPS, ks)
    # (compact records hold eleven entries: at the default 2.5 per bucket none of this index's buckets overflows; the 8-byte variants
    #  take buckets of about nine entries so that overflowing records -- pivots, then the run list -- are met here too)
    rec_per = "9" if recs and pos_bytes == 8 else None
    if rec_per:
        os.environ["RBG_RUN_REC_PER"] = rec_per
    if all_depths:   # run lists at every depth (otherwise the default: the deepest, half of it, ... 1)
        ra.set_default_option(capi.OPT_RUN_DEPTHS, (1 << ks) - 1)
    try:
        rb = _with_layout(capi.LAYOUT_RUNS, lambda: ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0))
    finally:
        ra.set_default_option(capi.OPT_RUN_PHI, 0)
        ra.set_default_option(capi.OPT_RUN_REC, 0)
        ra.set_default_option(capi.OPT_RUN_REC_DEPTHS, 0)
        ra.set_default_option(capi.OPT_RUN_DEPTHS, 0)
        ra.set_default_option(capi.OPT_POS_BYTES, 0)
        ra.set_default_option(capi.OPT_FTAB_K, -1)
        ra.set_default_option(capi.OPT_KMER_STEPS, DEFAULT_KMER_STEPS)
        os.environ.pop("RBG_RUN_REC_PER", None)
    info = rb.info()
    assert info.rank_layout == capi.LAYOUT_RUNS and info.kmer_steps == ks and info.pos_bytes == (pos_bytes or 4)
    assert info.rank_slots == 0 and (info.phi_slots == 0) != phi_slots
    want_mask = (1 << ks) - 1 if all_depths else default_depth_mask(ks)
    D = capi.MAX_KMER_DEPTH
    assert [d + 1 for d in range(D) if info.depth_runs[d]] == [d + 1 for d in range(D) if want_mask >> d & 1]
    assert info.depth_runs[0] == info.r and list(info.depth_runs[1:5]) == [info.pair_runs, info.triple_runs, info.quad_runs, info.quint_runs]
    li = rb.layout_info()
    assert li.run_fmt == 2 and li.depths_dropped_budget == 0 and li.depth_mask_kept == want_mask and li.depths_composed == ks
    with_dirs = not recs or (deepest_only and want_mask != 1 << (ks - 1))   # some kept depth answers its ranks through a directory
    assert li.rank_directories == (1 if with_dirs else 0) and li.phi_entries == len(S.heads) and sum(li.fillers) == 0
    rec_mask = 0 if not recs else (1 << (ks - 1)) if deepest_only else li.depth_mask_kept
    assert all((li.rec_bytes[d] > 0) == bool(rec_mask >> d & 1) for d in range(D)) and (not rec_per or deepest_only or sum(li.rec_overflow) > 0)
    assert (li.phi_slots > 0 and li.phi_directory == 0 and rb.info().phi_slots == li.phi_slots) if phi_slots else (li.phi_slots == 0 and li.phi_directory == 1)
    assert all((li.entries[d] > 0) == bool(li.depth_mask_kept >> d & 1) for d in range(D))
    _run_indexed_checks(S, rb)


@pytest.mark.parametrize("uniform", ["0", "1"])
@pytest.mark.parametrize("pos_bytes,ks,all_depths,rec_per", [(0, 8, False, None), (8, 8, False, "9"), (0, 6, True, None), (8, 7, False, None), (0, 8, True, "9")])
def test_uniform_directories_of_the_deepest_depth(synth, uniform, pos_bytes, ks, all_depths, rec_per, monkeypatch):
    """The deepest kept depth beyond the LDS-staged ones may get UNIFORM geometry (rbg_dev.h DevIndex::run_uni_*: one bucket shift and one record count for all of
    its tables, so that a step computes its table's hot word instead of reading it from the global array).  RBG_RUN_UNIFORM=1 forces it -- also where the tables'
    sizes differ and buckets overflow their records (rec_per 9: pivots, then the run list) --, =0 forbids it; the answers are the oracle's either way, on every
    read shape of test_synth_all_paths, at both position widths, with the default depth set and with every depth kept."""
    S = synth
    monkeypatch.setenv("RBG_RUN_UNIFORM", uniform)
    if rec_per:
        monkeypatch.setenv("RBG_RUN_REC_PER", rec_per)
    with capi.default_option(capi.OPT_RUN_PHI, 2), capi.default_option(capi.OPT_RUN_REC, 2), capi.default_option(capi.OPT_POS_BYTES, pos_bytes), \
            capi.default_option(capi.OPT_KMER_STEPS, ks), capi.default_option(capi.OPT_RUN_DEPTHS, (1 << ks) - 1 if all_depths else 0):
        rb = _with_layout(capi.LAYOUT_RUNS, lambda: ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0))
    info, li = rb.info(), rb.layout_info()
    assert info.rank_layout == capi.LAYOUT_RUNS and info.kmer_steps == ks and li.rec_bytes[ks - 1] > 0 and li.rank_directories == 0
    assert not rec_per or sum(li.rec_overflow) > 0
    _run_indexed_checks(S, rb)


def test_default_load_takes_eight_symbols_per_step_on_the_run_indexed_layout(synth):
    """RBG_OPT_KMER_STEPS defaults to 8 and RBG_OPT_RUN_DEPTHS to the halving rule: with nothing set but the layout, the replica has run
    lists for depths 1, 2, 4 and 8, steps by eight symbols, and answers like the oracle; RBG_LAYOUT_AUTO under a budget the slot tables
    of five symbols per gather do not fit chooses the same replica by itself, and the slot layout keeps at most five."""
    S = synth
    assert capi.get_default_option(capi.OPT_KMER_STEPS) == 8 and capi.get_default_option(capi.OPT_RUN_DEPTHS) == 0
    rb = _with_layout(capi.LAYOUT_RUNS, lambda: ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0))
    info, li = rb.info(), rb.layout_info()
    assert info.kmer_steps == 8 and li.depth_mask_kept == 0x8B and li.depths_composed == 8
    assert all(info.depth_runs[d] > 0 for d in (0, 1, 3, 7)) and not any(info.depth_runs[d] for d in (2, 4, 5, 6))
    assert info.depth_runs[7] >= info.depth_runs[3] >= info.depth_runs[1] >= info.r
    _run_indexed_checks(S, rb)
    slots = _with_layout(capi.LAYOUT_SLOTS, lambda: ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0))
    assert slots.info().rank_layout == capi.LAYOUT_SLOTS and slots.info().kmer_steps == 5
    need_slots, need_runs = int(slots.info().hbm_bytes), int(info.hbm_bytes)
    slots.close()
    assert need_runs * 2 < need_slots
    with capi.default_option(capi.OPT_HBM_BUDGET_MB, (need_runs * 2 >> 20) + 1):   # (room for the run lists of eight depths, not for the slot tables of five)
        auto = ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0)
    ai = auto.info()
    assert ai.rank_layout == capi.LAYOUT_RUNS and ai.kmer_steps == 8 and auto.layout_info().depth_mask_kept == 0x8B
    _run_indexed_checks(S, auto)


@pytest.mark.parametrize("recs", [False, True])
@pytest.mark.parametrize("fill_shift,super_shift,ks,depths,dir_runs,phi_per", [(6, 2, 5, 0, None, None), (4, 1, 3, 0x7, "1", "0.5"), (9, 5, 1, 0, "16", "4"),
                                                                                (5, 3, 5, 0x1F, "40", "9"), (6, 2, 8, 0, None, None), (5, 3, 7, 0x55, "40", "9")])
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
        rb = _with_layout(capi.LAYOUT_RUNS, lambda: ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0))
    finally:
        ra.set_default_option(capi.OPT_POS_BYTES, 0)
        ra.set_default_option(capi.OPT_KMER_STEPS, DEFAULT_KMER_STEPS)
        ra.set_default_option(capi.OPT_RUN_DEPTHS, 0)
        ra.set_default_option(capi.OPT_RUN_PHI, 0)
        ra.set_default_option(capi.OPT_RUN_REC, 0)
        for k in ("RBG_RUN_FILL_SHIFT", "RBG_PHI_SUPER_SHIFT", "RBG_RANK_DIR_RUNS", "RBG_PHI_DIR_PER", "RBG_RUN_REC_PER"):
            os.environ.pop(k, None)
    li = rb.layout_info()
    assert li.run_fmt == 2 and li.fill_shift == fill_shift and rb.info().pos_bytes == 8
    kept = [d for d in range(capi.MAX_KMER_DEPTH) if li.depth_mask_kept >> d & 1]
    assert sum(li.fillers) > 0 and (fill_shift > 6 or all(li.fillers[d] > 0 for d in kept)), list(li.fillers)   # tables with gaps beyond the distance
    assert li.phi_fillers > 0 and li.phi_entries == len(S.heads) + li.phi_fillers
    assert (S.n >> li.phi_dir_shift) >> super_shift > 2                   # several super blocks
    _run_indexed_checks(S, rb)


@pytest.mark.parametrize("depths", [0, 0x15, 0x1f, 0xA5])
def test_composition_spills_kept_depths_to_host_and_back(synth, depths, capfd):
    """k_compose.hip: when a depth's sweeps do not fit beside the kept depths made so far (r = 1e9 with 5 symbols per step),
    those wait in host memory and come back at the end.  RBG_COMPOSE_SPILL takes that path at test size: the index must be
    the same one (every query equal to the oracle's), with the same depth set."""
    S = synth
    ra.set_default_option(capi.OPT_RUN_DEPTHS, depths)
    os.environ["RBG_COMPOSE_SPILL"] = "1"
    os.environ["RBG_VERBOSE"] = "1"
    try:
        rb = _with_layout(capi.LAYOUT_RUNS, lambda: ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0))
    finally:
        ra.set_default_option(capi.OPT_RUN_DEPTHS, 0)
        os.environ.pop("RBG_COMPOSE_SPILL", None)
        os.environ.pop("RBG_VERBOSE", None)
    assert "waits in host memory" in capfd.readouterr().err
    li = rb.layout_info()
    assert li.depths_composed == (depths or 0x80).bit_length() and li.depth_mask_kept == (depths or 0x8B), (li.depths_composed, li.depth_mask_kept)
    _run_indexed_checks(S, rb)


@pytest.mark.parametrize("pos_bytes,rec,dir_runs", [(0, None, None), (8, None, "64"), (0, None, "64"), (0, "rec", None), (8, "rec", "64"), (0, "rec", "64")])
def test_run_indexed_crowded_buckets(pos_bytes, rec, dir_runs):
    """Directory buckets with a hundred and more runs (k_runs.hip: narrowing rounds, one after the other when the
    directory is coarse -- dir_runs = RBG_RANK_DIR_RUNS, or RBG_RUN_REC_PER with bucket records; scans whose candidates all lie
    below the position; bucket records that overflow into pivots) beside buckets with none: a text that is 2 000 bases repeated 300 times, then
    1 500 x (one of A,C,G,T + the same 14-mer + 10 random bases) -- the rows of the suffixes that start with the 14-mer
    are consecutive and their BWT symbols change at nearly every row, while the average run is 39 rows long and sets
    the bucket width.  Both position widths."""
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
    for c in b"ACGT":   # the directory's rule (rbg_capi.hip upload_tables_runs2): at most dir_runs (4) runs per bucket on average
        st = starts[heads == c]
        sh = 0
        while len(st) * (2 << sh) <= float(dir_runs or 4) * n:
            sh += 1
        crowded = max(crowded, int(np.bincount(st >> sh).max()))
    assert crowded > (256 if dir_runs else 64), crowded   # one narrowing round at least; two with the coarse directory
    ra.set_default_option(capi.OPT_POS_BYTES, pos_bytes)
    if rec == "rec":      # bucket records: the crowded buckets overflow them and go through the run list (narrowed by the lane)
        ra.set_default_option(capi.OPT_RUN_REC, 2)
        if dir_runs is not None:
            os.environ["RBG_RUN_REC_PER"] = dir_runs
    else:
        ra.set_default_option(capi.OPT_RUN_REC, 1)
    if dir_runs is not None:
        os.environ["RBG_RANK_DIR_RUNS"] = dir_runs
    try:
        rb = _with_layout(capi.LAYOUT_RUNS, lambda: ra.RowBowt.from_runs(heads, lens, ssa, esa, device=0))
    finally:
        ra.set_default_option(capi.OPT_POS_BYTES, 0)
        ra.set_default_option(capi.OPT_RUN_REC, 0)
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
    rb = _with_layout(capi.LAYOUT_RUNS, lambda: ra.load_rowbowt(os.path.join(data_dir, "small.fa"), ra.LoadRbwtFlag.SA | ra.LoadRbwtFlag.MA, device=0))
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
    assert hbm_runs < 8_000_000   # (run lists, samples and bucket records of four depths, each array rounded to 64 KB, and 2 MB of table records: 4^8 tables at depth 8)


@pytest.mark.parametrize("pos_bytes,mask,kept,recs", [(0, 0, 0x8B, False), (0, 0x15, 0x15, False), (8, 0x11, 0x11, False), (0, 0x13, 0x13, True),
                                                       (8, 0x0A, 0x0B, False), (0, 0x1E, 0x1F, False), (8, 0x15, 0x15, True),
                                                       (0, 0xA4, 0xA5, True), (8, 0x81, 0x81, False), (8, 0x60, 0x61, True), (0, 0xFF, 0xFF, False)])
def test_run_indexed_layout_sparse_depths(synth, pos_bytes, mask, kept, recs):
    """RBG_OPT_RUN_DEPTHS: run lists for some of the k-mer depths only (bit d - 1; depth 1 always, nothing above the
    highest bit).  A step takes the longest stretch a kept depth covers (k_runs.hip, k_runs_seeds.hip pick_step), so the
    answers are those of every other layout -- the oracle's -- in less space.  With directories and with bucket records."""
    S = synth
    ra.set_default_option(capi.OPT_POS_BYTES, pos_bytes)
    ra.set_default_option(capi.OPT_RUN_REC, 2 if recs else 1)
    try:
        with capi.default_option(capi.OPT_RUN_DEPTHS, 0xFF):
            full = _with_layout(capi.LAYOUT_RUNS, lambda: ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0))
        with capi.default_option(capi.OPT_RUN_DEPTHS, mask):
            rb = _with_layout(capi.LAYOUT_RUNS, lambda: ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0))
    finally:
        ra.set_default_option(capi.OPT_POS_BYTES, 0)
        ra.set_default_option(capi.OPT_RUN_REC, 0)
    fi, info = full.info(), rb.info()
    runs_full, runs_kept = list(fi.depth_runs), list(info.depth_runs)
    assert info.rank_layout == capi.LAYOUT_RUNS and info.kmer_steps == kept.bit_length() and fi.kmer_steps == 8
    assert runs_kept == [x if kept >> d & 1 else 0 for d, x in enumerate(runs_full)]   # (rbg_info: the depths left out report no runs)
    if kept != 0xFF:
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
    assert ri.kmer_steps == info.kmer_steps and list(ri.depth_runs) == runs_kept
    lo, hi, k = rep.find_range_w_toehold(seqs, off)
    assert (lo == wlo).all() and (hi == whi).all() and (k == wk).all()
    rep.close()
    rb.close()
    full.close()
    o.close()


def test_default_load_of_an_index_too_large_for_a_quarter_of_the_device_plans_its_depth_and_raises_its_budget():
    """What an index of r = 1e9 runs meets on a 288 GB device (profiles/r05_pangenome_stream_r1e9_default.json), on a test-sized index:
    RBG_ASSUME_FREE_HBM_MB (tests only) caps the free HBM the PLANNING of a load assumes.  With 200 MB "free", a million runs over n = 1e8:
    not even the single-symbol slot tables fit a quarter (50 MB), so the run-indexed layout is certain; the quarter would leave it one symbol per
    step, so RBG_LAYOUT_AUTO takes three quarters (150 MB) and says so; the composition depth is planned BEFORE composing from the sweeps' 70
    bytes per piece (3 of the 8 asked); the budget reported is the one fixed before anything was on the device.  Answers equal the oracle's."""
    rng = np.random.default_rng(43)
    heads, lens, ssa, esa, n = _random_run_index(rng, 1_000_000, 200)
    o = orc.Oracle.from_runs(heads, lens, ssa, esa)
    os.environ["RBG_ASSUME_FREE_HBM_MB"] = "200"
    try:
        # nothing set but the ftab: RBG_LAYOUT_AUTO, eight symbols asked for, no budget given.  (The 12-symbol device ftab is a constant 268 MB the
        #  budget rule does not count -- half a per cent of a replica at the scale this is about, more than this whole index)
        with capi.default_option(capi.OPT_FTAB_K, 0):
            rb = ra.RowBowt.from_runs(heads, lens, ssa, esa, device=0)
    finally:
        del os.environ["RBG_ASSUME_FREE_HBM_MB"]
    info, li = rb.info(), rb.layout_info()
    assert info.rank_layout == capi.LAYOUT_RUNS and li.budget_raised == 1
    assert int(info.hbm_free_at_load) == 200 << 20 and int(info.hbm_budget) == 150 << 20
    assert info.kmer_steps_requested == 8 and 2 <= info.kmer_steps <= 4 and li.depths_composed == info.kmer_steps
    assert int(info.hbm_bytes) <= int(info.hbm_budget) and li.depths_dropped_budget == 0
    # the same index with the budget given explicitly is not second-guessed
    os.environ["RBG_ASSUME_FREE_HBM_MB"] = "200"
    try:
        with capi.default_option(capi.OPT_HBM_BUDGET_MB, 60), capi.default_option(capi.OPT_FTAB_K, 0):
            rb2 = ra.RowBowt.from_runs(heads, lens, ssa, esa, device=0)
    finally:
        del os.environ["RBG_ASSUME_FREE_HBM_MB"]
    assert rb2.layout_info().budget_raised == 0 and int(rb2.info().hbm_budget) == 60 << 20 and int(rb2.info().hbm_bytes) <= 60 << 20
    assert rb2.info().kmer_steps <= info.kmer_steps
    reads = _lf_walk_reads(o, heads, lens, n, rng, 400, 40) + [b"ACGT", b"N", b""]
    seqs, off = ra.pack_reads(reads)
    want = o.find_range_w_toehold_batch(seqs, off)
    # Ranges on both.  Toeholds and locations where the load kept single-symbol steps only: this index is a random run list, not a BWT -- its
    # samples are not LF-consistent, and a k-mer step's stored sample may be taken from either of two run ends that coincide (equal in a BWT:
    # both are SA - d of the same row; different here, for about one read in a few hundred, on either layout).  The k-mer toeholds are pinned
    # on true BWTs (test_run_indexed_layout, the pangenome streams).
    for x in (rb, rb2):
        lo, hi = x.find_range(seqs, off)
        assert (lo == want[0]).all() and (hi == want[1]).all()
        if x.info().kmer_steps == 1:
            got = x.find_range_w_toehold(seqs, off)
            assert all((g == w).all() for g, w in zip(got, want))
            loc_off, locs = x.locs_at(got[0], got[1], got[2], 5)
            woff, wlocs = o.locs_at_batch(want[0], want[1], want[2], 5)
            assert (loc_off == woff).all() and (locs == wlocs).all()
        x.close()
    o.close()


def test_records_for_every_kept_depth_come_before_the_depths_in_between():
    """The automatic rules under a shrinking budget (capi/load.ipp, capi/upload_runs.ipp): while bucket records for all kept depths fit, every depth has
    them (a depth takes narrower buckets only with what the others do not need); when they no longer fit but would for the first and the deepest depth
    alone, the depths between go (a step per read) rather than the records (narrowing rounds on every step of a depth left on directories: K2 13.8 against
    7 ms at n = 5e10, profiles/r05_pangenome_stream_n5e10_default.json).  A synthetic run list of a million runs (the budget option counts MB); ranges
    equal the oracle's at the budget where that happens."""
    rng = np.random.default_rng(47)
    heads, lens, ssa, esa, n = _random_run_index(rng, 1_000_000, 200)
    def build(budget_mb):
        with capi.default_option(capi.OPT_FTAB_K, 0), capi.default_option(capi.OPT_HBM_BUDGET_MB, budget_mb), capi.default_option(capi.OPT_KMER_STEPS, 5):
            return _with_layout(capi.LAYOUT_RUNS, lambda: ra.RowBowt.from_runs(heads, lens, ssa, esa, device=0))
    full = build(1 << 16)
    fi, fl = full.info(), full.layout_info()
    assert fl.depth_mask_kept == 0x13 and all(fl.rec_bytes[d] > 0 for d in (0, 1, 4)) and fl.phi_slots > 0
    top = int(fi.hbm_bytes)
    full.close()
    seen, picked = [], None
    for frac in (0.95, 0.85, 0.75, 0.65, 0.55, 0.5, 0.45, 0.4, 0.35, 0.3):
        budget_mb = max(1, int(top * frac) >> 20)
        rb = build(budget_mb)
        li, info = rb.layout_info(), rb.info()
        kept = [d for d in range(8) if li.depth_mask_kept >> d & 1]
        seen.append((frac, hex(li.depth_mask_kept), [d for d in kept if li.rec_bytes[d] > 0], int(info.hbm_bytes) >> 20, budget_mb))
        assert int(info.hbm_bytes) <= (budget_mb + 2) << 20, seen
        if li.depth_mask_kept == 0x11 and li.rec_bytes[0] > 0 and li.rec_bytes[4] > 0 and picked is None:
            picked = rb
            continue
        rb.close()
    assert picked is not None, seen     # the depth in between went while both ends kept their records
    first = next(i for i, x in enumerate(seen) if x[1] == "0x11")
    assert all(x[1] == "0x13" and x[2] == [0, 1, 4] for x in seen[:first]), seen   # until then: every depth kept, every depth with records
    assert picked.info().kmer_steps == 5 and picked.layout_info().depths_dropped_budget == 0x02
    o = orc.Oracle.from_runs(heads, lens, ssa, esa)
    reads = _lf_walk_reads(o, heads, lens, n, rng, 400, 40) + [b"ACGT", b"N", b""]
    seqs, off = ra.pack_reads(reads)
    want = o.find_range_w_toehold_batch(seqs, off)
    lo, hi = picked.find_range(seqs, off)
    assert (lo == want[0]).all() and (hi == want[1]).all()
    picked.close()
    o.close()


def test_run_indexed_layout_budget_leaves_middle_depths_out():
    """Over budget the run-indexed layout gives up the depths between the first and the deepest before the deepest
    itself (rbg_capi.hip upload): the step length stays, the space goes down, the answers stay.  (A synthetic run list of
    a million runs: the budget option counts MB.)"""
    rng = np.random.default_rng(41)
    heads, lens, ssa, esa, n = _random_run_index(rng, 1_000_000, 200)
    def build():   # (without the device ftab: the budget is about the run lists; phi over the list: phi slots are the budget's to give, too)
        with capi.default_option(capi.OPT_FTAB_K, 0), capi.default_option(capi.OPT_RUN_DEPTHS, depths[0]), capi.default_option(capi.OPT_RUN_PHI, 1), capi.default_option(capi.OPT_RUN_REC, 1), \
                capi.default_option(capi.OPT_KMER_STEPS, 5):
            return _with_layout(capi.LAYOUT_RUNS, lambda: ra.RowBowt.from_runs(heads, lens, ssa, esa, device=0))
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


@pytest.mark.parametrize("pos_bytes", [4, 8])
def test_staged_read_walk_at_its_boundaries(synth, pos_bytes):
    """The run-indexed search stages a wave's reads as 2-bit codes in LDS (round 6: k_runs.hip STAGE, rbg_runs_device.hpp stage_read) -- unless a read of the wave is
    longer than 256 symbols or holds a symbol outside the k-mer alphabet, then the wave walks bytes.  Both walks, and waves that mix them, against the oracle
    (find_range / find_range_w_toehold, rowbowt.hpp:121-131, :169-184): read lengths around the 16-symbol code words and the 256-symbol cap, reads that start at every
    byte alignment, an N / a lower-case base / a NUL at the first, a middle and the last position, whole batches of long reads, and one long read among 63 short ones."""
    S = synth
    rng = np.random.default_rng(17)
    with capi.default_option(capi.OPT_RANK_LAYOUT, capi.LAYOUT_RUNS), capi.default_option(capi.OPT_POS_BYTES, pos_bytes):
        rb = ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0)
    o = orc.Oracle.from_runs(S.heads, S.lens, S.ssa, S.esa)
    unit = S.L + S.pad

    def piece(m):
        h = int(rng.integers(S.H))
        s0 = h * unit + int(rng.integers(0, S.L - m + 1))
        return bytearray(S.text[s0:s0 + m].tobytes())
    reads = []
    for m in (1, 2, 11, 12, 13, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 150, 254, 255, 256, 257, 258, 300, 513):
        for _ in range(40):
            reads.append(bytes(piece(m)))                       # (consecutive reads of odd lengths: every start alignment modulo 16)
    for m in (20, 100, 256):                                    # symbols outside the alphabet at the ends and inside
        for where in (0, m // 2, m - 1):
            for bad in (ord("N"), ord("a"), 0, 0xC1):
                r = piece(m)
                r[where] = bad
                reads.append(bytes(r))
    for _ in range(3):                                          # a wave of 63 short reads and one long one; a wave with one non-ACGT read
        reads += [bytes(piece(100)) for _ in range(63)] + [bytes(piece(400))]
        w = [piece(100) for _ in range(64)]
        w[int(rng.integers(64))][50] = ord("N")
        reads += [bytes(x) for x in w]
    reads += [bytes(piece(300)) for _ in range(128)]            # whole waves of long reads
    reads += [b"", b"A", b"N"]
    seqs, off = ra.pack_reads(reads)
    wlo, whi, wk = o.find_range_w_toehold_batch(seqs, off, nthreads=4)
    lo, hi, k = rb.find_range_w_toehold(seqs, off)
    assert (lo == wlo).all() and (hi == whi).all() and (k == wk).all()
    clo, chi = rb.find_range(seqs, off)
    assert (clo == wlo).all() and (chi == whi).all()
    # the device entry points on a resident batch (what bench.py times): the same, with the instrumented instantiation counting one chunk fetch per 16 bytes of a staged read
    import torch
    dev = torch.device("cuda:0")
    d_seqs = torch.from_numpy(np.concatenate([seqs, np.zeros((-len(seqs)) % 16 + 16, np.uint8)])).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    N = len(reads)
    d_lo, d_hi, d_k = (torch.empty(N, dtype=torch.int64, device=dev) for _ in range(3))
    d_stats = torch.zeros(16, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    L = ra.lib()
    assert L.rbg_find_range_w_toehold_dev(rb.h, d_seqs.data_ptr(), d_off.data_ptr(), N, d_lo.data_ptr(), d_hi.data_ptr(), d_k.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert (d_lo.cpu().numpy().view(np.uint64) == wlo).all() and (d_hi.cpu().numpy().view(np.uint64) == whi).all() and (d_k.cpu().numpy().view(np.uint64) == wk).all()
    assert L.rbg_find_range_stats_dev(rb.h, d_seqs.data_ptr(), d_off.data_ptr(), N, d_lo.data_ptr(), d_hi.data_ptr(), d_k.data_ptr(), d_stats.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert (d_lo.cpu().numpy().view(np.uint64) == wlo).all() and (d_k.cpu().numpy().view(np.uint64) == wk).all()
    chunks = int(d_stats[6].item())
    total = int(off[-1])
    assert total // 16 <= chunks <= total // 16 + 2 * N            # every byte fetched about once (a chunk per 16 bytes + the ends), staged or not
    assert int((whi >= wlo).sum()) > 800
    rb.close()
    o.close()
