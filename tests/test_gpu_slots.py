"""GPU parity (through the C-ABI, bit-exact against the oracle; needs an MI355X): the slot layout: every position width, bucket shift and k-mer depth; composition; budget rule; overflow paths."""
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
        ra.set_default_option(capi.OPT_KMER_STEPS, DEFAULT_KMER_STEPS)
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
                    rb = _with_layout(layout, lambda: ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0))
            finally:
                os.environ.pop("RBG_HOST_COMPOSE", None)
            i = rb.info()
            infos.append((i.kmer_steps, list(i.depth_runs), i.rank_slots_overflow))
            assert i.kmer_steps == (5 if layout == capi.LAYOUT_SLOTS else 8) and i.pos_bytes == pos_bytes   # (the slot layout stages at most five symbols per gather)
            lo, hi, k = rb.find_range_w_toehold(seqs, off)
            assert (lo == wlo).all() and (hi == whi).all() and (k == wk).all()
            loc_off, locs = rb.locs_at(lo, hi, k)
            assert (loc_off == woff).all() and (locs == wlocs).all()
            for ks in (2, 3, 4) + ((6,) if layout == capi.LAYOUT_RUNS else ()):   # fewer levels asked for: only those are composed
                with capi.default_option(capi.OPT_KMER_STEPS, ks), capi.default_option(capi.OPT_POS_BYTES, pos_bytes):
                    if host:
                        os.environ["RBG_HOST_COMPOSE"] = host
                    try:
                        rb2 = _with_layout(layout, lambda: ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0))
                    finally:
                        os.environ.pop("RBG_HOST_COMPOSE", None)
                assert rb2.info().kmer_steps == ks
                lo2, hi2, k2 = rb2.find_range_w_toehold(seqs, off)
                assert (lo2 == wlo).all() and (hi2 == whi).all() and (k2 == wk).all()
                rb2.close()
            rb.close()
        assert infos[0] == infos[1], infos
    o.close()


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
    run-indexed layout with at least as many symbols per step (up to eight) while that fits (about 110 bytes per run); answers do not change."""
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
    with capi.default_option(capi.OPT_RANK_LAYOUT, capi.LAYOUT_RUNS):
        r8 = ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0)
    need_runs8 = int(r8.info().hbm_bytes)   # (the run-indexed replica with eight symbols per step: a fraction of the slot tables of five)
    assert r8.info().kmer_steps == 8 and need_runs8 * 4 < full
    r8.close()
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
        assert int(rb.info().kmer_steps_requested) == (8 if runs else 5) and int(rb.info().hbm_budget) == max(1, int(full * frac) >> 20) << 20   # (asked for: eight; the slot layout takes five of them at most)
        if runs and layout == capi.LAYOUT_AUTO and int(rb.info().hbm_budget) >= 2 * need_runs8:
            assert int(rb.info().kmer_steps) == 8        # the switch was made to keep the symbols per step: all eight while the budget holds them
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
