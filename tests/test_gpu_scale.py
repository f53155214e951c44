"""GPU parity (through the C-ABI, bit-exact against the oracle; needs an MI355X): scale: positions beyond 32 bits, width limits, a mid-scale pangenome, BASELINE-size properties, the streamed pangenome, the two-rank rehearsal."""
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
        ra.set_default_option(capi.OPT_KMER_STEPS, DEFAULT_KMER_STEPS)
    assert rb1.info().n == n and rb1.info().pos_bytes == 8 and rb1.info().kmer_steps == 1
    lo, hi, k = rb1.find_range_w_toehold(seqs, off)
    assert (lo == wlo).all() and (hi == whi).all() and (k == wk).all()
    loc_off, locs = rb1.locs_at(lo, hi, k, max_hits=64)
    assert (loc_off == woff).all() and (locs == wlocs).all()
    rb1.close()
    # k-mer levels (the deepest one dropped by the budget rule): ranges, and the phi walks from the oracle's
    # toeholds.  The toeholds of k-mer steps are not compared here: composing the run-end samples of a k-mer
    # table presumes samples that are the suffix array's, which random ones are not.
    rb = _with_layout(capi.LAYOUT_PREFER_SLOTS, lambda: ra.RowBowt.from_runs(heads, lens, ssa, esa, device=0))
    i = rb.info()
    assert i.n == n and i.pos_bytes == 8 and 2 <= i.kmer_steps <= 5 and i.hbm_bytes < 235e9   # (the budget rule widens the deep levels' buckets, then drops levels)
    assert i.rank_layout == capi.LAYOUT_SLOTS
    lo, hi, _ = rb.find_range_w_toehold(seqs, off)
    lo2, hi2 = rb.find_range(seqs, off)
    assert (lo == wlo).all() and (hi == whi).all() and (lo2 == wlo).all() and (hi2 == whi).all()
    loc_off, locs = rb.locs_at(wlo, whi, wk, max_hits=64)
    assert (loc_off == woff).all() and (locs == wlocs).all()
    # the run-indexed layout at this size: 3 GB instead of 140, sampled index three levels deep with a 16-key top
    # (single steps: toeholds compared; k-mer depths: ranges and the walks from the oracle's toeholds, as above)
    for top_kb, ks in ((48, 1), (0, 1), (48, 5), (0, 3), (48, 8)):   # (top_kb: a label only since round 5 -- it pairs the depth with a phi structure)
        ra.set_default_option(capi.OPT_KMER_STEPS, ks)
        ra.set_default_option(capi.OPT_RUN_PHI, 1 if top_kb else 2)   # phi over the list of sampled positions / through phi slots of about n / r rows
        try:
            rbr = _with_layout(capi.LAYOUT_RUNS, lambda: ra.RowBowt.from_runs(heads, lens, ssa, esa, device=0))
        finally:
            ra.set_default_option(capi.OPT_KMER_STEPS, DEFAULT_KMER_STEPS)
            ra.set_default_option(capi.OPT_RUN_PHI, 0)
        ir = rbr.info()
        assert ir.rank_layout == capi.LAYOUT_RUNS and ir.pos_bytes == 8 and ir.kmer_steps == ks and ir.hbm_bytes < (4e9 if ks == 1 else 12e9 if ks <= 5 else 20e9) + (0 if top_kb else 3e9)
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
            rbr = _with_layout(capi.LAYOUT_RUNS, lambda: ra.RowBowt.from_runs(*c[:4], device=0))
        ir = rbr.info()
        assert ir.rank_layout == capi.LAYOUT_RUNS and ir.pos_bytes == 8 and ir.n == n and ir.hbm_bytes < 24e9
        check(rbr, c, toeholds=ks == 1)
        rbr.close()
    # slot tables, single-symbol level, 256-row rank buckets and 256-position phi buckets: n/256 x (5 x 20 + 36) bytes
    with capi.default_option(capi.OPT_KMER_STEPS, 1), capi.default_option(capi.OPT_PHI_BUCKET_SHIFT, 8):
        rbs = _with_layout(capi.LAYOUT_SLOTS, lambda: ra.RowBowt.from_runs(*c[:4], device=0))
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
        rbw = _with_layout(capi.LAYOUT_SLOTS, lambda: ra.RowBowt.from_runs(*c[:4], device=0))
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
            _with_layout(capi.LAYOUT_SLOTS, lambda: ra.RowBowt.from_runs(*c[:4], device=0))
        assert ei.value.code == -4                          # RBG_EARG: 40-bit ranks cannot hold this index
    rba = ra.RowBowt.from_runs(*c[:4], device=0)            # AUTO: the single-symbol slot level (n/256 x 100 B = 430 GB) does not fit
    ia = rba.info()
    assert ia.rank_layout == capi.LAYOUT_RUNS and ia.pos_bytes == 8 and ia.hbm_bytes < 24e9
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
            _with_layout(layout, lambda: ra.RowBowt.from_runs(heads, lens, ssa, esa, device=0))
        assert ei.value.code == -4


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
    # the post-mortem block of a multi-GPU run: who ran where, how long each rank loaded / waited / stepped, what the per-node cache cost
    per = d["per_rank"]
    assert [x["rank"] for x in per] == [0, 1] and all(x["device"] == 0 and x["load_s"] > 0 and x["ms_per_step"] > 0 and x["k2_ms"] > 0 and x["hbm_bytes"] > 0 for x in per)
    assert per[1]["wait_for_rank0_s"] >= 0 and d["rccl_ranks_seen"] == 2 and "gloo" in d["collective_backend"]
    assert d["cache_write_s"] > 0 and d["cache_bytes"] > 100_000 and d["cache_path"].endswith(".rbgpu") and d["suffix_array_s"] > 0
    assert abs(d["ms_per_step"] - max(x["ms_per_step"] for x in per)) < 1e-6      # the line's time is the max over the ranks


def test_full_size_properties_and_parity_sample():
    """BASELINE.json's size (n = 2.0e9, r = 3.7e7, 10 M x 100 bp reads) through the size-independent
    properties and an oracle sample: one bench.py step in a subprocess (about a minute and a half: the
    index synthesis dominates)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                        "--check-reads", "5000", "--property-reads", "300000", "--markers", "--no-pangenome-shape"],   # (that block is the test below)
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
        assert li["run_fmt"] == 2 and li["depths_dropped_budget"] == 0
        assert (li["rank_directories"] == 1) != (sum(li["rec_bytes"]) > 0)   # ranks: directories over the run lists, or bucket records
        assert (li["phi_directory"] == 1) != (li["phi_slots"] > 0)      # phi: the list of sampled positions with its directory, or slots of about n / r rows
    print(f"n = {ix['n']:.3e}, {layout}: {d['value']:.3e} reads/s streamed, {ix['hbm_bytes'] / 1e9:.1f} GB replica")


def test_pangenome_shape_r_above_1e8_default_load():
    """BASELINE.json configs[3]'s INDEX at its named scale (r >= 1e8) under the driver's own test run: `tools/pangenome_stream.py --preset driver`
    = a true BWT of n = 2.0e10 symbols (L = 1e8, H = 200; rle_string.hpp:131-161 / toehold_sa.hpp:56-72 are the structures being scaled),
    a DEFAULT rbg_load -- no option, no budget, no RBG_* variable --, five batches of 10 M x 150 bp reads generated on the device.  Ranges,
    toeholds (above 2^32 among them), locations and the count-only kernel bit-exact against the oracle on 2 000 reads; the size-independent
    properties on 100 000; and what the load-time budget rules decided (rbg_layout_info), so that a rule edit shows here and not only in a
    builder-side full-size run."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not k.startswith("RBG_")}
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "pangenome_stream.py"), "--preset", "driver"], capture_output=True, timeout=1200, cwd=root, env=env)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    d = json.loads(p.stdout.decode().strip().splitlines()[-1])
    ix = d["config"]["index"]
    assert ix["default_load"] is True and ix["true_bwt"] and ix["r"] >= 100_000_000 and ix["n"] > 19_000_000_000 and ix["pos_bytes"] == 8
    # the rules: beyond toy sizes a default load builds the run-indexed layout within a quarter of the free HBM (no raise at this r),
    # eight symbols per step, depth 1 and the deepest kept with bucket records on every kept depth, phi slots
    li = ix["layout_info"]
    assert ix["rank_layout"] == 2 and li["run_fmt"] == 2 and li["budget_raised"] == 0
    assert ix["hbm_bytes"] <= ix["hbm_budget"] <= ix["hbm_free_at_load"] // 4 + (1 << 20)
    assert ix["symbols_per_gather"] == 8 and 1 in li["depths_kept"] and 8 in li["depths_kept"]
    assert li["depths_with_records"] == li["depths_kept"] and li["rank_directories"] == 0 and li["phi_slots"] > 0 and li["phi_directory"] == 0
    par = d["parity"]
    assert par["reads_checked"] == 2000 and par["bit_exact_vs_oracle"] and par["count_only_kernel_bit_exact"]
    assert par["toeholds_above_2^32"] > 100 and par["locs_checked"] > 100_000
    props = d["properties"]
    assert props["reads"] == 100000 and props["locations"] > 5_000_000
    assert all(props[k] for k in ("unmutated_reads_all_found", "empty_is_{1,0}", "every_location_is_an_occurrence", "locations_distinct",
                                  "occ_equals_range_width", "own_position_reported"))
    c = d["counters"]
    assert c["reads"] == 50_000_000 and c["sum_occ"] == c["sum_locs"] > 5_000_000_000
    roof = d["roofline"]
    assert roof["bound"] == "hbm" and 0 < roof["frac"] < 1 and set(roof["kernels"]) == {"find_range_w_toehold", "locate_fill"}
    assert d["search_touched_per_read"]["searched_ranks"] < 0.5      # records: (almost) no narrowing rounds
    print(f"n = {ix['n']:.3e}, r = {ix['r']:.3e}: {d['value']:.3e} reads/s streamed from a {ix['hbm_bytes'] / 1e9:.1f} GB default replica; "
          f"K2 {d['kernel_ms_one_batch']['find_range_w_toehold']:.1f} ms, K3 {d['kernel_ms_one_batch']['locate_fill']:.1f} ms per 10 M reads")


def test_bench_replicas_flag_three_replicas_on_device_0():
    """`bench.py --replicas 3 --replica-devices 0,0,0`: the count+locate step on three replicas of ONE process (one build,
    rbg_replicate_many, a host thread + stream + batch per replica).  Every copy reproduces the primary's ranges, toeholds and
    locations on the primary's batch; the reduced counters are the stream's; the line's n_gpus is the replicas formed."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--replicas", "3", "--replica-devices", "0,0,0", "--L", "1500000", "--H", "8", "--reads", "150000",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-space-speed", "--no-markers", "--check-reads", "2000", "--property-reads", "20000"],
                       capture_output=True, timeout=600, cwd=root)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    d = json.loads(p.stdout.decode().strip().splitlines()[-1])
    r = d["replicas_one_process"]
    assert d["n_gpus"] == 3 and r["formed"] == 3 and r["every_copy_identical_to_the_primary_on_its_batch"] and r["counters"]["as_streamed"]
    assert r["counters"]["reads"] == 3 * 150000 * 3 and d["value"] == r["value"] and d["single_replica"]["value"] > 0
    assert "ONE process" in d["config"]["parallelism"] and d["parity"]["bit_exact_vs_oracle"]


def test_pangenome_stream_one_process_three_replicas_on_device_0():
    """BASELINE.json configs[3]'s several-GPU shape as ONE process (tools/pangenome_stream.py --replicas 3 --replica-devices 0,0,0): the
    index is built once, rbg_replicate_many makes two peer copies, three host threads stream the three rbg_shard_bounds blocks of the
    read indices on three HIP streams with their own generators, the counters are reduced at the end.  Checked: the primary against
    the oracle and the properties, every copy against the primary on the same reads, the reduced counters against the stream
    (reads == total, sum of range widths == locations written), and the shard arithmetic (blocks cover the stream, sizes differ by
    at most one).  The reference's dispatcher is one process with one index too (rb_align.cpp:176-178)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    total = 3_000_001
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "pangenome_stream.py"), "--L", "2000000", "--H", "40", "--total-reads", str(total),
                        "--reads", "400000", "--check-reads", "4000", "--property-reads", "50000", "--layout", "runs", "--implicit-text", "on",
                        "--replicas", "3", "--replica-devices", "0,0,0", "--hbm-reserve-gb", "0"],
                       capture_output=True, timeout=900, cwd=root)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    d = json.loads(p.stdout.decode().strip().splitlines()[-1])
    rep = d["replicas"]
    assert d["n_gpus"] == 3 and rep["formed"] == 3 and rep["devices"] == [0, 0, 0] and rep["identical_outputs_on_every_replica"] is True
    per = rep["per_replica"]
    assert sorted(x["reads"] for x in per) == [1_000_000, 1_000_000, 1_000_001] and all(x["batches"] == 3 and x["ms"] > 0 for x in per)
    assert d["seconds"] * 1e3 >= max(x["ms"] for x in per) - 1e-6       # max over the replicas (first gate of any to last end of any)
    c = d["counters"]
    assert c["reads"] == total and c["sum_occ"] == c["sum_locs"] > 10 * total and "3 replicas" in c["reduced_over"]
    assert d["parity"]["reads_checked"] == 4000 and d["parity"]["bit_exact_vs_oracle"] and d["parity"]["count_only_kernel_bit_exact"]
    props = d["properties"]
    assert all(props[k] for k in ("unmutated_reads_all_found", "empty_is_{1,0}", "every_location_is_an_occurrence", "locations_distinct",
                                  "occ_equals_range_width", "own_position_reported"))
    assert d["config"]["index"]["rank_layout"] == 2 and d["peaks"]["host_bytes"] > 0


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
