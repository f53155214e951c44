"""GPU parity (through the C-ABI, bit-exact against the oracle; needs an MI355X): the *_dev entry points, graph capture, instrumented kernels, packed reads, the host-pointer pipeline, concurrency, replicas."""
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


@pytest.mark.parametrize("layout,phi", [(capi.LAYOUT_SLOTS, 0), (capi.LAYOUT_RUNS, 2), (capi.LAYOUT_RUNS, 1)])
@pytest.mark.parametrize("pos_bytes", [4, 8])
def test_locate_chains_in_locus_order(synth, layout, phi, pos_bytes, monkeypatch):
    """K3's chain order by LOCUS (round 6): with a document table attached (rbg_set_docs: one document per haplotype, doclist.hpp:46-79) the chains
    are sorted by {offset inside the document, document} instead of absolute text position.  Result-neutral: the locations -- ToeholdSA::locate_range,
    toehold_sa.hpp:37-49 -- are the oracle's whatever the order, for every max_hits, on the slot layout and on the run-indexed one with phi slots and
    with the phi list, at both position widths; a toehold that wrapped below zero (a match at text position 0) travels through the sort's all-ones
    key; documents of different lengths; the order by absolute position (no documents) on the same handle before they are attached."""
    import torch
    S = synth
    monkeypatch.setenv("RBG_LOCATE_ORDER", "locus")   # (by default the locus order starts at 128 documents: capi/load.ipp upload_order_docs)
    with capi.default_option(capi.OPT_RANK_LAYOUT, layout), capi.default_option(capi.OPT_RUN_PHI, phi), capi.default_option(capi.OPT_POS_BYTES, pos_bytes):
        rb = ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0)
    o = orc.Oracle.from_runs(S.heads, S.lens, S.ssa, S.esa)
    reads = S.sample_reads(3000, 60, seed=8, sub_rate=0.1) + [bytes(S.text[:40]), bytes(S.text[:25]), bytes(S.text[1:30])]
    seqs, off = ra.pack_reads(reads)
    N = len(reads)
    dev = torch.device("cuda:0")
    d_seqs = torch.from_numpy(np.concatenate([seqs, np.zeros((-len(seqs)) % 16 + 16, np.uint8)])).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    d_lo, d_hi, d_k = (torch.empty(N, dtype=torch.int64, device=dev) for _ in range(3))
    st = torch.cuda.current_stream().cuda_stream
    L = ra.lib()
    assert L.rbg_find_range_w_toehold_dev(rb.h, d_seqs.data_ptr(), d_off.data_ptr(), N, d_lo.data_ptr(), d_hi.data_ptr(), d_k.data_ptr(), st) == 0
    wlo, whi, wk = o.find_range_w_toehold_batch(seqs, off, nthreads=4)
    assert (d_k.cpu().numpy().view(np.uint64) == wk).all()
    # a toehold that wrapped below zero (2^64 - 1: what LF_w_loc's k - 1 leaves of a toehold 0, rowbowt.hpp:561): planted on two reads with several
    # locations -- outside phi's domain, the reference's arithmetic on the last sampled position is followed (k_locate.hip phi_step), and the sort's key has no
    # document for it
    wk = wk.copy()
    multi = np.flatnonzero(whi - wlo + 1 >= 3)[:2]
    assert len(multi) == 2
    wk[multi] = np.uint64(MAXU)
    d_k.copy_(torch.from_numpy(wk.view(np.int64)))
    d_loc_off = torch.empty(N + 1, dtype=torch.int64, device=dev)
    tmp_bytes, ws_bytes = L.rbg_locate_plan_tmp_bytes(N), L.rbg_locate_order_ws_bytes(N)
    d_tmp = torch.empty(tmp_bytes, dtype=torch.uint8, device=dev)
    d_ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)

    def located(max_hits):
        assert L.rbg_locate_plan_dev(rb.h, d_lo.data_ptr(), d_hi.data_ptr(), N, max_hits, d_loc_off.data_ptr(), d_tmp.data_ptr(), tmp_bytes, st) == 0
        total = int(d_loc_off[-1].item())
        d_locs = torch.full((max(total, 1),), -1, dtype=torch.int64, device=dev)
        assert L.rbg_locate_order_dev(rb.h, d_k.data_ptr(), N, d_ws.data_ptr(), ws_bytes, st) == 0
        assert L.rbg_locate_fill_dev(rb.h, d_lo.data_ptr(), d_hi.data_ptr(), d_k.data_ptr(), N, max_hits, d_loc_off.data_ptr(), d_locs.data_ptr(), d_ws.data_ptr(), st) == 0
        torch.cuda.synchronize()
        perm = d_ws[:4 * N].view(torch.int32).cpu().numpy().astype(np.int64)
        return d_loc_off.cpu().numpy().view(np.uint64).copy(), d_locs[:total].cpu().numpy().view(np.uint64).copy(), perm

    want = {mh: o.locs_at_batch(wlo, whi, wk, mh, nthreads=4) for mh in (MAXU, 3, 1)}
    abs_perm = None
    for docs in (None, (S.doc_names, S.doc_starts), (["a", "b", "c"], [0, 7, S.n - 5])):   # (no documents; one per haplotype; three of very different lengths)
        if docs:
            rb.set_docs(*docs)
        for mh in (MAXU, 3, 1):
            g_off, g_locs, perm = located(mh)
            assert (g_off == want[mh][0]).all() and (g_locs == want[mh][1]).all(), (docs and len(docs[0]), mh)
            assert sorted(perm.tolist()) == list(range(N))          # the order is a permutation of the reads
        if docs is None:
            abs_perm = perm
        elif len(docs[0]) == S.H:
            # the locus order is another order: chains of DIFFERENT haplotypes at the same locus are neighbours in it, never in the absolute order
            unit = S.doc_starts[1]

            def cross(pm):
                live = np.array([i for i in pm if whi[i] >= wlo[i] and wk[i] < S.n])
                kk = wk[live].astype(np.int64)
                doc, offs = kk // unit, kk % unit
                return float(np.mean((doc[1:] != doc[:-1]) & (np.abs(np.diff(offs)) < 1024)))
            # (most toeholds of a locus share a haplotype -- SA[hi] is the haplotype sorted last there -- so a tenth of the neighbours crossing is a lot)
            assert (perm != abs_perm).any() and cross(perm) > 0.05 and cross(abs_perm) < 0.01, (cross(perm), cross(abs_perm))
    rb.close()
    o.close()


@pytest.mark.parametrize("layout,phi", [(capi.LAYOUT_SLOTS, 0), (capi.LAYOUT_RUNS, 2), (capi.LAYOUT_RUNS, 1)])
@pytest.mark.parametrize("pos_bytes", [4, 8, -8])
def test_k3_staging_at_its_edges(synth, layout, phi, pos_bytes, monkeypatch):
    """K3's staging (rbg_device.hpp ChainStage, round 6: values at the position width -- 8-byte positions below 2^40 as a low word + one high byte, or whole
    (pos_bytes -8) --, flush WINDOWS on 64-byte boundaries of the output array taken from a ring, the first location of a chain that is no text position
    stored by its owner): chains cut by max_hits at, just below and just above the rounds of 8 steps, at every alignment of a read's first location;
    toeholds that wrapped below zero by one AND by two (LF_w_loc's k - 1, rowbowt.hpp:561, once or twice past text position 0 -- at 4-byte
    positions the staged word for them is all ones, whatever the value) on chains of several locations; the ordered and the unordered walk;
    64-bit locations against the oracle (toehold_sa.hpp:37-49), 32-bit ones = their low words, nothing written outside a read's segment."""
    import subprocess
    import torch
    S = synth
    if pos_bytes < 0:
        # 8-byte positions staged WHOLE (what an index of 2^40 positions or more gets; below that a value travels as its low word + one high byte): the switch is read
        # once per process, so this case runs in a child
        if os.environ.get("RBG_K3_HI8") != "0":
            env = dict(os.environ, RBG_K3_HI8="0")
            r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", f"{__file__}::test_k3_staging_at_its_edges[-8-{layout}-{phi}]"], env=env,
                               capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
            return
        pos_bytes = 8
    with capi.default_option(capi.OPT_RANK_LAYOUT, layout), capi.default_option(capi.OPT_RUN_PHI, phi), capi.default_option(capi.OPT_POS_BYTES, pos_bytes):
        rb = ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0)
    o = orc.Oracle.from_runs(S.heads, S.lens, S.ssa, S.esa)
    reads = S.sample_reads(1500, 24, seed=77, sub_rate=0.05) + [bytes(S.text[:30]), b"", b"ACGTN"]   # (short reads: long chains)
    seqs, off = ra.pack_reads(reads)
    N = len(reads)
    wlo, whi, wk = o.find_range_w_toehold_batch(seqs, off, nthreads=4)
    occ = np.where(whi >= wlo, whi - wlo + 1, 0).astype(np.int64)
    assert occ.max() > 33                                          # (chains longer than two rounds of sixteen)
    wk = wk.copy()
    multi = np.flatnonzero(occ >= 3)
    wk[multi[0]] = np.uint64(MAXU)
    wk[multi[1]] = np.uint64(MAXU - 1)
    wk[multi[2]] = np.uint64(S.n + 5)                               # (no toehold the search produces; the walk's arithmetic is the reference's all the same)
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    L = ra.lib()
    d_lo, d_hi, d_k = (torch.from_numpy(a.view(np.int64).copy()).to(dev) for a in (wlo, whi, wk))
    d_loc_off = torch.empty(N + 1, dtype=torch.int64, device=dev)
    tmp_bytes, ws_bytes = L.rbg_locate_plan_tmp_bytes(N), L.rbg_locate_order_ws_bytes(N)
    d_tmp = torch.empty(tmp_bytes, dtype=torch.uint8, device=dev)
    d_ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    assert L.rbg_locate_order_dev(rb.h, d_k.data_ptr(), N, d_ws.data_ptr(), ws_bytes, st) == 0
    for mh in (1, 7, 8, 9, 15, 16, 17, 31, 32, 33, MAXU):
        woff, wlocs = o.locs_at_batch(wlo, whi, wk, mh, nthreads=4)
        assert L.rbg_locate_plan_dev(rb.h, d_lo.data_ptr(), d_hi.data_ptr(), N, mh, d_loc_off.data_ptr(), d_tmp.data_ptr(), tmp_bytes, st) == 0
        total = int(d_loc_off[-1].item())
        assert total == int(woff[-1])
        for order in (d_ws.data_ptr(), None):
            d_locs = torch.full((total + 2,), -7, dtype=torch.int64, device=dev)
            assert L.rbg_locate_fill_dev(rb.h, d_lo.data_ptr(), d_hi.data_ptr(), d_k.data_ptr(), N, mh, d_loc_off.data_ptr(), d_locs.data_ptr(), order, st) == 0
            torch.cuda.synchronize()
            assert (d_locs[:total].cpu().numpy().view(np.uint64) == wlocs).all(), (mh, order is None)
            assert d_locs[total:].tolist() == [-7, -7]
            if pos_bytes == 4:
                d_locs32 = torch.full((total + 2,), -7, dtype=torch.int32, device=dev)
                assert L.rbg_locate_fill_dev32(rb.h, d_lo.data_ptr(), d_hi.data_ptr(), d_k.data_ptr(), N, mh, d_loc_off.data_ptr(), d_locs32.data_ptr(), order, st) == 0
                torch.cuda.synchronize()
                assert (d_locs32[:total].cpu().numpy().view(np.uint32) == (wlocs & np.uint64(0xFFFFFFFF)).astype(np.uint32)).all(), (mh, order is None)
                assert d_locs32[total:].tolist() == [-7, -7]
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
    rb = _with_layout(layout, lambda: ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0))
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
        rb8 = _with_layout(layout, lambda: ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0))
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
    rb = _with_layout(layout, lambda: ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0))
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
