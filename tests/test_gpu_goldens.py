"""GPU parity (through the C-ABI, bit-exact against the oracle; needs an MI355X): the reference's own golden vectors, config 1, error reads, tiny / random indexes through the HIP path."""
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
    slot tables, the run-indexed layout at several k-mer depths (up to eight symbols per step) with directories and with bucket records, and 8-byte positions -- all equal to
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
    # rec: RBG_OPT_RUN_REC (1 = directories over the run lists, 2 = bucket records, 0 = the library's choice)
    for layout, ks, pb, rec in ((capi.LAYOUT_SLOTS, 5, 0, 0), (capi.LAYOUT_RUNS, 5, 0, 1), (capi.LAYOUT_RUNS, 2, 8, 1), (capi.LAYOUT_SLOTS, 3, 8, 0),
                                (capi.LAYOUT_RUNS, 8, 0, 2), (capi.LAYOUT_RUNS, 3, 8, 2), (capi.LAYOUT_RUNS, 8, 8, 0), (capi.LAYOUT_RUNS, 6, 0, 1)):
        ra.set_default_option(capi.OPT_KMER_STEPS, ks)
        ra.set_default_option(capi.OPT_POS_BYTES, pb)
        ra.set_default_option(capi.OPT_RUN_REC, rec)
        try:
            rb = _with_layout(layout, lambda: ra.RowBowt.from_runs(heads, lens, ssa, esa, device=0))
        finally:
            ra.set_default_option(capi.OPT_KMER_STEPS, DEFAULT_KMER_STEPS)
            ra.set_default_option(capi.OPT_POS_BYTES, 0)
            ra.set_default_option(capi.OPT_RUN_REC, 0)
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
    rbr = _with_layout(capi.LAYOUT_RUNS, lambda: ra.RowBowt.from_runs(heads, lens, ssa, esa, device=0))
    lo, hi, k = rbr.find_range_w_toehold(seqs, off)
    assert (lo == wlo).all() and (hi == whi).all() and (k == wk).all()
    loc_off, locs = rbr.locs_at(lo, hi, k)
    assert (loc_off == woff).all() and (locs == wlocs).all()
    rbr.close()
    o.close()


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
