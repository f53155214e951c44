"""Oracle (reference-shaped C restatement) vs the naive explicit-text FM index, on the toy
fixture (config 1 of BASELINE.json: 10k synthetic 100 bp reads) and on random small texts."""
import os

import numpy as np
import pytest

import naive
import orc


def sample_reads(text, n_reads, m, rng, sub_rate=0.1, spans=None):
    """reads sampled from the text, 10% with one substitution."""
    L = len(text)
    reads = []
    for _ in range(n_reads):
        if spans:
            a, b = spans[rng.integers(len(spans))]
            s = int(rng.integers(a, b - m + 1))
        else:
            s = int(rng.integers(0, L - m))
        r = bytearray(text[s:s + m].tobytes())
        if rng.random() < sub_rate:
            p = int(rng.integers(m))
            r[p] = rng.choice([c for c in b"ACGT" if c != r[p]])
        reads.append(bytes(r))
    return reads


@pytest.fixture(scope="module")
def small_pair(data_dir):
    o = orc.Oracle.load(os.path.join(data_dir, "small.fa"), orc.SA | orc.MA)
    heads, lens = o.runs()
    text = naive.invert_bwt(naive.expand_bwt(heads, lens))
    fm = naive.NaiveFM(text)
    yield o, fm, text
    o.close()


def test_small_text_layout(small_pair, data_dir):
    # T = ref + A*10 + hap1 + A*10 + hap2 + A*10 + 0x01
    _o, _fm, text = small_pair
    fa = b"".join(open(os.path.join(data_dir, "small.fa"), "rb").read().split(b"\n")[1:])
    assert len(fa) == 10000
    t = text.tobytes()
    assert len(t) == 30031 and t[-1] == 1
    assert t[:10000] == fa and t[10000:10010] == b"A" * 10
    hap1 = bytearray(t[10010:20010])
    diffs = [i for i in range(10000) if hap1[i] != fa[i]]
    assert diffs == [289, 1859, 2239, 3193, 3734, 4121, 4650, 5500, 9035]


def test_bwt_matches_naive_sa(small_pair):
    o, fm, text = small_pair
    heads, lens = o.runs()
    assert (naive.bwt_from_sa(text, fm.sa) == naive.expand_bwt(heads, lens)).all()


def test_config1_10k_reads(small_pair):
    o, fm, text = small_pair
    rng = np.random.default_rng(20240231)
    spans = [(0, 10000), (10010, 20010), (20020, 30020)]
    reads = sample_reads(text, 10000, 100, rng, spans=spans)
    seqs, off = orc.pack_reads(reads)
    lo, hi = o.find_range_batch(seqs, off, nthreads=4)
    lo2, hi2, k = o.find_range_w_toehold_batch(seqs, off, nthreads=4)
    assert (lo == lo2).all() and (hi == hi2).all()
    n_empty = 0
    for i, q in enumerate(reads):
        want = fm.find_range(q)
        assert (int(lo[i]), int(hi[i])) == want
        if want == (1, 0):
            n_empty += 1
            assert int(k[i]) == 0
        else:
            assert int(k[i]) == int(fm.sa[want[1]])  # toehold = SA[hi]
    assert 500 < n_empty < 1500
    loc_off, locs = o.locs_at_batch(lo, hi, k, nthreads=4)
    for i in range(0, len(reads), 97):
        assert locs[int(loc_off[i]):int(loc_off[i + 1])].tolist() == fm.locs(int(lo[i]), int(hi[i]))


def test_primitives_vs_naive(small_pair):
    o, fm, text = small_pair
    heads, lens = o.runs()
    bwt = naive.expand_bwt(heads, lens)
    rng = np.random.default_rng(7)
    for c in (1, 65, 67, 71, 84):
        occ = np.concatenate(([0], np.cumsum(bwt == c)))
        pos_c = np.flatnonzero(bwt == c)
        for i in rng.integers(0, len(bwt) + 1, 200):
            assert o.rank(int(i), c) == int(occ[i])
        for j in rng.integers(0, len(pos_c), 100):
            assert o.select(int(j), c) == int(pos_c[j])
    run_id = np.repeat(np.arange(len(heads)), lens.astype(np.int64))
    for i in rng.integers(0, len(bwt), 300):
        assert o.access(int(i)) == int(bwt[i])
        assert o.run_of_position(int(i)) == int(run_id[i])
    # phi(SA[i]) == SA[i-1]
    for i in rng.integers(1, len(bwt), 300):
        assert o.phi(int(fm.sa[i])) == int(fm.sa[i - 1])
    assert o.rank(5, 0x4E) == 0  # absent symbol


@pytest.mark.parametrize("seed,n,sigma", [(1, 500, 2), (2, 2000, 4), (3, 3000, 3), (4, 1200, 4)])
def test_random_texts_build_from_runs(seed, n, sigma):
    """orc_build_from_runs (rle_string ctor + ToeholdSA build restated) on random repetitive texts."""
    rng = np.random.default_rng(seed)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)[:sigma]
    base = alpha[rng.integers(0, sigma, n // 4)]
    parts = []
    for _ in range(4):
        h = base.copy()
        for p in rng.integers(0, len(h), 5):
            h[p] = alpha[rng.integers(0, sigma)]
        parts.append(h)
    text = np.concatenate(parts + [np.array([1], dtype=np.uint8)])
    fm = naive.NaiveFM(text)
    bwt = naive.bwt_from_sa(text, fm.sa)
    heads, lens, brk = naive.rle(bwt)
    ssa, esa = naive.run_samples(fm.sa, brk, len(text))
    for B in (1, 2, 3):
        o = orc.Oracle.from_runs(heads, lens, ssa, esa, B=B)
        assert o.n == len(text) and o.r == len(heads)
        for _ in range(150):
            m = int(rng.integers(1, 40))
            s = int(rng.integers(0, len(text) - 1 - m))
            q = bytearray(text[s:s + m].tobytes())
            if rng.random() < 0.3:
                q[int(rng.integers(m))] = int(alpha[rng.integers(0, sigma)])
            q = bytes(q)
            want = fm.find_range(q)
            assert o.find_range(q) == want
            lo, hi, k = o.find_range_w_toehold(q)
            assert (lo, hi) == want
            if want != (1, 0):
                assert o.locs_at(lo, hi, k) == fm.locs(lo, hi)
                assert o.locs_at(lo, hi, k, 3) == fm.locs(lo, hi)[:3]
        o.close()


def _toy_fm(seed, n, sigma=4):
    rng = np.random.default_rng(seed)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)[:sigma]
    base = alpha[rng.integers(0, sigma, n // 3)]
    parts = []
    for _ in range(3):
        h = base.copy()
        for p in rng.integers(0, len(h), 4):
            h[p] = alpha[rng.integers(0, sigma)]
        parts.append(h)
    text = np.concatenate(parts + [np.array([1], dtype=np.uint8)])
    fm = naive.NaiveFM(text)
    heads, lens, brk = naive.rle(naive.bwt_from_sa(text, fm.sa))
    ssa, esa = naive.run_samples(fm.sa, brk, len(text))
    return rng, text, fm, orc.Oracle.from_runs(heads, lens, ssa, esa)


def test_marker_seeds_ftab_miss_walkthrough():
    """get_markers_greedy_seeding with an ftab whose restart k-mer is ABSENT (rowbowt.hpp:454-464).
    Derived by hand from the reference's statements on T = "ACACAC" + 0x01, K = 2, wsize = 1, read
    q = "GGTAC" (m = 5).  Suffixes in order: $ | AC$ ACAC$ ACACAC$ | C$ CAC$ CACAC$  -> "AC" = [1,3], "C" = [4,6].
      :430-433  search_ftab("AC") hits: range [1,3], i = 2, prev_range = [1,3]
      i = 2     LF([1,3], 'T') is empty -> fn(prev_range [1,3], (3, 4)): seed q[3,5) = "AC"; seed_ei = 2;
                m-i-1 = 2 >= K: search_ftab(q[0,2) = "GG") MISSES and returns {full_range(), 0} (:757); the
                test at :459 holds for [0,6], so i += 2 (-> 4), prev_range = full, break; the outer ++i -> 5
      i = 5     loop ends; :481 fn(range = [0,6], (m-i, seed_ei-1) = (0, 1)): the "seed" q[0,2) = "GG" is
                reported with the FULL range although GG does not occur -- the reference's behaviour."""
    text = np.frombuffer(b"ACACAC\x01", dtype=np.uint8)
    fm = naive.NaiveFM(text)
    assert fm.find_range(b"AC") == (1, 3) and fm.find_range(b"GG") == (1, 0)
    want = [(1, 3, 3, 5), (0, 6, 0, 2)]
    assert naive.greedy_marker_seed_bounds_literal(fm, b"GGTAC", 1, 2) == want
    heads, lens, brk = naive.rle(naive.bwt_from_sa(text, fm.sa))
    ssa, esa = naive.run_samples(fm.sa, brk, len(text))
    o = orc.Oracle.from_runs(heads, lens, ssa, esa)
    assert [tuple(s[:4]) for s in o.markers_greedy_seeding(b"GGTAC", 1, 1000, 2)] == want
    o.close()


@pytest.mark.parametrize("seed,n,K,wsize", [(11, 240, 5, 2), (12, 400, 6, 3), (13, 900, 6, 4), (14, 150, 4, 1)])
def test_marker_seeds_control_flow_vs_literal(seed, n, K, wsize):
    """the oracle's get_markers_greedy_seeding (restructured) against the statement-by-statement
    transliteration in naive.py, on reads with absent k-mers (texts this short miss many K-mers)"""
    rng, text, fm, o = _toy_fm(seed, n)
    stats = {}
    for _ in range(300):
        m = int(rng.integers(K + 1, 60))
        s = int(rng.integers(0, len(text) - 1 - m))
        q = bytearray(text[s:s + m].tobytes())
        for _ in range(int(rng.integers(0, 4))):
            q[int(rng.integers(m))] = int(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8)))
        q = bytes(q)
        for kk in (0, K):
            want = naive.greedy_marker_seed_bounds_literal(fm, q, wsize, kk, stats)
            got = [tuple(s_[:4]) for s_ in o.markers_greedy_seeding(q, wsize, 1000, kk)]
            assert got == want, (q, kk)
    assert stats.get("restart_misses", 0) > 20   # restarts on an absent k-mer really occurred
    o.close()


@pytest.mark.parametrize("seed,n,sigma", [(1, 400, 4), (2, 3000, 4), (3, 5000, 2), (4, 2500, 20), (5, 64, 1), (6, 6000, 200)])
def test_reference_shaped_mode_equals_the_arrays(seed, n, sigma):
    """orc_set_reference_shaped: Elias-Fano rank / select and wavelet-tree access / rank / select give what the decoded
    arrays give, on random repetitive texts of one to 200 symbols and block sizes B = 1, 2, 5: ranges, toeholds, locations
    (i.e. rank, select, run_of_position, phi) of sampled and random reads, with the mode on, and off again; and the
    explicit-text FM index agrees."""
    rng = np.random.default_rng(seed)
    alpha = rng.choice(np.arange(2, 255), size=sigma, replace=False).astype(np.uint8)
    base = alpha[rng.integers(0, sigma, n // 4)]
    parts = []
    for _ in range(4):
        h = base.copy()
        for p in rng.integers(0, len(h), 6):
            h[p] = alpha[rng.integers(0, sigma)]
        parts.append(h)
    text = np.concatenate(parts + [np.array([1], dtype=np.uint8)])
    fm = naive.NaiveFM(text)
    heads, lens, brk = naive.rle(naive.bwt_from_sa(text, fm.sa))
    ssa, esa = naive.run_samples(fm.sa, brk, len(text))
    reads = []
    for _ in range(250):
        m = int(rng.integers(1, 40))
        s0 = int(rng.integers(0, max(1, len(text) - 1 - m)))
        q = bytearray(text[s0:s0 + m].tobytes())
        if rng.random() < 0.3:
            q[int(rng.integers(len(q)))] = int(alpha[rng.integers(0, sigma)])
        reads.append(bytes(q))
    for B in (1, 2, 5):
        o = orc.Oracle.from_runs(heads, lens, ssa, esa, B=B)

        def answers():
            out = []
            for q in reads:
                lo, hi, k = o.find_range_w_toehold(q)
                out.append((lo, hi, k, tuple(o.locs_at(lo, hi, k, 25)) if lo <= hi else ()))
            return out

        plain = answers()
        o.set_reference_shaped(True)
        shaped = answers()
        for i in rng.integers(0, len(text) + 1, 200):   # the primitives themselves
            for c in (int(alpha[0]), int(alpha[-1]), 1):
                assert o.rank(int(i), c) == int(np.count_nonzero(naive.bwt_from_sa(text, fm.sa)[: int(i)] == c))
        o.set_reference_shaped(False)
        again = answers()
        assert shaped == plain and again == plain
        for q, a in zip(reads[:60], plain):
            assert (a[0], a[1]) == fm.find_range(q)
        assert sum(1 for a in plain if a[0] <= a[1]) > 100
        o.close()
