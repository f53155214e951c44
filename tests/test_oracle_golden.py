"""Pins the oracle (oracle/rb_oracle.c) against every golden value the reference's tests hold.

CPU only.  If these fail the oracle cannot be used as the parity checker.
"""
import os

import numpy as np
import pytest

import golden_values as G
import orc


@pytest.fixture(scope="module", params=["arrays", "reference_shaped"])
def small(data_dir, request):
    """both forms of the oracle reproduce every golden value: the decoded arrays, and the reference-shaped mode (Elias-Fano
    vectors + Huffman-shaped wavelet tree: what sdsl holds for the reference; rb_oracle.h orc_set_reference_shaped)"""
    o = orc.Oracle.load(os.path.join(data_dir, "small.fa"), orc.SA | orc.MA)
    if request.param == "reference_shaped":
        o.set_reference_shaped(True)
    yield o
    o.close()


@pytest.fixture(scope="module")
def simple_reads(data_dir):
    return orc.read_fastx(os.path.join(data_dir, "simple_query.fq"))[1]


@pytest.fixture(scope="module")
def error_reads(data_dir):
    return orc.read_fastx(os.path.join(data_dir, "error_query.fq"))[1]


def test_fixture_shape(small):
    # n=30031, R=7573, alphabet {0x01,A,C,G,T} counts 1/7649/7374/7580/7427
    assert small.n == 30031 and small.r == 7573
    f = small.f()
    assert int(f[1]) == 0 and int(f[2]) == 1
    assert [int(f[c]) for c in b"ACGT"] == [1, 7650, 15024, 22604]
    heads, lens = small.runs()
    assert int(lens.sum()) == 30031
    assert sorted(set(heads.tolist())) == [1, 65, 67, 71, 84]
    assert [int((heads == c).sum()) for c in (1, 65, 67, 71, 84)] == [1, 1929, 1833, 1917, 1893]
    assert (heads[1:] != heads[:-1]).all()


def test_count_golden(small, simple_reads):
    assert [small.find_range(q) for q in simple_reads] == G.SIMPLE_RANGES
    assert [small.count(q) for q in simple_reads] == [hi - lo + 1 for lo, hi in G.SIMPLE_RANGES]


def test_kmer_golden(small):
    for q, rng in G.KMER_RANGES.items():
        assert small.find_range(q) == rng


def test_locate_golden(small, simple_reads):
    all_locs = []
    for q, want in zip(simple_reads, G.SIMPLE_LOCS_PER_READ):
        lo, hi, k = small.find_range_w_toehold(q)
        assert (lo, hi) == small.find_range(q)
        locs = small.locs_at(lo, hi, k, G.MAXU)
        assert locs == want
        all_locs += locs
    assert all_locs == G.SIMPLE_ALL_LOCS


def test_locate_max_hits(small, simple_reads):
    lo, hi, k = small.find_range_w_toehold(simple_reads[2])
    assert small.locs_at(lo, hi, k, 2) == G.SIMPLE_LOCS_PER_READ[2][:2]
    assert small.locs_at(lo, hi, k, 0) == []


def test_error_reads(small, error_reads):
    # SURVEY 4.3 (verified, not asserted upstream): reads 1,2,5,6 -> (1,0); 3,4 -> (27430,27432)
    got = [small.find_range(q) for q in error_reads]
    assert got == [(1, 0), (1, 0), (27430, 27432), (27430, 27432), (1, 0), (1, 0)]
    # failure clears LFData: rn={1,0}, ssamp=0 (rowbowt.hpp:153-159,177-180)
    assert small.find_range_w_toehold(error_reads[0]) == (1, 0, 0)


def test_greedy_locate_golden(small, error_reads):
    for q, want in zip(error_reads, G.GREEDY_LOCS_PREFIX):
        locs, _seed = small.greedy_locate(q, 10)
        if want is None:
            assert locs == []
        else:
            assert locs[: len(want)] == want


def test_marker_golden(small, simple_reads):
    for q, want in zip(simple_reads, G.SIMPLE_FIRST_MARKER):
        _rng, mk = small.find_range_w_markers(q, 10, G.MAXU)
        if want is None:
            assert mk == []
        else:
            assert (G.get_pos(mk[0]), G.get_allele(mk[0])) == want


def test_greedy_seeding_fixture_loads(data_dir):
    o = orc.Oracle.load(os.path.join(data_dir, "greedy_seeding", "ref.fa"), orc.SA | orc.DL)
    assert o.n == 20047 and o.r == 14949
    assert o.resolve_offset(1234) == ("greedy_seeding", 1234)
    _names, reads = orc.read_fastx(os.path.join(data_dir, "greedy_seeding", "query.fq"))
    lo, hi, k = o.find_range_w_toehold(reads[0])
    assert hi >= lo
    locs = o.locs_at(lo, hi, k)
    assert len(locs) == hi - lo + 1
    o.close()


def test_rb_markers_model_pieces():
    """the pieces of the rb_markers model the CLI test leans on"""
    import rb_markers_model as RM
    rng = RM.MT19937()
    assert [rng() for _ in range(2)] == [3499211612, 581869302]   # std::mt19937's first outputs, default seed
    r = RM.MT19937()
    for _ in range(9999):
        r()
    assert r() == 4123659995                                        # the C++ standard's 10000th value
    assert RM.NT[ord("a")] == ord("A") and RM.NT[ord("N")] == ord("A") and RM.NT[ord("x")] == ord("N") and RM.NT[200] == ord("N")
    assert b"ACGTN".translate(RM.COMP) == b"TGCAN"
    mk = lambda seq, pos, al: (al << 60) | (seq << 48) | pos
    ms = [mk(0, 5, 0), mk(0, 5, 1), mk(0, 9, 0), mk(1, 9, 1)]
    assert RM.filter_identical_pos(ms) == [mk(0, 9, 0), mk(1, 9, 1)]
    assert RM.filter_identical_pos([mk(0, 0, 1), mk(0, 3, 0)]) == [mk(0, 3, 0)]   # pm starts as marker 0
    assert RM.clear_if_conflicting([mk(0, 5, 0), mk(0, 200, 0)], 101) == []
    assert RM.clear_if_conflicting([mk(0, 5, 0), mk(0, 100, 0)], 101) == [mk(0, 5, 0), mk(0, 100, 0)]
    assert RM.clear_if_conflicting([mk(0, 5, 0), mk(1, 6, 0)], 101) == []


def test_oracle_marker_seeds_fixture(small_oracle=None):
    """get_markers_greedy_seeding (rowbowt.hpp:406-482) on the shipped fixture: exact reads give one
    seed over the whole read whose range is find_range's; its markers include markers_at's"""
    import os
    import orc
    data = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
    o = orc.Oracle.load(os.path.join(data, "small.fa"), orc.SA | orc.MA)
    names, seqs = orc.read_fastx(os.path.join(data, "simple_query.fq"))
    for q in seqs:
        seeds = o.markers_greedy_seeding(q, 19, 1000)
        lo, hi = o.find_range(q)
        assert len(seeds) == 1 and seeds[0][:4] == (lo, hi, 0, len(q))
        assert set(o.markers_at(lo, hi)) <= set(seeds[0][4]) or hi - lo + 1 > 1000
    # a read with one error: two seeds, right one first, the failing base skipped
    names, seqs = orc.read_fastx(os.path.join(data, "error_query.fq"))
    seeds = o.markers_greedy_seeding(seqs[0], 19, 1000)
    assert [(s[2], s[3]) for s in seeds] == [(5, 20), (0, 4)]
    # every seed is an exact match of its piece of the read and maximal to the left
    for q in seqs:
        for lo, hi, qs, qe, _ in o.markers_greedy_seeding(q, 5, 1000):
            if qe > qs:
                assert o.find_range(q[qs:qe]) == (lo, hi)
                if qs > 0:
                    l2, h2 = o.find_range(q[qs - 1:qe])
                    assert h2 < l2
    o.close()


def test_oracle_reproduces_committed_golden_files(tmp_path):
    """tests/golden/*: regenerate with the generator script into a scratch directory and compare byte for byte"""
    import importlib.util
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(gold, "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main(str(tmp_path))
    made = sorted(os.listdir(tmp_path))
    assert made == ["toy_error_query_locate.json", "toy_k4.ftab", "toy_marker_seeds.json", "toy_rb_markers_default.txt",
                    "toy_rb_markers_heuristic.txt"]
    for f in made:
        assert open(tmp_path / f, "rb").read() == open(os.path.join(gold, f), "rb").read(), f
