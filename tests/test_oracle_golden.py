"""Pins the oracle (oracle/rb_oracle.c) against every golden value the reference's tests hold.

CPU only.  If these fail the oracle cannot be used as the parity checker.
"""
import os

import numpy as np
import pytest

import golden_values as G
import orc


@pytest.fixture(scope="module")
def small(data_dir):
    o = orc.Oracle.load(os.path.join(data_dir, "small.fa"), orc.SA | orc.MA)
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
