"""Golden values held by the reference's own test-suite (tests/rb_tests.cpp in the reference), loaded from
tests/golden/reference_rb_tests.json (which carries the line references).  Commented-out assertions of
that file are included: they are still known answers for the shipped fixtures.
"""
import json
import os

MAXU = 2**64 - 1
GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_R = json.load(open(os.path.join(GOLDEN_DIR, "reference_rb_tests.json")))

SIMPLE_RANGES = [tuple(r) for r in _R["count_simple_query"]["ranges"]]                  # :115-120 CountTester
SIMPLE_ALL_LOCS = _R["locate_simple_query"]["all_locs"]                                  # :47-58 LocateTester
SIMPLE_LOCS_PER_READ = _R["locate_simple_query"]["per_read"]
GREEDY_LOCS_PREFIX = _R["greedy_error_query"]["locs_prefix"]                             # :83-95 GreedyLocateTester (None = empty)
SIMPLE_FIRST_MARKER = [tuple(m) if m else None for m in _R["markers_simple_query"]["first_marker"]]   # :131-140 MarkerTester
KMER_RANGES = {k.encode(): tuple(v) for k, v in _R["ftab_kmers"]["ranges"].items()}      # :147-173 FTab tests


# marker_array.hpp (pfbwt-f) helpers used at rb_align.cpp:142, rb_tests.cpp:131-140; bit layout per SURVEY 8b-format
def get_pos(m):
    return m & ((1 << 48) - 1)


def get_allele(m):
    return (m >> 60) & 0xF
