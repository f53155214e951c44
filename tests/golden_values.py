"""Golden values held by the reference's own test-suite (tests/rb_tests.cpp in the reference).

line refs are to the reference's tests/rb_tests.cpp.  Commented-out assertions in that file are
included: they are still known answers for the shipped fixtures.
"""
MAXU = 2**64 - 1

# :115-120 CountTester, find_range on simple_query.fq
SIMPLE_RANGES = [(24279, 24280), (24175, 24175), (27430, 27432), (27430, 27432), (17409, 17409), (17416, 17417)]

# :47-58 LocateTester, find_range_w_toehold + locs_at(max_hits=-1), concatenated over reads
SIMPLE_ALL_LOCS = [20306, 286, 10296, 11897, 21907, 1887, 11897, 21907, 1887, 4644, 14654, 24664]
SIMPLE_LOCS_PER_READ = [[20306, 286], [10296], [11897, 21907, 1887], [11897, 21907, 1887], [4644], [14654, 24664]]

# :83-95 GreedyLocateTester, get_seeds_greedy_w_sample(seq,10) -> locate_from_longest_seed(-1) on error_query.fq
# (None = "size()==0"; only the asserted prefix of each list is held by the reference)
GREEDY_LOCS_PREFIX = [[10296, 20306, 286], [10296], [11897, 21907, 1887], [11897, 21907, 1887], None, [14654, 4644]]

# :131-140 MarkerTester, find_range_w_markers(seq, 10, -1): first marker (pos, allele) or None for empty
SIMPLE_FIRST_MARKER = [(289, 0), (289, 1), None, None, (4650, 0), (4650, 1)]

# :147-173 FTab tests (disabled upstream, "take too long"); ftab is result-neutral so these are find_range answers
KMER_RANGES = {
    b"TTCGTCGTAA": (28942, 28944),
    b"CCGCGGACAT": (10673, 10675),
    b"GGCAGGCGGA": (19418, 19423),
    b"TATCGTGGAA": (24272, 24274),
    b"GTATCGTGGAA": (21142, 21144),
    b"GGAGATATTG": (19097, 19099),
    b"TGGAGATATTG": (27180, 27182),
}


# marker_array.hpp (pfbwt-f) helpers used at rb_align.cpp:142, rb_tests.cpp:131-140; bit layout per SURVEY 8b-format
def get_pos(m):
    return m & ((1 << 48) - 1)


def get_allele(m):
    return (m >> 60) & 0xF
