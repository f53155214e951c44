"""Test infrastructure: a plain-Python restatement of what the reference's rb_markers worker does with
each read (reference src/rb_markers.cpp), on top of the oracle's get_markers_greedy_seeding.  The
expected stdout of the rb_markers-compatible CLI is computed with it.  Single worker thread order
(--threads 1), the only order the reference makes deterministic."""
import golden_values as G

M64 = 2**64 - 1


def get_seq(m):
    return (m >> 48) & 0xFFF  # presumed position of the sequence id


def nt_table():
    """seq_ntoa_table, rb_markers.cpp:139-156: ACGT (either case) kept, N/n -> A, everything else -> N"""
    t = [ord("N")] * 256
    for a, b in ((b"Aa", "A"), (b"Cc", "C"), (b"Gg", "G"), (b"Tt", "T"), (b"Nn", "A")):
        for c in a:
            t[c] = ord(b)
    return bytes(t)


NT = nt_table()
COMP = bytes.maketrans(b"ACGT", b"TGCA")  # comp_tab restricted to what NT can produce (N stays N)


class MT19937:
    """std::mt19937 with its default seed 5489 (RandomBoolGenerator's rng, rb_markers.cpp:210-225)"""

    def __init__(self, seed=5489):
        self.mt = [0] * 624
        self.mt[0] = seed
        for i in range(1, 624):
            self.mt[i] = (1812433253 * (self.mt[i - 1] ^ (self.mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
        self.idx = 624

    def __call__(self):
        if self.idx >= 624:
            mt = self.mt
            for i in range(624):
                y = (mt[i] & 0x80000000) | (mt[(i + 1) % 624] & 0x7FFFFFFF)
                mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)
            self.idx = 0
        y = self.mt[self.idx]
        self.idx += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF


class Booler:
    def __init__(self):
        self.rng, self.data, self.bits = MT19937(), 0, 0

    def get_bool(self):
        if self.bits == 0:
            self.data, self.bits = self.rng(), 32
        b = self.data & 1
        self.data >>= 1
        self.bits -= 1
        return bool(b)


def marker_key(m):  # marker_cmp, rb_markers.cpp:228-236
    return (get_seq(m), G.get_pos(m), G.get_allele(m))


def filter_identical_pos(markers):  # rb_markers.cpp:266-276 (remove_if with the look-ahead predicate)
    if not markers:
        return markers
    kept, pm = [], 0
    for i, m in enumerate(markers):
        if get_seq(m) == get_seq(pm) and G.get_pos(m) == G.get_pos(pm):
            continue
        pm = m
        if i + 1 < len(markers) and get_seq(markers[i + 1]) == get_seq(m) and G.get_pos(markers[i + 1]) == G.get_pos(m):
            continue
        kept.append(m)
    return kept


def clear_if_conflicting(markers, read_len):  # rb_markers.cpp:279-284
    if not markers:
        return markers
    if get_seq(markers[-1]) != get_seq(markers[0]) or ((G.get_pos(markers[-1]) - G.get_pos(markers[0])) & M64) >= read_len:
        return []
    return markers


def expected_stdout(o, records, wsize=19, max_range=1000, min_range=0, heuristic=False, best_strand=False, min_seed_len=0,
                    read_len=101, clear_conflicting=False, clear_identical=False, ftab_k=0):
    """records = [(name bytes, seq bytes)] in file order -> the text rb_markers prints"""
    out = []
    booler = Booler()
    for name, raw in records:
        fwd = raw.translate(NT)                     # rb_markers.cpp:396-398
        rev = fwd.translate(COMP)[::-1]             # :399-400
        n = len(fwd)
        seeds = []                                  # (strand, range_size, qstart, qlen, markers)

        def run(seq, strand):
            stop = False
            for lo, hi, qs, qe, mk in o.markers_greedy_seeding(seq, wsize, max_range, ftab_k):
                range_size = (hi - lo + 1) & M64
                qstart = ((n - qs - 1) & M64) if strand == "-" else qs        # :371
                qlen = (qe - qs) & M64                                         # :372
                if hi < lo or (heuristic and qlen < min_seed_len):             # :373 / :447
                    continue
                ms = []
                if range_size >= min_range and mk:                             # :374-380
                    ms = sorted(set(mk), key=marker_key)
                if heuristic:
                    if clear_conflicting:
                        ms = clear_if_conflicting(ms, read_len)
                    if clear_identical:
                        ms = filter_identical_pos(ms)
                seeds.append((strand, range_size, qstart, qlen, ms))
                if heuristic and best_strand and ((read_len - (qstart + qlen)) & M64) < min_seed_len:   # :460
                    stop = True
            return stop

        if not heuristic:                           # worker, :401-413
            run(fwd, "+")
            run(rev, "-")
        else:                                       # worker_heuristic, :483-503
            first_fwd = booler.get_bool()
            one, two = ((fwd, "+"), (rev, "-")) if first_fwd else ((rev, "-"), (fwd, "+"))
            if not run(*one):
                run(*two)
            if best_strand and seeds:               # :505, :292-314 (max_element keeps the first maximum)
                best = max(range(len(seeds)), key=lambda t: (seeds[t][3], -t))
                keep = seeds[best][0]
                seeds = [s for s in seeds if s[0] == keep]
            if min_seed_len:                        # :506
                seeds = [s for s in seeds if s[3] >= min_seed_len]
        for strand, range_size, qstart, qlen, ms in seeds:   # print_buf, :253-262
            line = f"{name.decode()} {range_size} {strand} {qstart} {qlen}"
            if ms:
                line += "".join(f" {get_seq(m)}/{G.get_pos(m)}/{G.get_allele(m)}" for m in ms)
            else:
                line += " ."
            out.append(line + "\n")
    return "".join(out)
