"""Second, independent line of defence: a naive explicit-text FM-index in numpy.

Nothing here follows the reference's data structures: it inverts the BWT to the text, builds
the suffix array by prefix doubling and answers queries by binary search over SA.  Toy sizes only.
"""
import numpy as np


def expand_bwt(heads, lens):
    return np.repeat(np.asarray(heads, dtype=np.uint8), np.asarray(lens, dtype=np.int64))


def invert_bwt(bwt):
    """BWT (terminator = smallest symbol, unique) -> text ending with the terminator."""
    n = len(bwt)
    order = np.argsort(bwt, kind="stable")  # order[j] = position in BWT of the j-th char of F
    lf = np.empty(n, dtype=np.int64)
    lf[order] = np.arange(n)
    text = np.empty(n, dtype=np.uint8)
    i = 0  # row 0 = suffix starting with the terminator; BWT[0] = char before terminator
    text[n - 1] = bwt.min()
    for k in range(n - 2, -1, -1):
        text[k] = bwt[i]
        i = lf[i]
    return text


def suffix_array(text):
    """Prefix doubling, O(n log^2 n) with numpy sorts.  text: uint8 array (any bytes)."""
    n = len(text)
    rank = np.unique(text, return_inverse=True)[1].astype(np.int64)  # dense: the doubling key needs ranks < n + 2
    k = 1
    sa = np.argsort(rank, kind="stable")
    while True:
        r2 = np.full(n, -1, dtype=np.int64)
        r2[: n - k] = rank[k:]
        key = rank * (n + 2) + (r2 + 1)
        sa = np.argsort(key, kind="stable")
        ks = key[sa]
        newrank = np.empty(n, dtype=np.int64)
        newrank[sa] = np.concatenate(([0], np.cumsum(ks[1:] != ks[:-1])))
        rank = newrank
        if rank.max() == n - 1:
            break
        k *= 2
    return sa


def bwt_from_sa(text, sa):
    return text[(sa - 1) % len(text)]


def rle(bwt):
    b = np.asarray(bwt)
    brk = np.flatnonzero(np.concatenate(([True], b[1:] != b[:-1])))
    heads = b[brk]
    lens = np.diff(np.concatenate((brk, [len(b)])))
    return heads.astype(np.uint8), lens.astype(np.uint64), brk


def run_samples(sa, brk, n):
    """Raw .ssa/.esa 'y' values: SA at the first / last position of every BWT run
    (the reference stores y ? y-1 : n-1, toehold_sa.hpp:139-140,151-152)."""
    starts = brk
    ends = np.concatenate((brk[1:], [n])) - 1
    return sa[starts].astype(np.uint64), sa[ends].astype(np.uint64)


class NaiveFM:
    def __init__(self, text):
        self.text = np.asarray(text, dtype=np.uint8)
        self.n = len(self.text)
        self.sa = suffix_array(self.text)
        self.tb = self.text.tobytes()

    def find_range(self, q):
        """Inclusive SA interval of q, or (1,0).  Plain binary search on suffixes."""
        m = len(q)
        if m == 0:
            return 0, self.n - 1
        tb, sa, n = self.tb, self.sa, self.n
        lo, hi = 0, n
        while lo < hi:
            mid = (lo + hi) // 2
            s = int(sa[mid])
            if tb[s:s + m] < q:
                lo = mid + 1
            else:
                hi = mid
        left = lo
        hi = n
        while lo < hi:
            mid = (lo + hi) // 2
            s = int(sa[mid])
            if tb[s:s + m] <= q:
                lo = mid + 1
            else:
                hi = mid
        right = lo
        if right <= left:
            return 1, 0
        return left, right - 1

    def locs(self, lo, hi):
        """SA[hi], SA[hi-1], ..., SA[lo]: the order ToeholdSA::locate_range emits."""
        if hi < lo:
            return []
        return self.sa[lo:hi + 1][::-1].tolist()


def greedy_marker_seed_bounds_literal(fm, q, wsize, K, stats=None):
    """Seed records {range, (m-i, seed_ei-1)} of RowBowt::get_markers_greedy_seeding (rowbowt.hpp:406-482)
    with an ftab of k-mer size K loaded (K == 0: none), written statement by statement after the
    reference -- including the restart loop :454-464 whose `else` branch is never taken, because
    search_ftab (:745-758) answers an absent k-mer with {full_range(), 0} and full_range() passes the
    `range.first <= range.second` test.  LF is "extend the pattern to the left and search again" on the
    explicit text; the ftab is the std::map build_ftab(K) makes (:726-744): k-mers over ACGT that occur.
    Markers are left out (this pins the control flow: ranges and seed bounds).  -> [(lo, hi, qs, qe_excl)]"""
    m = len(q)
    full = (0, fm.n - 1)

    def search_ftab(kmer):                                   # :745-758
        if len(kmer) != K:
            raise ValueError("only strings of size k are allowed for ftab queries")
        if all(c in b"ACGT" for c in kmer):
            r = fm.find_range(bytes(kmer))
            if r[0] <= r[1]:                                 # the k-mer is a key of the map
                return r, K
        return full, 0                                       # :757

    out = []
    prev_range = full                                        # :427
    rng = full                                               # :428
    pat = b""                                                # the pattern whose range `rng` is
    i = 0
    if K:                                                    # :430-433
        rng, i = search_ftab(q[m - K:])
        prev_range = rng
        pat = q[m - K:] if i else b""
    window_ei, seed_ei = m, m                                # :434
    while i < m:                                             # :442
        c = q[m - i - 1:m - i]
        npat = c + pat
        nr = fm.find_range(npat) if rng[0] <= rng[1] else (1, 0)   # :443 LF(range, c)
        if nr[1] < nr[0]:                                    # :444
            out.append((prev_range[0], prev_range[1], m - i, seed_ei))   # :448
            prev_range = full                                # :450
            seed_ei = m - i - 1                              # :452
            window_ei = m - i - 1
            if K and m - i - 1 >= K:                         # :454
                while m - i - 1 >= K:                        # :455
                    seed_ei = m - i - 1
                    window_ei = m - i - 1
                    kmer = q[m - i - 1 - K:m - i - 1]
                    rng, hit = search_ftab(kmer)             # :458
                    if stats is not None and not hit:
                        stats["restart_misses"] = stats.get("restart_misses", 0) + 1
                    if rng[0] <= rng[1]:                     # :459 (true for a miss as well)
                        pat = kmer if hit else b""
                        i += K                               # :460
                        prev_range = rng                     # :461
                        break
                    rng = full                               # :463 (unreachable)
                    pat = b""
                    i += 1
            else:
                rng = full                                   # :466
                pat = b""
        else:
            rng = nr
            pat = npat
            if window_ei - (m - i - 1) >= wsize:             # :469-472
                window_ei = m - i - 1
            prev_range = rng                                 # :473
        i += 1
    out.append((rng[0], rng[1], m - i, seed_ei))             # :481
    return out
