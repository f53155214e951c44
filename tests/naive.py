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
