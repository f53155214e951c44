"""Test-side WRITER of the reference's on-disk index formats (.rbwt, .tsa, .mab), so that the sdsl-format reader of the library
(rowbowt_amd/csrc/rbg_host.cpp parse_rbwt / parse_tsa) can be pinned at sizes the reference's toolchain -- sdsl-lite is an
empty submodule in /root/reference, its build needs cmake and three more un-vendored libraries -- cannot produce here.

What is written (all little-endian):
  .rbwt  rle_string::serialize, rle_string.hpp:248-262: n, R, B, `runs` (sparse_sd_vector: a bit at the last position of every B-th
         run, the final run excepted, :58-80), 256 sparse_sd_vectors (one per byte value: in the concatenation of that letter's runs, a
         bit at each run's last position), the run heads as an sdsl::wt_huff<> (huff_string.hpp:54-63)
  .tsa   ToeholdSA::serialize, toehold_sa.hpp:74-82: r, n, the sampled positions as a sparse_sd_vector, samples_last and pred_to_run as
         sdsl::int_vector<0> of widths bitsize(n) and bitsize(r)
  sparse_sd_vector::serialize (sparse_sd_vector.hpp:182-192): u, then -- unless u == 0 -- sdsl::sd_vector<>: size, wl (one byte), low
         (int_vector<0> of width wl), high (bit_vector of m + 2^logm bits), select_support_mcl<1> and <0> over high
  sdsl::int_vector<w>: one u64 (bits in the low 56 bits, the element width in the top byte), ceil(bits / 64) words
  sdsl::select_support_mcl<b>: arg_cnt; if non-zero the superblock vector (position of every 4096th argument, width
         hi(capacity) + 1), the mini_or_long bit vector (empty when no superblock is long), one vector per superblock: 64 offsets of every
         64th argument (width hi(span) + 1), or -- a superblock spanning more than logn^4 bits -- all its 4096 positions
  sdsl::wt_huff<>: size, sigma, the concatenated node bit vectors in breadth-first node order, rank_support_v (two words per 512 bits:
         ones before the block; seven 9-bit counts of the ones before each later word of the block), select supports for 1 and 0,
         the tree: node count, 22-byte nodes {bv_pos, bv_pos_rank (leaves: bv size, the symbol), parent, child[2]}, c_to_leaf[256],
         path[256] (length in the top byte, branch bits below; an absent symbol: the largest present symbol below it)

  .mab   pfbwt-f MarkerArray (rowbowt_io.hpp:185 loads it): mab_bytes below
PINNED: re-serialising the decoded toy index reproduces tests/data/small.fa.rbwt, small.fa.tsa and small.fa.mab byte for byte
(tests/test_sdsl_writer.py).  NOT pinned by any fixture (none has them): long superblocks of a select support, Huffman ties between equal
frequencies -- the reader skips the former and decodes any valid tree, so neither matters to what the writer is for.
Everything is numpy: an index of r = 2e6 runs and n > 2^32 is written in seconds.
"""
import struct

import numpy as np

U64 = np.uint64


def hi(x):
    """sdsl::bits::hi: position of the most significant set bit (0 for 0)"""
    return max(int(x).bit_length() - 1, 0)


def pack_bits(positions, nbits):
    """bit vector of nbits bits with ones at `positions` (ascending) -> u64 words"""
    nwords = (nbits + 63) // 64
    words = np.zeros(nwords, dtype=U64)
    if len(positions):
        p = np.asarray(positions, dtype=U64)
        np.bitwise_or.at(words, (p >> U64(6)).astype(np.int64), U64(1) << (p & U64(63)))
    return words


def int_vector_bytes(values, width):
    """sdsl::int_vector<0> of the given element width"""
    v = np.asarray(values, dtype=U64)
    n = len(v)
    bits = n * width
    nwords = (bits + 63) // 64
    words = np.zeros(nwords + 1, dtype=U64)
    if n:
        b = np.arange(n, dtype=U64) * U64(width)
        w, sh = (b >> U64(6)).astype(np.int64), b & U64(63)
        np.bitwise_or.at(words, w, v << sh)            # (numpy: a u64 shift drops what leaves the word)
        spill = (sh + U64(width)) > U64(64)
        if spill.any():
            np.bitwise_or.at(words, w[spill] + 1, v[spill] >> (U64(64) - sh[spill]))
    return struct.pack("<Q", bits | (width << 56)) + words[:nwords].tobytes()


def bit_vector_bytes(words, nbits):
    return struct.pack("<Q", nbits | (1 << 56)) + np.asarray(words[:(nbits + 63) // 64], dtype=U64).tobytes()


def select_support_bytes(args, nbits):
    """sdsl::select_support_mcl over a bit vector of nbits bits whose arguments (ones, or zeros) sit at `args` (ascending)"""
    args = np.asarray(args, dtype=np.int64)
    cnt = len(args)
    out = [struct.pack("<Q", cnt)]
    if cnt == 0:
        return out[0]
    capacity = ((nbits + 63) >> 6) << 6
    logn = hi(capacity) + 1
    logn4 = logn ** 4
    sb = (cnt + 4095) >> 12
    out.append(int_vector_bytes(args[::4096], logn))
    blocks, is_long = [], []
    for i in range(sb):
        a = args[i * 4096:(i + 1) * 4096]
        span = int(a[-1] - a[0])
        if span > logn4:        # (long superblock: every position; not met by any fixture)
            full = np.zeros(4096, dtype=np.int64)
            full[:len(a)] = a
            blocks.append(int_vector_bytes(full, hi(int(a[-1])) + 1))
            is_long.append(True)
        else:
            mini = np.zeros(64, dtype=np.int64)
            mini[:(len(a) + 63) // 64] = a[::64] - a[0]
            blocks.append(int_vector_bytes(mini, hi(span) + 1))
            is_long.append(False)
    if any(is_long):
        out.append(bit_vector_bytes(pack_bits([i for i in range(sb) if not is_long[i]], sb), sb))   # (bit i: superblock i has mini blocks)
    else:
        out.append(bit_vector_bytes(np.zeros(0, dtype=U64), 0))
    out += blocks
    return b"".join(out)


def sd_vector_bytes(ones, u):
    """sdsl::sd_vector<> of a bit vector of u bits with ones at `ones` (ascending)"""
    ones = np.asarray(ones, dtype=U64)
    m = len(ones)
    logu, logm = hi(u) + 1, hi(m) + 1
    if logm == logu:
        logm -= 1
    wl = logu - logm
    low = ones & U64((1 << wl) - 1)
    high_pos = ((ones >> U64(wl)) + np.arange(m, dtype=U64)).astype(np.int64)
    nhigh = m + (1 << logm)
    high_words = pack_bits(high_pos, nhigh)
    is_one = np.zeros(nhigh, dtype=bool)
    is_one[high_pos] = True
    zeros = np.flatnonzero(~is_one)
    return (struct.pack("<QB", u, wl) + int_vector_bytes(low, wl) + bit_vector_bytes(high_words, nhigh) +
            select_support_bytes(high_pos, nhigh) + select_support_bytes(zeros, nhigh))


def sparse_bytes(ones, u):
    """ri::sparse_sd_vector::serialize"""
    return struct.pack("<Q", u) + (sd_vector_bytes(ones, u) if u else b"")


def rank_support_v_bytes(words, nbits):
    """sdsl::rank_support_v<1>: int_vector<64> of ((capacity >> 9) + 1) * 2 words"""
    capacity = ((nbits + 63) >> 6) << 6
    nblocks = (capacity >> 9) + 1
    w = np.zeros(nblocks * 8, dtype=U64)
    w[:len(words)] = words
    pc = np.array([bin(int(x)).count("1") for x in w], dtype=np.int64) if len(w) < 4096 else _popcount(w)
    pc = pc.reshape(nblocks, 8)
    before_block = np.concatenate([[0], np.cumsum(pc.sum(axis=1))[:-1]])
    inner = np.cumsum(pc, axis=1)[:, :7]                   # ones before word 1 .. 7 of the block
    second = np.zeros(nblocks, dtype=U64)
    nwords_cap = capacity >> 6
    first_word = np.arange(nblocks) * 8
    for i in range(7):          # (the counts stop with the vector's last word: the fields of words beyond it stay 0)
        live = (first_word + i + 1) <= nwords_cap
        second |= np.where(live, inner[:, i], 0).astype(U64) << U64(63 - 9 * (i + 1))
    bb = np.empty(nblocks * 2, dtype=U64)
    bb[0::2] = before_block.astype(U64)
    bb[1::2] = second
    return struct.pack("<Q", len(bb) * 64 | (64 << 56)) + bb.tobytes()


def _popcount(w):
    x = w.copy()
    x = x - ((x >> U64(1)) & U64(0x5555555555555555))
    x = (x & U64(0x3333333333333333)) + ((x >> U64(2)) & U64(0x3333333333333333))
    x = (x + (x >> U64(4))) & U64(0x0F0F0F0F0F0F0F0F)
    return ((x * U64(0x0101010101010101)) >> U64(56)).astype(np.int64)


def wt_huff_bytes(seq):
    """sdsl::wt_huff<> of a byte sequence (at least two distinct symbols)"""
    seq = np.asarray(seq, dtype=np.uint8)
    size = len(seq)
    freq = np.bincount(seq, minlength=256)
    syms = [int(c) for c in np.flatnonzero(freq)]
    sigma = len(syms)
    assert sigma >= 2, "a run-length BWT has at least two distinct heads"
    # Huffman: the two least frequent nodes merge, the lesser becomes child 0 (ties: the node made first)
    import heapq
    heap = [(int(freq[c]), i) for i, c in enumerate(syms)]
    heapq.heapify(heap)
    child = {}                        # temporary node id -> (child0, child1); leaves are ids < sigma
    nxt = sigma
    while len(heap) > 1:
        f0, a = heapq.heappop(heap)
        f1, b = heapq.heappop(heap)
        child[nxt] = (a, b)
        heapq.heappush(heap, (f0 + f1, nxt))
        nxt += 1
    root = heap[0][1]
    # breadth-first numbering from the root
    order, queue = [], [root]
    while queue:
        t = queue.pop(0)
        order.append(t)
        if t in child:
            queue += list(child[t])
    num = {t: i for i, t in enumerate(order)}
    nn = len(order)
    parent = {root: 0xFFFF}
    for t, (a, b) in child.items():
        parent[a] = parent[b] = num[t]
    # the sequence of every internal node and its bits, in node order
    sub = {root: seq}
    bits_parts, bv_pos, ones_before = [], {}, {}
    total_bits = total_ones = 0
    leaves_under = {}

    def leaves(t):
        if t not in leaves_under:
            leaves_under[t] = [syms[t]] if t < sigma else leaves(child[t][0]) + leaves(child[t][1])
        return leaves_under[t]
    for t in order:
        if t not in child:
            continue
        s = sub.pop(t)
        right = np.zeros(256, dtype=bool)
        right[leaves(child[t][1])] = True
        b = right[s]
        bv_pos[t], ones_before[t] = total_bits, total_ones
        bits_parts.append(b)
        total_bits += len(b)
        total_ones += int(b.sum())
        sub[child[t][0]], sub[child[t][1]] = s[~b], s[b]
    allbits = np.concatenate(bits_parts) if bits_parts else np.zeros(0, dtype=bool)
    ones_pos = np.flatnonzero(allbits)
    zeros_pos = np.flatnonzero(~allbits)
    words = pack_bits(ones_pos, total_bits)
    out = [struct.pack("<QQ", size, sigma), bit_vector_bytes(words, total_bits), rank_support_v_bytes(words, total_bits),
           select_support_bytes(ones_pos, total_bits), select_support_bytes(zeros_pos, total_bits), struct.pack("<Q", nn)]
    for t in order:
        if t in child:
            out.append(struct.pack("<QQHHH", bv_pos[t], ones_before[t], parent[t], num[child[t][0]], num[child[t][1]]))
        else:
            out.append(struct.pack("<QQHHH", total_bits, syms[t], parent[t], 0xFFFF, 0xFFFF))
    c_to_leaf = [0xFFFF] * 256
    path = [0] * 256
    for i, c in enumerate(syms):
        c_to_leaf[c] = num[i]
        t, bits_, ln = i, [], 0
        while t != root:                     # walk up: which child of its parent
            p = order[parent[t]]
            bits_.append(1 if child[p][1] == t else 0)
            t = p
        bits_.reverse()                      # bit 0 = the branch taken at the root
        path[c] = (len(bits_) << 56) | sum(b << k for k, b in enumerate(bits_))
    prev = 0
    for c in range(256):
        if freq[c]:
            prev = c
        else:
            path[c] = prev                   # (an absent symbol: the largest present one below it; length 0)
    out.append(struct.pack("<256H", *c_to_leaf))
    out.append(struct.pack("<256Q", *path))
    return b"".join(out)


def rbwt_bytes(heads, lens, B=2):
    """rle_string::serialize of the run-length BWT with these run heads and lengths"""
    heads = np.asarray(heads, dtype=np.uint8)
    lens = np.asarray(lens, dtype=np.int64)
    R, n = len(heads), int(lens.sum())
    ends = np.cumsum(lens) - 1
    j = np.arange(R)
    block = (j % B == B - 1) & (j != R - 1)
    out = [struct.pack("<QQQ", n, R, B), sparse_bytes(ends[block], n)]
    for c in range(256):
        mine = heads == c
        if not mine.any():
            out.append(struct.pack("<Q", 0))
            continue
        lc = lens[mine]
        out.append(sparse_bytes(np.cumsum(lc) - 1, int(lc.sum())))
    out.append(wt_huff_bytes(heads))
    return b"".join(out)


def tsa_bytes(n, pred_pos, samples_last, pred_to_run):
    """ToeholdSA::serialize: r, n, the sampled positions (ascending), samples_last (by run), pred_to_run (by rank of the position)"""
    r = len(pred_pos)
    return (struct.pack("<QQ", r, n) + sparse_bytes(pred_pos, n) + int_vector_bytes(samples_last, hi(n) + 1) +
            int_vector_bytes(pred_to_run, hi(r) + 1))


def tsa_arrays_from_samples(n, ssa_y, esa_y):
    """the three arrays of ToeholdSA from the second values of a BWT's .ssa / .esa pairs, one per run (toehold_sa.hpp:133-155: sample =
    y ? y - 1 : n - 1; build_phi :105-131): pred_pos = the run starts' text positions in ascending order, pred_to_run = the run each belongs to,
    samples_last = the sample at each run's END, by run"""
    ssa = np.asarray(ssa_y, dtype=np.int64)
    esa = np.asarray(esa_y, dtype=np.int64)
    key = np.where(ssa > 0, ssa - 1, n - 1)
    last = np.where(esa > 0, esa - 1, n - 1)
    order = np.argsort(key, kind="stable")
    return key[order], last, order


# ---- decoders of the same structures (toy sizes: what the byte-for-byte test re-serialises) ----------------------------------------
def mab_bytes(starts, ends, off, vals, wsize, universe=None):
    """pfbwt-f's MarkerArray as rb_build writes it and load_rowbowt reads it (rowbowt_io.hpp:185; SURVEY 8b-format, inferred from the shipped
    small.fa.mab and reproduced byte for byte below): three raw sdsl::sd_vector<> -- the first and the last BWT row of every marker run
    (inclusive intervals, ascending, disjoint) over one universe, and a bit at the index of every run's first value over a universe of
    `count` values -- then u64 count, count x u64 MarkerT, i32 window size.  `off`: nruns + 1 offsets into vals (off[-1] == len(vals)).
    universe: the fixture's is its last run end + 2 (29 598 + 2; n = 30 031 there): taken as the default, the reader does not use it."""
    starts, ends, off = (np.asarray(a, dtype=U64) for a in (starts, ends, off))
    vals = np.ascontiguousarray(np.asarray(vals, dtype="<u8"))
    assert len(starts) == len(ends) == len(off) - 1 and int(off[-1]) == len(vals)
    if universe is None:
        universe = int(ends[-1]) + 2 if len(ends) else 0
    return (sd_vector_bytes(starts, universe) + sd_vector_bytes(ends, universe) + sd_vector_bytes(off[:-1], len(vals)) +
            struct.pack("<Q", len(vals)) + vals.tobytes() + struct.pack("<i", int(wsize)))


def decode_mab(data):
    """-> universe, starts, ends, off (nruns + 1), vals (uint64 array), wsize"""
    c = _Cur(data)
    u0, s = c.sd_vector()
    u1, e = c.sd_vector()
    u2, f = c.sd_vector()
    cnt = c.u64()
    vals = np.frombuffer(c.d, dtype="<u8", count=cnt, offset=c.p).copy()
    c.p += cnt * 8
    wsize = struct.unpack_from("<i", c.d, c.p)[0]
    c.p += 4
    assert c.p == len(data) and u0 == u1 and u2 == cnt and len(s) == len(e) == len(f)
    return u0, np.array(s, dtype=np.int64), np.array(e, dtype=np.int64), np.array(f + [cnt], dtype=np.int64), vals, wsize


class _Cur:
    def __init__(self, data):
        self.d, self.p = data, 0

    def u64(self):
        v = struct.unpack_from("<Q", self.d, self.p)[0]
        self.p += 8
        return v

    def u8(self):
        v = self.d[self.p]
        self.p += 1
        return v

    def int_vector(self):
        h = self.u64()
        bits, w = h & ((1 << 56) - 1), h >> 56
        nw = (bits + 63) // 64
        words = np.frombuffer(self.d, dtype="<u8", count=nw, offset=self.p)
        self.p += nw * 8
        n = bits // w
        big = int.from_bytes(words.tobytes(), "little")
        return [(big >> (i * w)) & ((1 << w) - 1) for i in range(n)], bits, w

    def skip_select(self):
        cnt = self.u64()
        if cnt:
            self.int_vector()
            self.int_vector()
            for _ in range((cnt + 4095) >> 12):
                self.int_vector()

    def sd_vector(self):
        u, wl = self.u64(), self.u8()
        low, _, _ = self.int_vector()
        high, _, _ = self.int_vector()
        self.skip_select()
        self.skip_select()
        ones, k = [], 0
        for pos, b in enumerate(high):
            if b:
                ones.append(((pos - k) << wl) | low[k])
                k += 1
        return u, ones

    def sparse(self):
        u = self.u64()
        if u == 0:
            return 0, []
        inner, ones = self.sd_vector()
        assert inner == u
        return u, ones


def decode_rbwt(data):
    """-> n, R, B, heads (uint8 array), lens (int64 array)"""
    c = _Cur(data)
    n, R, B = c.u64(), c.u64(), c.u64()
    c.sparse()
    letters = [c.sparse() for _ in range(256)]
    size, _sigma = c.u64(), c.u64()
    bv, nbits, _ = c.int_vector()
    c.int_vector()
    c.skip_select()
    c.skip_select()
    nn = c.u64()
    nodes = []
    for _ in range(nn):
        nodes.append(struct.unpack_from("<QQHHH", c.d, c.p))
        c.p += 22
    c.p += 512 + 2048
    assert c.p == len(data) and size == R
    pre = np.concatenate([[0], np.cumsum(bv)])
    heads = np.zeros(R, dtype=np.uint8)
    for i in range(R):
        v, pos = 0, i
        while nodes[v][3] != 0xFFFF:
            p = nodes[v][0] + pos
            ones_in = int(pre[p]) - nodes[v][1]
            b = bv[p]
            pos = ones_in if b else pos - ones_in
            v = nodes[v][3 + b]
        heads[i] = nodes[v][1]
    lens = np.zeros(R, dtype=np.int64)
    nxt = [0] * 256
    for i in range(R):
        s = int(heads[i])
        ones = letters[s][1]
        k = nxt[s]
        lens[i] = ones[k] - ones[k - 1] if k else ones[0] + 1
        nxt[s] += 1
    return n, R, B, heads, lens


def decode_tsa(data):
    """-> r, n, pred_pos, samples_last, pred_to_run"""
    c = _Cur(data)
    r, n = c.u64(), c.u64()
    _u, pred = c.sparse()
    sl, _, _ = c.int_vector()
    p2r, _, _ = c.int_vector()
    assert c.p == len(data)
    return r, n, np.array(pred, dtype=np.int64), np.array(sl, dtype=np.int64), np.array(p2r, dtype=np.int64)
