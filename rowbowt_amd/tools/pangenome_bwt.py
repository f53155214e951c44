"""Run-length BWT + run-boundary SA samples of a synthetic pangenome WITHOUT a suffix array of the text.

synth_pangenome.suffix_array (prefix doubling) needs ~50 bytes per text symbol and 64-bit sort keys of n^2: it
stops near n = 3e9.  A pangenome text is H near-copies of one base sequence, and that structure gives the
suffix order directly (exactly, not approximately):

  * the first K = 32 characters of the suffix at (haplotype h, offset p) are the base sequence's 32-mer at p with
    the alternative alleles h carries at the variant sites inside [p, p+K) -- so suffixes fall into CLASSES
    (p, local allele pattern), a few per offset, and a class's rank among all classes is the rank of its
    2-bit-packed 32-mer (one 64-bit radix sort of ~1.4 L keys; the build aborts if two classes of different
    offsets share a 32-mer, which a random base sequence makes a < 1e-3 event up to L = 1e9);
  * inside a class all suffixes start at the same offset and agree on K characters; they are ordered by what
    follows, i.e. by the haplotypes' alleles at the sites from p+K on and then by what follows the haplotype in
    the text -- the positional-BWT order computed once, right to left, over the sites (pbwt.c);
  * the BWT character of a suffix is the character before it: the base character at p-1, unless p-1 is a
    variant site (then it depends on the haplotype: those classes are expanded member by member) or p = 0.

Runs, run heads and the SA values at run boundaries (the reference's .bwt/.ssa/.esa build inputs,
rb_build.cpp:83-93) are read off the sorted classes: O(L + S*H) work for a text of n = H*(L+pad)+1 symbols.
tests/test_pangenome_bwt.py checks the result against the true suffix array on small and mid-size texts.

This is input synthesis (plumbing): none of it is on the measured path, and the product never imports it.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

K = 32
ACGT = (65, 67, 71, 84)
EXPLICIT_CHUNK = 1 << 27   # members of explicit classes expanded at once (build_runs; tests lower it)
_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def _pbwt():
    global _lib
    if _lib is None:
        so = os.path.join(_HERE, "libpbwt.so")
        src = os.path.join(_HERE, "pbwt.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", src, "-o", so])
        _lib = ctypes.CDLL(so)
        for fn in (_lib.pbwt_suffix_ranks, _lib.pbwt_suffix_ranks16):
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_uint64, ctypes.c_uint32] + [ctypes.c_void_p] * 5
    return _lib


def make_pangenome(L, H, site_rate, seed, device, pad=10):
    """Random base sequence (2-bit codes), SNV sites at `site_rate` (none within K of either end), U-shaped
    allele frequencies, H haplotypes (haplotype 0 = the base sequence).  Everything stays on `device`."""
    assert 2 <= H <= 32767 and L > 4 * K   # (more than 255 haplotypes: 16-bit ranks, pbwt.c)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    base = torch.empty(L, dtype=torch.uint8, device=device)
    chunk = 1 << 28
    parts = []
    for a in range(0, L, chunk):
        b = min(L, a + chunk)
        base[a:b] = torch.randint(0, 4, (b - a,), generator=g, device=device, dtype=torch.uint8)
        m = torch.rand(b - a, generator=g, device=device) < site_rate
        parts.append(torch.nonzero(m).flatten() + a)
    sites = torch.cat(parts)
    sites = sites[(sites >= K) & (sites < L - K)]
    S = sites.numel()
    alt = ((base[sites].to(torch.int64) + torch.randint(1, 4, (S,), generator=g, device=device)) % 4).to(torch.uint8)
    u = torch.rand(S, generator=g, device=device, dtype=torch.float64)
    freq = torch.sin(u * (np.pi / 2)) ** 6
    freq = torch.where(torch.rand(S, generator=g, device=device, dtype=torch.float64) < 0.5, freq, 1 - freq).clamp(0.02, 0.98)
    G = torch.zeros((S, H), dtype=torch.uint8, device=device)
    for a in range(0, S, 1 << 22):
        b = min(S, a + (1 << 22))
        G[a:b, 1:] = (torch.rand((b - a, H - 1), generator=g, device=device) < freq[a:b, None].to(torch.float32)).to(torch.uint8)
    unit = L + pad
    return dict(L=L, H=H, pad=pad, unit=unit, n=H * unit + 1, seed=seed, base=base, sites=sites, alt=alt, G=G, n_sites=S)


def materialize_text(pg):
    """The text itself (uint8 on the pangenome's device), laid out like the reference's fixture:
    hap_0 + 'A'*pad + hap_1 + ... + 'A'*pad + 0x01."""
    dev = pg["base"].device
    lut = torch.tensor(ACGT, dtype=torch.uint8, device=dev)
    L, H, unit, n = pg["L"], pg["H"], pg["unit"], pg["n"]
    text = torch.empty(n, dtype=torch.uint8, device=dev)
    base_b = lut[pg["base"].long()] if L < (1 << 28) else torch.cat([lut[pg["base"][a:a + (1 << 28)].long()] for a in range(0, L, 1 << 28)])
    alt_b = lut[pg["alt"].long()]
    for h in range(H):
        hap = text[h * unit:h * unit + L]
        hap.copy_(base_b)
        carry = pg["G"][:, h] != 0
        hap[pg["sites"][carry]] = alt_b[carry]
        text[h * unit + L:(h + 1) * unit] = 65
    text[n - 1] = 1
    return text


class TextView:
    """text[pos] of the pangenome WITHOUT the text: symbol (h, p) is the base sequence's, or the alternative allele when p
    is a variant site haplotype h carries; pads are 'A', the last symbol is 0x01.  Holds the structure on the pangenome's
    device (L bytes of base sequence, the sites, the S x H allele matrix): what rbg_sample_reads_pangenome_dev samples from."""

    def __init__(self, pg):
        dev = pg["base"].device
        lut = torch.tensor(ACGT, dtype=torch.uint8, device=dev)
        L = pg["L"]
        self.base_b = lut[pg["base"].long()] if L < (1 << 28) else torch.cat([lut[pg["base"][a:a + (1 << 28)].long()] for a in range(0, L, 1 << 28)])
        self.alt_b = lut[pg["alt"].long()]
        self.sites = pg["sites"].contiguous()
        self.G = pg["G"].contiguous()
        self.L, self.H, self.unit, self.n, self.S = L, pg["H"], pg["unit"], pg["n"], pg["n_sites"]
        # site_dir[b] = # sites below b << shift (for the sampler: a read's first site without a search over all sites)
        self.site_dir_shift = 7
        edges = torch.arange((L >> self.site_dir_shift) + 2, device=dev, dtype=torch.int64) << self.site_dir_shift
        self.site_dir = torch.searchsorted(self.sites, edges).to(torch.int32).contiguous() if self.S else None

    def at(self, pos):
        """uint8 tensor of text[pos] (pos: int64 tensor of any shape, values in [0, n))"""
        h = torch.div(pos, self.unit, rounding_mode="floor")
        p = pos - h * self.unit
        c = torch.where(p < self.L, self.base_b[p.clamp(max=self.L - 1)], torch.full_like(p, 65, dtype=torch.uint8))
        if self.S:
            j = torch.searchsorted(self.sites, p).clamp(max=self.S - 1)
            hit = (self.sites[j] == p) & (p < self.L) & (h < self.H)
            carry = hit & (self.G[j, h.clamp(max=self.H - 1)] != 0)
            c = torch.where(carry, self.alt_b[j], c)
        return torch.where(pos == self.n - 1, torch.ones_like(c), c)


def _suffix_ranks(pg):
    """rank[s][h] (uint8, or int16 beyond 255 haplotypes; (S+1) x H, on device): order of the haplotypes by (alleles at
    sites s.., what follows the haplotype).  What follows haplotype h is haplotype h+1 from its start, i.e.
    rank[0][h+1]; the last haplotype is followed by the terminator (smallest).  Fixed point of the right-to-left pass
    (two passes when the allele vectors are distinct)."""
    dev = pg["base"].device
    S, H = pg["n_sites"], pg["H"]
    wide = H > 255
    rt = np.uint16 if wide else np.uint8
    fn = _pbwt().pbwt_suffix_ranks16 if wide else _pbwt().pbwt_suffix_ranks
    alt_smaller = pg["alt"] < pg["base"][pg["sites"]]
    small = np.empty((S, H), dtype=np.uint8)
    for a in range(0, S, 1 << 22):    # (chunked: the comparison makes temporaries of the chunk's size on the device)
        b = min(S, a + (1 << 22))
        small[a:b] = ((pg["G"][a:b] != 0) == alt_smaller[a:b, None]).to(torch.uint8).cpu().numpy()
    tail = np.arange(1, H + 1, dtype=rt)
    tail[H - 1] = 0
    rank = np.empty((S + 1, H), dtype=rt)
    first = np.empty(S + 1, dtype=rt)
    last = np.empty(S + 1, dtype=rt)
    for _ in range(H + 2):
        rc = fn(S, H, small.ctypes.data, tail.ctypes.data, rank.ctypes.data, first.ctypes.data, last.ctypes.data)
        assert rc == 0
        nxt = np.empty(H, dtype=rt)
        nxt[H - 1] = 0
        order = np.argsort(rank[0][1:], kind="stable")      # haplotypes 1..H-1 by their order from offset 0
        nxt[order] = np.arange(1, H, dtype=rt)               # haplotype h (0..H-2) is followed by haplotype h+1
        if (nxt == tail).all():
            return torch.from_numpy(rank.view(np.int16) if wide else rank).to(dev)   # (ranks stay below 2^15)
        tail = nxt
    raise RuntimeError("haplotype order did not converge")


def _pack_keys(ext):
    """ext: uint8 codes [m] -> int64 [m - K + 1]: the K = 32 codes from each position, 2 bits each, first code in
    the top bits (so unsigned integer order = lexicographic order)."""
    k = ext.to(torch.int64)
    w = 1
    while w < K:
        k = (k[:-w] << (2 * w)) | k[w:]
        w *= 2
    return k


_SIGN = -(1 << 63)


def build_runs(pg, log=None):
    """-> numpy dict like synth_pangenome.index_inputs: heads u8[R], lens u64[R], ssa u64[R], esa u64[R], n, r."""
    dev = pg["base"].device
    L, H, pad, unit, n, S = pg["L"], pg["H"], pg["pad"], pg["unit"], pg["n"], pg["n_sites"]
    sites, base, alt, G = pg["sites"], pg["base"], pg["alt"], pg["G"]
    say = log or (lambda *a: None)
    i64 = torch.int64
    lut = torch.tensor(ACGT, dtype=torch.uint8, device=dev)
    rank = _suffix_ranks(pg)                                   # [(S+1), H] uint8
    say("haplotype orders done")

    # ---- 32-mer keys of every offset of a generic haplotype unit (base + pad + start of the next haplotype)
    ext = torch.cat([base, torch.zeros(pad, dtype=torch.uint8, device=dev), base[:K]])
    basekey = _pack_keys(ext)[:unit]
    del ext

    # ---- window states: maximal offset intervals [a, b) of the regular region [0, unit-K] over which the set of
    # sites inside [p, p+K) is the same: sites j0 .. j0+c-1; the first site at or after p+K is j1 = j0 + c
    R_end = unit - K + 1
    a = torch.unique(torch.cat([torch.zeros(1, dtype=i64, device=dev), sites - K + 1, sites + 1]))
    a = a[a < R_end]
    b = torch.cat([a[1:], torch.tensor([R_end], dtype=i64, device=dev)])
    j0 = torch.searchsorted(sites, a)
    j1 = torch.searchsorted(sites, a + K)
    c = j1 - j0
    nst = a.numel()
    cmax = int(c.max().item()) if nst else 0
    assert cmax <= 22, "too many variant sites inside one 32-mer window for a 32-bit pattern key"
    # explicit states: the class at the state's FIRST offset has haplotype-dependent BWT characters
    # (offset 0: haplotype 0 is preceded by the terminator; a-1 a variant site)
    am1 = (a - 1).clamp(min=0)
    jprev = torch.searchsorted(sites, am1).clamp(max=max(S - 1, 0))
    st_explicit = (a == 0) | ((S > 0) & (sites[jprev] == am1) & (a > 0)) if S > 0 else (a == 0)

    # ---- groups of every state: haplotypes by (pattern over the window's sites, order from site j1 on)
    g_state, g_pat, g_cnt, g_first, g_last, g_col = [], [], [], [], [], []
    hdt = torch.int16 if H > 255 else torch.uint8
    order_rows = torch.empty((nst, H), dtype=hdt, device=dev)   # haplotypes of each state in class-then-rank order
    chunk = max(1, (1 << 25) // H)
    ar_h = torch.arange(H, device=dev)
    for s0 in range(0, nst, chunk):
        s1 = min(nst, s0 + chunk)
        pat = torch.zeros((s1 - s0, H), dtype=torch.int32, device=dev)
        for t in range(cmax):
            use = c[s0:s1] > t
            rows = (j0[s0:s1] + t).clamp(max=max(S - 1, 0))
            pat += (G[rows].to(torch.int32) << t) * use[:, None].to(torch.int32)
        key = pat.to(i64) * 65536 + rank[j1[s0:s1]].to(i64)
        skey, order = torch.sort(key, dim=1)
        order_rows[s0:s1] = order.to(hdt)
        spat = skey >> 16
        flag = torch.ones_like(spat, dtype=torch.bool)
        flag[:, 1:] = spat[:, 1:] != spat[:, :-1]
        rr, cc = torch.nonzero(flag, as_tuple=True)             # row-major: groups of a row are consecutive
        nxt = torch.empty_like(cc)
        nxt[:-1] = cc[1:]
        nxt[-1] = H
        last_of_row = torch.ones_like(rr, dtype=torch.bool)
        last_of_row[:-1] = rr[1:] != rr[:-1]
        nxt[last_of_row] = H
        g_state.append(rr + s0)
        g_pat.append(spat[rr, cc])
        g_cnt.append(nxt - cc)
        g_first.append(order[rr, cc])
        g_last.append(order[rr, nxt - 1])
        g_col.append(cc)
        del pat, key, skey, order, spat, flag
    g_state, g_pat, g_cnt = torch.cat(g_state), torch.cat(g_pat), torch.cat(g_cnt)
    g_first, g_last, g_col = torch.cat(g_first), torch.cat(g_last), torch.cat(g_col)
    ng = g_state.numel()
    say(f"{nst} window states, {ng} groups")

    # ---- classes of the regular region: (group, offset in the state's interval); key = base 32-mer with the
    # group's alternative alleles substituted
    span = (b - a)[g_state]
    cls_off = torch.cumsum(span, 0) - span
    ncls_r = int(span.sum().item())
    cls_g = torch.repeat_interleave(torch.arange(ng, device=dev), span)
    cls_p = torch.arange(ncls_r, device=dev) - cls_off[cls_g] + a[g_state[cls_g]]
    W = basekey[cls_p].clone()
    cg_state = g_state[cls_g]
    for t in range(cmax):
        sel = torch.nonzero((c[cg_state] > t) & (((g_pat[cls_g] >> t) & 1) != 0)).flatten()
        if sel.numel() == 0:
            continue
        j = j0[cg_state[sel]] + t
        s = sites[j]
        delta = (base[s] ^ alt[j]).to(i64)
        W[sel] ^= delta << (2 * (K - 1 - (s - cls_p[sel])))
    del cg_state
    # crossing region: offsets (unit-K, unit) of haplotypes 0..H-2 (their window runs into the next haplotype's
    # first symbols, where there is no site): one class per offset, ordered by what follows = rank[0][h+1]
    px = torch.arange(R_end, unit, device=dev)
    nx = px.numel()
    nxt_rank = rank[0][1:].to(i64)                              # of haplotype h+1, h = 0..H-2
    x_first = int(torch.argmin(nxt_rank).item())
    x_last = int(torch.argmax(nxt_rank).item())
    W = torch.cat([W, basekey[px]])
    ncls = ncls_r + nx
    say(f"{ncls} suffix classes for n = {n}")

    # ---- the order of the classes
    if dev.type == "cuda":
        torch.cuda.empty_cache()   # (the cached blocks of the stages above are of other sizes than the ones that follow)
    Ws, perm = torch.sort(W ^ _SIGN)
    del W
    if ncls > 1 and bool((Ws[1:] == Ws[:-1]).any().item()):
        raise RuntimeError("two suffix classes share their first 32 symbols: this builder needs a base sequence without 32-mer repeats")
    is_x = perm >= ncls_r
    gidx = torch.where(is_x, torch.zeros_like(perm), cls_g[perm.clamp(max=max(ncls_r - 1, 0))])
    p_s = torch.where(is_x, px[(perm - ncls_r).clamp(min=0)] if nx else torch.zeros_like(perm), cls_p[perm.clamp(max=max(ncls_r - 1, 0))])
    del cls_g, cls_p, perm
    cnt = torch.where(is_x, torch.full_like(gidx, H - 1), g_cnt[gidx])
    t_first = torch.where(is_x, torch.full_like(gidx, x_first), g_first[gidx]) * unit + p_s
    t_last = torch.where(is_x, torch.full_like(gidx, x_last), g_last[gidx]) * unit + p_s
    # BWT character of a class (uniform classes): the character before offset p of a generic haplotype
    pm1 = (p_s - 1).clamp(min=0)
    chr_ = torch.where(pm1 < L, lut[base[pm1.clamp(max=L - 1)].long()], torch.full_like(pm1, 65, dtype=torch.uint8))
    st_of = g_state[gidx]
    explicit = (~is_x) & st_explicit[st_of] & (p_s == a[st_of])
    ex_pos = torch.nonzero(explicit).flatten()

    # ---- explicit classes member by member -> their pieces (maximal equal-character stretches inside the class).  In
    # chunks of classes: their members number S * H (3e9 at the n = 3e11 pangenome: beyond one kernel launch, and 150 GB
    # of index tensors if expanded at once)
    ne = ex_pos.numel()
    e_g = gidx[ex_pos]
    e_cnt = g_cnt[e_g]
    per = max(1, EXPLICIT_CHUNK // H)
    ep_cls_l, ep_chr_l, ep_len_l, ep_tf_l, ep_tl_l = [], [], [], [], []
    for c0 in range(0, ne, per):
        c1 = min(ne, c0 + per)
        cnt_c = e_cnt[c0:c1]
        off_c = torch.cumsum(cnt_c, 0) - cnt_c
        m_tot = int(cnt_c.sum().item())
        if m_tot == 0:
            continue
        eg_c = e_g[c0:c1]
        m_cls = torch.repeat_interleave(torch.arange(c1 - c0, device=dev), cnt_c)
        m_k = torch.arange(m_tot, device=dev) - off_c[m_cls]
        m_state = g_state[eg_c][m_cls]
        m_h = order_rows[m_state, g_col[eg_c][m_cls] + m_k].to(i64)
        m_p = a[m_state]
        m_t = m_h * unit + m_p
        jp = torch.searchsorted(sites, (m_p - 1).clamp(min=0)).clamp(max=max(S - 1, 0))
        if S > 0:
            m_chr = torch.where(G[jp, m_h] != 0, lut[alt[jp].long()], lut[base[sites[jp]].long()])
        else:
            m_chr = torch.full((m_tot,), 65, dtype=torch.uint8, device=dev)
        at0 = m_p == 0
        m_chr = torch.where(at0, torch.where(m_h == 0, torch.full_like(m_chr, 1), torch.full_like(m_chr, 65)), m_chr)
        brk = torch.ones(m_tot, dtype=torch.bool, device=dev)
        if m_tot > 1:
            brk[1:] = (m_chr[1:] != m_chr[:-1]) | (m_cls[1:] != m_cls[:-1])
        pstart = torch.nonzero(brk).flatten()
        pend = torch.empty_like(pstart)
        pend[:-1] = pstart[1:] - 1
        pend[-1] = m_tot - 1
        ep_cls_l.append(m_cls[pstart] + c0)
        ep_chr_l.append(m_chr[pstart]); ep_len_l.append(pend - pstart + 1); ep_tf_l.append(m_t[pstart]); ep_tl_l.append(m_t[pend])
        del m_cls, m_k, m_state, m_h, m_p, m_t, jp, m_chr, at0, brk, pstart, pend
    if ep_cls_l:
        ep_cls = torch.cat(ep_cls_l)
        ep_chr, ep_len, ep_tf, ep_tl = torch.cat(ep_chr_l), torch.cat(ep_len_l), torch.cat(ep_tf_l), torch.cat(ep_tl_l)
    else:
        ep_cls = torch.zeros(0, dtype=i64, device=dev)
        ep_chr = torch.zeros(0, dtype=torch.uint8, device=dev)
        ep_len = ep_tf = ep_tl = torch.zeros(0, dtype=i64, device=dev)
    del ep_cls_l, ep_chr_l, ep_len_l, ep_tf_l, ep_tl_l
    ep_n = torch.bincount(ep_cls, minlength=ne)
    ep_first = torch.cumsum(ep_n, 0) - ep_n
    ep_k = torch.arange(ep_cls.numel(), device=dev) - ep_first[ep_cls]

    if dev.type == "cuda":
        torch.cuda.empty_cache()
    # ---- all pieces in class order
    pc = torch.ones(ncls, dtype=i64, device=dev)
    pc[ex_pos] = ep_n
    poff = torch.cumsum(pc, 0) - pc
    npieces = int(pc.sum().item())
    P_chr = torch.empty(npieces, dtype=torch.uint8, device=dev)
    P_len = torch.empty(npieces, dtype=i64, device=dev)
    P_tf = torch.empty(npieces, dtype=i64, device=dev)
    P_tl = torch.empty(npieces, dtype=i64, device=dev)
    uni = torch.nonzero(~explicit).flatten()
    P_chr[poff[uni]] = chr_[uni]
    P_len[poff[uni]] = cnt[uni]
    P_tf[poff[uni]] = t_first[uni]
    P_tl[poff[uni]] = t_last[uni]
    dst = poff[ex_pos][ep_cls] + ep_k
    P_chr[dst], P_len[dst], P_tf[dst], P_tl[dst] = ep_chr, ep_len, ep_tf, ep_tl
    del chr_, cnt, t_first, t_last, uni, dst

    # ---- the last haplotype's final K suffixes (they contain the terminator within K symbols) and the terminator
    # suffix itself: X$ sorts before every suffix whose first |X| symbols are >= X
    tail_codes = torch.cat([base[L - K:], torch.zeros(pad, dtype=torch.uint8, device=dev)]).cpu().numpy()   # offsets [L-K, unit)
    tails = []
    for d in range(K):
        X = tail_codes[len(tail_codes) - d:] if d else tail_codes[:0]
        keyv = 0
        for x in X.tolist():
            keyv = (keyv << 2) | x
        keyv <<= 2 * (K - d)
        signed = (keyv ^ (1 << 63)) - (1 << 64) if (keyv ^ (1 << 63)) >= (1 << 63) else (keyv ^ (1 << 63))
        lb = int(torch.searchsorted(Ws, torch.tensor([signed], dtype=i64, device=dev)).item())
        pos_in_unit = unit - d
        q = pos_in_unit - 1                                      # the character before the suffix, inside the last unit
        ch = 65 if q >= L else ACGT[int(base[q].item())]
        tails.append((lb, tuple(X.tolist()) + (-1,), (H - 1) * unit + pos_in_unit, ch))
    tails.sort(key=lambda t: (t[0], t[1]))   # among themselves: X$ against X'$, the terminator (-1) smallest
    ins_at = torch.tensor([int(poff[t[0]].item()) if t[0] < ncls else npieces for t in tails], dtype=i64)
    segs_chr, segs_len, segs_tf, segs_tl = [], [], [], []
    prev = 0
    for (lb, _x, tpos, ch), at in zip(tails, ins_at.tolist()):
        if at > prev:
            segs_chr.append(P_chr[prev:at]); segs_len.append(P_len[prev:at]); segs_tf.append(P_tf[prev:at]); segs_tl.append(P_tl[prev:at])
            prev = at
        segs_chr.append(torch.tensor([ch], dtype=torch.uint8, device=dev))
        segs_len.append(torch.ones(1, dtype=i64, device=dev))
        segs_tf.append(torch.tensor([tpos], dtype=i64, device=dev))
        segs_tl.append(torch.tensor([tpos], dtype=i64, device=dev))
    if prev < npieces:
        segs_chr.append(P_chr[prev:]); segs_len.append(P_len[prev:]); segs_tf.append(P_tf[prev:]); segs_tl.append(P_tl[prev:])
    P_chr, P_len, P_tf, P_tl = torch.cat(segs_chr), torch.cat(segs_len), torch.cat(segs_tf), torch.cat(segs_tl)
    del segs_chr, segs_len, segs_tf, segs_tl

    # ---- merge equal neighbours into BWT runs
    brk = torch.ones(P_chr.numel(), dtype=torch.bool, device=dev)
    brk[1:] = P_chr[1:] != P_chr[:-1]
    starts = torch.nonzero(brk).flatten()
    ends = torch.empty_like(starts)
    ends[:-1] = starts[1:] - 1
    ends[-1] = P_chr.numel() - 1
    csum = torch.cumsum(P_len, 0)
    lens = csum[ends] - csum[starts] + P_len[starts]
    assert int(csum[-1].item()) == n, (int(csum[-1].item()), n)
    # (views, not .astype copies: 8 bytes per run each, 8 GB at r = 1e9, on a host whose container has a memory limit)
    out = dict(heads=P_chr[starts].cpu().numpy(), lens=lens.cpu().numpy().view(np.uint64),
               ssa=P_tf[starts].cpu().numpy().view(np.uint64), esa=P_tl[ends].cpu().numpy().view(np.uint64),
               n=n, r=int(starts.numel()))
    return out
