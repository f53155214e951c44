"""Synthetic pangenome r-index inputs for bench.py and the at-scale parity checks.

There is no real data and no reference index builder in this environment (SURVEY.md 7 "Hard
parts"), so the benchmark index is synthesised: a random base sequence, H haplotypes that differ
from it by SNVs, laid out exactly like the reference's fixture text
(hap_0 + 'A'*pad + hap_1 + ... + 0x01), then a true suffix array of that text by
prefix doubling with torch sorts (on the GPU at bench scale, on the CPU for tests), from which the
run-length BWT and the run-boundary SA samples (the contents of the reference's .bwt/.ssa/.esa
build inputs, rb_build.cpp:83-93) are read off.  Always labelled "synthetic"; never "chr22".

This is input synthesis (plumbing): none of it is on the measured path.
"""
import numpy as np
import torch

ACGT = (65, 67, 71, 84)


def make_text(L, H, site_rate, seed, device, pad=10):
    """-> (text uint8 [n] on device, info dict).  n = H*(L+pad) + 1."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    lut = torch.tensor(ACGT, dtype=torch.uint8)
    base_code = torch.randint(0, 4, (L,), generator=g, dtype=torch.int64)
    n_sites = max(1, int(L * site_rate))
    sites = torch.randperm(L, generator=g)[:n_sites].sort().values
    shift = torch.randint(1, 4, (n_sites,), generator=g, dtype=torch.int64)
    alt_code = (base_code[sites] + shift) % 4
    # allele frequencies: many rare, some common (U-shaped)
    u = torch.rand(n_sites, generator=g, dtype=torch.float64)
    freq = torch.sin(u * (np.pi / 2)) ** 6
    freq = torch.where(torch.rand(n_sites, generator=g, dtype=torch.float64) < 0.5, freq, 1 - freq).clamp(0.02, 0.98)
    unit = L + pad
    n = H * unit + 1
    text = torch.empty(n, dtype=torch.uint8, device=device)
    base_d = lut[base_code].to(device)
    sites_d = sites.to(device)
    alt_d = lut[alt_code].to(device)
    freq_d = freq.to(device)
    gd = torch.Generator(device=device)
    gd.manual_seed(seed + 1)
    for h in range(H):
        hap = base_d.clone()
        if h > 0:
            carry = torch.rand(n_sites, generator=gd, device=device, dtype=torch.float64) < freq_d
            hap[sites_d[carry]] = alt_d[carry]
        text[h * unit:h * unit + L] = hap
        text[h * unit + L:(h + 1) * unit] = 65
    text[n - 1] = 1
    return text, dict(L=L, H=H, pad=pad, unit=unit, n=n, n_sites=n_sites, seed=seed)


def suffix_array(text):
    """Prefix doubling with torch.sort; text uint8 [n] (terminator = unique smallest byte, last).
    Returns int64 SA on text.device.  O(n log n) memory-heavy but simple and exact."""
    n = text.numel()
    dev = text.device
    rank = text.to(torch.int64)
    k = 1
    # pack the first characters so the first rounds are skipped: 8 chars x 8 bits would overflow the
    # key product below, so start from 4-char ranks (values < 2^32) and let doubling do the rest
    if n > 4:
        r = rank.clone()
        for d in (1, 2, 3):
            nxt = torch.zeros_like(rank)
            nxt[: n - d] = rank[d:]
            r = r * 256 + nxt
        # compress to dense ranks
        vals, sa = torch.sort(r)
        flags = torch.ones(n, dtype=torch.int64, device=dev)
        flags[1:] = (vals[1:] != vals[:-1]).to(torch.int64)
        dense = torch.cumsum(flags, 0) - 1
        rank = torch.empty_like(dense)
        rank[sa] = dense
        del r, vals, flags, nxt
        k = 4
        if int(dense[-1]) == n - 1:
            return sa
        del dense, sa
    while True:
        r2 = torch.zeros(n, dtype=torch.int64, device=dev)
        r2[: n - k] = rank[k:] + 1
        key = rank * (n + 2) + r2
        del r2
        vals, sa = torch.sort(key)
        del key
        flags = torch.ones(n, dtype=torch.int64, device=dev)
        flags[1:] = (vals[1:] != vals[:-1]).to(torch.int64)
        del vals
        dense = torch.cumsum(flags, 0) - 1
        del flags
        rank[sa] = dense
        done = int(dense[-1]) == n - 1
        del dense
        if done:
            return sa
        k *= 2


def index_inputs(text, sa):
    """-> numpy dict: heads u8[R], lens u64[R], ssa u64[R], esa u64[R] (raw SA values at the first
    / last position of each BWT run: the 'y' of the reference's .ssa/.esa pairs), n, r."""
    n = text.numel()
    prev = sa - 1
    prev[prev < 0] = n - 1
    bwt = text[prev]
    del prev
    brk = torch.ones(n, dtype=torch.bool, device=text.device)
    brk[1:] = bwt[1:] != bwt[:-1]
    starts = torch.nonzero(brk).flatten()
    del brk
    R = starts.numel()
    ends = torch.empty_like(starts)
    ends[:-1] = starts[1:] - 1
    ends[-1] = n - 1
    out = dict(
        heads=bwt[starts].cpu().numpy().astype(np.uint8),
        lens=(ends - starts + 1).cpu().numpy().astype(np.uint64),
        ssa=sa[starts].cpu().numpy().astype(np.uint64),
        esa=sa[ends].cpu().numpy().astype(np.uint64),
        n=n, r=R,
    )
    return out


def sample_reads(text, info, n_reads, m, seed, sub_rate=0.1):
    """reads sampled inside haplotypes (never crossing the pads), `sub_rate` of them with
    one substitution.  -> uint8 [n_reads, m] on text.device (row-major = concatenated reads)."""
    dev = text.device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    H, L, unit = info["H"], info["L"], info["unit"]
    hap = torch.randint(0, H, (n_reads,), generator=g, device=dev)
    pos = torch.randint(0, L - m + 1, (n_reads,), generator=g, device=dev)
    start = hap * unit + pos
    reads = torch.empty((n_reads, m), dtype=torch.uint8, device=dev)
    chunk = max(1, (1 << 27) // m)
    ar = torch.arange(m, device=dev)
    for a in range(0, n_reads, chunk):
        b = min(n_reads, a + chunk)
        reads[a:b] = text[(start[a:b, None] + ar[None, :])]
    mut = torch.rand(n_reads, generator=g, device=dev) < sub_rate
    idx = torch.nonzero(mut).flatten()
    if idx.numel():
        p = torch.randint(0, m, (idx.numel(),), generator=g, device=dev)
        old = reads[idx, p].to(torch.int64)
        code = (old == 67).to(torch.int64) + 2 * (old == 71).to(torch.int64) + 3 * (old == 84).to(torch.int64)
        sh = torch.randint(1, 4, (idx.numel(),), generator=g, device=dev)
        lut = torch.tensor(ACGT, dtype=torch.uint8, device=dev)
        reads[idx, p] = lut[(code + sh) % 4]
    return reads, start
