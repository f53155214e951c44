"""Small synthetic pangenomes for parity tests (built with the naive suffix sorter in naive.py).

Text layout follows the reference's fixture: hap_0 + 'A'*pad + hap_1 + 'A'*pad ... + 0x01.
"""
import numpy as np

import naive

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def make_text(L, H, n_sites, seed, pad=10):
    rng = np.random.default_rng(seed)
    base = ACGT[rng.integers(0, 4, L)]
    sites = np.sort(rng.choice(L, size=min(n_sites, L), replace=False))
    alts = np.array([rng.choice([c for c in ACGT if c != base[s]]) for s in sites], dtype=np.uint8)
    freq = rng.beta(0.3, 0.3, len(sites))
    parts, carriers = [], []
    for h in range(H):
        hap = base.copy()
        carry = rng.random(len(sites)) < freq if h else np.zeros(len(sites), bool)
        hap[sites[carry]] = alts[carry]
        carriers.append(carry)
        parts += [hap, np.full(pad, ord("A"), np.uint8)]
    text = np.concatenate(parts + [np.array([1], np.uint8)])
    return text, base, sites, alts, np.array(carriers)


class SynthIndex:
    def __init__(self, L=3000, H=6, n_sites=40, seed=0, pad=10):
        self.text, self.base, self.sites, self.alts, self.carriers = make_text(L, H, n_sites, seed, pad)
        self.L, self.H, self.pad = L, H, pad
        self.fm = naive.NaiveFM(self.text)
        self.n = len(self.text)
        bwt = naive.bwt_from_sa(self.text, self.fm.sa)
        self.heads, self.lens, self.brk = naive.rle(bwt)
        self.ssa, self.esa = naive.run_samples(self.fm.sa, self.brk, self.n)
        self.doc_names = [f"hap{h}" for h in range(H)]
        self.doc_starts = [h * (L + pad) for h in range(H)]

    def markers(self, wsize=10):
        """Synthetic marker array shaped like small.fa.mab: for every variant site and allele, the SA
        indexes whose suffix starts within `wsize` bases before the site (on a haplotype carrying
        that allele) form runs; each run lists MarkerT = pos | allele << 60."""
        n, unit = self.n, self.L + self.pad
        tag = {}
        for h in range(self.H):
            for si, s in enumerate(self.sites):
                allele = int(self.carriers[h][si])
                for d in range(1, wsize + 1):
                    p = s - d + 1
                    if p < 0:
                        continue
                    tag.setdefault(h * unit + p, []).append(int(s) | (allele << 60))
        isa = np.empty(n, dtype=np.int64)
        isa[self.fm.sa] = np.arange(n)
        by_sa = sorted((int(isa[t]), tuple(sorted(set(v)))) for t, v in tag.items())
        run_start, run_end, mk_off, mk_vals = [], [], [0], []
        for idx, vals in by_sa:
            if run_start and run_end[-1] == idx - 1 and tuple(mk_vals[mk_off[-2]:mk_off[-1]]) == vals:
                run_end[-1] = idx
            else:
                run_start.append(idx)
                run_end.append(idx)
                mk_vals += list(vals)
                mk_off.append(len(mk_vals))
        return (np.array(run_start, np.uint64), np.array(run_end, np.uint64),
                np.array(mk_off, np.uint64), np.array(mk_vals, np.uint64))

    def sample_reads(self, n_reads, m, seed, sub_rate=0.1, ragged=False):
        rng = np.random.default_rng(seed)
        unit = self.L + self.pad
        reads = []
        for _ in range(n_reads):
            mm = int(rng.integers(1, m + 1)) if ragged else m
            h = int(rng.integers(self.H))
            s = h * unit + int(rng.integers(0, self.L - mm + 1))
            r = bytearray(self.text[s:s + mm].tobytes())
            if rng.random() < sub_rate:
                p = int(rng.integers(mm))
                r[p] = int(rng.choice([c for c in b"ACGT" if c != r[p]]))
            reads.append(bytes(r))
        return reads
