"""ctypes binding of oracle/liborc.so (TEST INFRASTRUCTURE: the CPU restatement of the reference).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "liborc.so")

U64 = C.c_uint64
P64 = C.POINTER(C.c_uint64)
VP = C.c_void_p

NONE, SA, MA, DL, FT = 0, 1, 2, 4, 8
MAXU = 2**64 - 1

_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_SO)
        L.orc_load.restype = VP
        L.orc_load.argtypes = [C.c_char_p, C.c_int]
        L.orc_build_from_runs.restype = VP
        L.orc_build_from_runs.argtypes = [VP, VP, U64, U64, VP, VP]
        L.orc_set_markers.argtypes = [VP, VP, VP, U64, VP, VP]
        L.orc_set_reference_shaped.argtypes = [VP, C.c_int]
        L.orc_set_reference_shaped.restype = None
        L.orc_set_docs.argtypes = [VP, C.c_char_p, VP, U64]
        L.orc_free.argtypes = [VP]
        for f in ("orc_n", "orc_r", "orc_marker_nruns", "orc_marker_nvals", "orc_last_run_sample"):
            getattr(L, f).restype = U64
            getattr(L, f).argtypes = [VP]
        L.orc_has_tsa.argtypes = [VP]
        L.orc_has_markers.argtypes = [VP]
        L.orc_get_f.argtypes = [VP, VP]
        L.orc_get_runs.argtypes = [VP, VP, VP]
        L.orc_get_tsa.argtypes = [VP, VP, VP, VP]
        L.orc_get_markers.argtypes = [VP, VP, VP, VP, VP]
        L.orc_rank.restype = U64
        L.orc_rank.argtypes = [VP, U64, C.c_uint8]
        L.orc_select.restype = U64
        L.orc_select.argtypes = [VP, U64, C.c_uint8]
        L.orc_access.restype = C.c_uint8
        L.orc_access.argtypes = [VP, U64]
        L.orc_run_of_position.restype = U64
        L.orc_run_of_position.argtypes = [VP, U64]
        L.orc_phi.restype = U64
        L.orc_phi.argtypes = [VP, U64]
        L.orc_LF.argtypes = [VP, U64, U64, C.c_uint8, P64, P64]
        L.orc_find_range.argtypes = [VP, C.c_char_p, U64, P64, P64]
        L.orc_find_range_w_toehold.argtypes = [VP, C.c_char_p, U64, P64, P64, P64]
        L.orc_locs_at.restype = U64
        L.orc_locs_at.argtypes = [VP, U64, U64, U64, U64, VP]
        L.orc_markers_at.restype = U64
        L.orc_markers_at.argtypes = [VP, U64, U64, VP]
        L.orc_find_range_w_markers.restype = U64
        L.orc_find_range_w_markers.argtypes = [VP, C.c_char_p, U64, U64, U64, P64, P64, VP, U64]
        L.orc_greedy_locate.restype = U64
        L.orc_greedy_locate.argtypes = [VP, C.c_char_p, U64, U64, U64, VP, U64, P64, P64, P64, P64, P64]
        L.orc_markers_greedy_seeding_ftab.restype = U64
        L.orc_markers_greedy_seeding_ftab.argtypes = [VP, C.c_char_p, U64, U64, U64, U64, VP, U64, VP, U64, P64]
        L.orc_resolve_offset.restype = C.c_char_p
        L.orc_resolve_offset.argtypes = [VP, U64, P64]
        L.orc_find_range_batch.argtypes = [VP, VP, VP, U64, VP, VP, C.c_int]
        L.orc_find_range_w_toehold_batch.argtypes = [VP, VP, VP, U64, VP, VP, VP, C.c_int]
        L.orc_locs_at_batch.argtypes = [VP, VP, VP, VP, U64, U64, VP, VP, C.c_int]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(VP)


def pack_reads(reads):
    """list of bytes -> (uint8 concat, uint64 offsets[N+1])"""
    off = np.zeros(len(reads) + 1, dtype=np.uint64)
    if reads:
        off[1:] = np.cumsum([len(r) for r in reads], dtype=np.uint64)
    seqs = np.frombuffer(b"".join(reads), dtype=np.uint8).copy() if reads else np.zeros(0, np.uint8)
    return seqs, off


class Oracle:
    """Mirrors the rbwt::RowBowt query surface (rowbowt.hpp) over the C restatement."""

    def __init__(self, handle):
        if not handle:
            raise RuntimeError("oracle: load/build failed")
        self.h = handle
        self.L = lib()

    @classmethod
    def load(cls, prefix, flags=NONE):
        return cls(lib().orc_load(prefix.encode(), flags))

    @classmethod
    def from_runs(cls, heads, lens, ssa=None, esa=None, B=2):
        heads = np.ascontiguousarray(heads, dtype=np.uint8)
        lens = np.ascontiguousarray(lens, dtype=np.uint64)
        if ssa is not None:
            ssa = np.ascontiguousarray(ssa, dtype=np.uint64)
            esa = np.ascontiguousarray(esa, dtype=np.uint64)
        h = lib().orc_build_from_runs(_p(heads), _p(lens), len(heads), B,
                                      _p(ssa) if ssa is not None else None,
                                      _p(esa) if esa is not None else None)
        return cls(h)

    def set_reference_shaped(self, on=True):
        """rank / select / access through Elias-Fano vectors and a Huffman-shaped wavelet tree instead of the decoded arrays
        (SURVEY 8d's optional CPU mode): same answers, the reference's memory behaviour"""
        self.L.orc_set_reference_shaped(self.h, 1 if on else 0)

    def set_markers(self, run_start, run_end, mk_off, mk_vals):
        a = [np.ascontiguousarray(v, dtype=np.uint64) for v in (run_start, run_end, mk_off, mk_vals)]
        self.L.orc_set_markers(self.h, _p(a[0]), _p(a[1]), len(a[0]), _p(a[2]), _p(a[3]))

    def set_docs(self, names, starts):
        joined = b"\0".join(n.encode() for n in names) + b"\0"
        s = np.ascontiguousarray(starts, dtype=np.uint64)
        self.L.orc_set_docs(self.h, joined, _p(s), len(names))

    def close(self):
        if self.h:
            self.L.orc_free(self.h)
            self.h = None

    # --- info / decoded views
    @property
    def n(self):
        return self.L.orc_n(self.h)

    @property
    def r(self):
        return self.L.orc_r(self.h)

    def f(self):
        out = np.zeros(256, dtype=np.uint64)
        self.L.orc_get_f(self.h, _p(out))
        return out

    def runs(self):
        heads = np.zeros(self.r, dtype=np.uint8)
        lens = np.zeros(self.r, dtype=np.uint64)
        self.L.orc_get_runs(self.h, _p(heads), _p(lens))
        return heads, lens

    def tsa(self):
        a = [np.zeros(self.r, dtype=np.uint64) for _ in range(3)]
        self.L.orc_get_tsa(self.h, _p(a[0]), _p(a[1]), _p(a[2]))
        return a  # pred_pos, samples_last, pred_to_run

    def markers(self):
        nr, nv = self.L.orc_marker_nruns(self.h), self.L.orc_marker_nvals(self.h)
        s, e, o, v = (np.zeros(nr, np.uint64), np.zeros(nr, np.uint64), np.zeros(nr + 1, np.uint64), np.zeros(nv, np.uint64))
        self.L.orc_get_markers(self.h, _p(s), _p(e), _p(o), _p(v))
        return s, e, o, v

    # --- primitives
    def rank(self, i, c):
        return self.L.orc_rank(self.h, i, c)

    def select(self, i, c):
        return self.L.orc_select(self.h, i, c)

    def access(self, i):
        return self.L.orc_access(self.h, i)

    def run_of_position(self, i):
        return self.L.orc_run_of_position(self.h, i)

    def phi(self, i):
        return self.L.orc_phi(self.h, i)

    def last_run_sample(self):
        return self.L.orc_last_run_sample(self.h)

    def LF(self, lo, hi, c):
        a, b = U64(), U64()
        self.L.orc_LF(self.h, lo, hi, c, a, b)
        return a.value, b.value

    # --- RowBowt surface
    def find_range(self, q):
        a, b = U64(), U64()
        self.L.orc_find_range(self.h, q, len(q), a, b)
        return a.value, b.value

    def count(self, q):
        lo, hi = self.find_range(q)
        return hi - lo + 1 if hi >= lo else 0

    def find_range_w_toehold(self, q):
        a, b, k = U64(), U64(), U64()
        self.L.orc_find_range_w_toehold(self.h, q, len(q), a, b, k)
        return a.value, b.value, k.value

    def locs_at(self, lo, hi, k, max_hits=MAXU):
        occ = hi - lo + 1 if hi >= lo else 0
        occ = min(occ, max_hits)
        out = np.zeros(max(occ, 1), dtype=np.uint64)
        n = self.L.orc_locs_at(self.h, lo, hi, k, max_hits, _p(out))
        return out[:n].tolist()

    def markers_at(self, lo, hi):
        n = self.L.orc_markers_at(self.h, lo, hi, None)
        out = np.zeros(max(n, 1), dtype=np.uint64)
        self.L.orc_markers_at(self.h, lo, hi, _p(out))
        return out[:n].tolist()

    def find_range_w_markers(self, q, wsize, max_range):
        a, b = U64(), U64()
        cap = 4096
        out = np.zeros(cap, dtype=np.uint64)
        n = self.L.orc_find_range_w_markers(self.h, q, len(q), wsize, max_range & MAXU, a, b, _p(out), cap)
        return (a.value, b.value), out[:min(n, cap)].tolist()

    def greedy_locate(self, q, min_length, max_hits=MAXU):
        s = [U64() for _ in range(5)]
        dummy = np.zeros(1, dtype=np.uint64)
        n = self.L.orc_greedy_locate(self.h, q, len(q), min_length, max_hits, _p(dummy), 0, *s)  # count
        out = np.zeros(max(n, 1), dtype=np.uint64)
        n = self.L.orc_greedy_locate(self.h, q, len(q), min_length, max_hits, _p(out), max(n, 1), *s)
        return out[:n].tolist(), tuple(v.value for v in s)

    def markers_greedy_seeding(self, q, wsize, max_range, ftab_k=0):
        """-> list of (lo, hi, q_first, q_end_exclusive, [markers]) in callback order (rowbowt.hpp:406-482);
        ftab_k > 0: with the ftab of that k-mer size loaded"""
        nmk = U64(0)
        ns = lib().orc_markers_greedy_seeding_ftab(self.h, q, len(q), wsize, max_range, ftab_k, None, 0, None, 0, C.byref(nmk))
        seeds = np.zeros(6 * max(ns, 1), dtype=np.uint64)
        mk = np.zeros(max(nmk.value, 1), dtype=np.uint64)
        lib().orc_markers_greedy_seeding_ftab(self.h, q, len(q), wsize, max_range, ftab_k, _p(seeds), ns, _p(mk), nmk.value, C.byref(nmk))
        out = []
        for s_ in range(ns):
            lo, hi, qs, qe, b, e = (int(v) for v in seeds[6 * s_:6 * s_ + 6])
            out.append((lo, hi, qs, qe, mk[b:e].tolist()))
        return out

    def resolve_offset(self, i):
        off = U64()
        name = self.L.orc_resolve_offset(self.h, i, off)
        return (name.decode() if name else None), off.value

    # --- batched
    def find_range_batch(self, seqs, off, nthreads=1):
        N = len(off) - 1
        lo = np.zeros(N, np.uint64)
        hi = np.zeros(N, np.uint64)
        self.L.orc_find_range_batch(self.h, _p(seqs), _p(off), N, _p(lo), _p(hi), nthreads)
        return lo, hi

    def find_range_w_toehold_batch(self, seqs, off, nthreads=1):
        N = len(off) - 1
        lo, hi, k = np.zeros(N, np.uint64), np.zeros(N, np.uint64), np.zeros(N, np.uint64)
        self.L.orc_find_range_w_toehold_batch(self.h, _p(seqs), _p(off), N, _p(lo), _p(hi), _p(k), nthreads)
        return lo, hi, k

    def locs_at_batch(self, lo, hi, k, max_hits=MAXU, nthreads=1):
        N = len(lo)
        occ = np.where(hi >= lo, hi - lo + np.uint64(1), np.uint64(0)).astype(np.uint64)
        if max_hits < MAXU:
            occ = np.minimum(occ, np.uint64(max_hits))
        loc_off = np.zeros(N + 1, np.uint64)
        loc_off[1:] = np.cumsum(occ, dtype=np.uint64)
        locs = np.zeros(max(int(loc_off[-1]), 1), np.uint64)
        self.L.orc_locs_at_batch(self.h, _p(lo), _p(hi), _p(k), N, max_hits, _p(loc_off), _p(locs), nthreads)
        return loc_off, locs[: int(loc_off[-1])]


def read_fastx(path):
    """Minimal FASTA/FASTQ reader (kseq semantics: name = header up to first whitespace)."""
    names, seqs = [], []
    with open(path, "rb") as f:
        lines = [l.rstrip(b"\r\n") for l in f]
    i = 0
    while i < len(lines):
        l = lines[i]
        if l.startswith(b">"):
            names.append(l[1:].split()[0] if l[1:].split() else b"")
            i += 1
            s = b""
            while i < len(lines) and not lines[i].startswith((b">", b"@")):
                s += lines[i]
                i += 1
            seqs.append(s)
        elif l.startswith(b"@"):
            names.append(l[1:].split()[0] if l[1:].split() else b"")
            seqs.append(lines[i + 1])
            i += 4
        else:
            i += 1
    return names, seqs
