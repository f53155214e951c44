"""ctypes binding of rowbowt_amd/librbg.so (the C-ABI of include/rbg.h).

Names mirror the reference's operator API for this path: rbwt::load_rowbowt (rowbowt_io.hpp:176),
rbwt::LoadRbwtFlag (:146-152) and rbwt::RowBowt<> methods find_range / find_range_w_toehold /
locs_at / count / markers_at / find_range_w_markers / resolve_offset (rowbowt.hpp), batched.
The library must be present: a missing or unloadable librbg.so is an error, never a fallback.
"""
import ctypes as C
import enum
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librbg.so")

U64 = C.c_uint64
VP = C.c_void_p
MAXU = 2**64 - 1
DEVICE_NONE = -1

OPT_BLOCK_THREADS, OPT_RANK_BUCKET_SHIFT, OPT_PHI_BUCKET_SHIFT, OPT_POS_BYTES, OPT_KMER_STEPS, OPT_HBM_BUDGET_MB, OPT_FTAB_K, OPT_PACKED_READS, OPT_DEEP_BUCKET_SHIFT, OPT_DENSE_OVERFLOW, OPT_RANK_LAYOUT = range(1, 12)
OPT_RUN_DEPTHS, OPT_RUN_PHI, OPT_RUN_REC, OPT_RUN_REC_DEPTHS = 14, 16, 17, 18   # (12, 13, 15 were TREE_TOP_KB, SLOT_BYTES, RUN_FMT: retired with ABI 3)
ABI_VERSION = 3          # include/rbg.h RBG_ABI_VERSION this binding is written against
MAX_KMER_DEPTH = 8       # RBG_OPT_KMER_STEPS / the depth arrays of Info and LayoutInfo
LAYOUT_AUTO, LAYOUT_SLOTS, LAYOUT_RUNS, LAYOUT_PREFER_SLOTS = 0, 1, 2, 3
(ARR_RUN_HEADS, ARR_RUN_START, ARR_SAMPLES_LAST, ARR_PRED_POS, ARR_PHI_BASE,
 ARR_MARKER_START, ARR_MARKER_END, ARR_MARKER_OFF, ARR_MARKER_VALS) = range(9)


class LoadRbwtFlag(enum.IntFlag):
    """rbwt::LoadRbwtFlag, rowbowt_io.hpp:146-152"""
    NONE = 0
    SA = 1
    MA = 2
    DL = 4
    FT = 8


class RbgError(RuntimeError):
    def __init__(self, code, what):
        self.code = code
        super().__init__(f"{what}: {lib().rbg_strerror(code).decode()} ({code})")


class Info(C.Structure):
    _fields_ = [("n", U64), ("r", U64), ("sigma", C.c_uint32), ("pos_bytes", C.c_uint32), ("device", C.c_int32),
                ("has_tsa", C.c_uint32), ("has_markers", C.c_uint32), ("has_docs", C.c_uint32),
                ("hbm_bytes", U64), ("marker_runs", U64), ("marker_vals", U64),
                ("rank_bucket_shift", C.c_uint32), ("phi_bucket_shift", C.c_uint32), ("slot_bytes", C.c_uint32),
                ("rank_slots", U64), ("rank_slots_overflow", U64), ("phi_slots", U64), ("phi_slots_overflow", U64),
                ("kmer_steps", U64), ("kmer_symbols", U64), ("pair_runs", U64), ("triple_runs", U64), ("quad_runs", U64), ("ftab_k", U64), ("quint_runs", U64),
                ("kmer_steps_requested", U64), ("hbm_free_at_load", U64), ("hbm_budget", U64), ("rank_layout", U64), ("replicas", U64),
                ("depth_runs", U64 * 8)]


class LayoutInfo(C.Structure):
    """rbg_layout_info_t"""
    _fields_ = [("run_fmt", C.c_uint32), ("depths_composed", C.c_uint32), ("depth_mask_asked", C.c_uint32), ("depth_mask_kept", C.c_uint32),
                ("depths_dropped_budget", C.c_uint32), ("rank_directories", C.c_uint32),
                ("phi_directory", C.c_uint32), ("fill_shift", C.c_uint32),
                ("entries", U64 * 8), ("fillers", U64 * 8), ("dir_bytes", U64 * 8),
                ("phi_entries", U64), ("phi_fillers", U64), ("phi_dir_bytes", U64), ("phi_dir_shift", U64), ("phi_slots", U64), ("phi_slot_bytes", U64), ("rec_bytes", U64 * 8), ("rec_overflow", U64 * 8), ("budget_raised", U64)]


# every symbol include/rbg.h declares: (name, restype, argtypes)
_PROTOS = [
    ("rbg_abi_version", C.c_int, []),
    ("rbg_strerror", C.c_char_p, [C.c_int]),
    ("rbg_load", C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(VP)]),
    ("rbg_build_from_runs", C.c_int, [VP, VP, U64, VP, VP, C.c_int, C.POINTER(VP)]),
    ("rbg_build_from_files", C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(VP)]),
    ("rbg_convert_index", C.c_int, [C.c_char_p, C.c_int, C.c_char_p]),
    ("rbg_convert_raw", C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]),
    ("rbg_load_cache", C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(VP)]),
    ("rbg_doc_table", C.c_int, [VP, C.POINTER(U64), C.POINTER(VP), C.POINTER(VP), C.POINTER(U64)]),
    ("rbg_align_text", C.c_int, [VP, VP, VP, VP, U64, U64, C.c_uint32, VP, VP, VP, C.POINTER(VP), C.POINTER(U64)]),
    ("rbg_wait_text", C.c_int, [VP, VP]),
    ("rbg_release_text", C.c_int, [VP, VP]),
    ("rbg_reserve_text", C.c_int, [VP, U64, C.c_int]),
    ("rbg_convert_runs", C.c_int, [VP, VP, U64, VP, VP, C.c_char_p]),
    ("rbg_convert_runs_markers", C.c_int, [VP, VP, U64, VP, VP, VP, VP, U64, VP, VP, C.c_char_p, C.c_char_p]),
    ("rbg_write_ftab", C.c_int, [VP, U64, C.c_char_p]),
    ("rbg_check_ftab", C.c_int, [VP, C.c_char_p, C.POINTER(U64)]),
    ("rbg_set_markers", C.c_int, [VP, VP, VP, U64, VP, VP]),
    ("rbg_set_docs", C.c_int, [VP, C.c_char_p, VP, U64]),
    ("rbg_free", None, [VP]),
    ("rbg_info", C.c_int, [VP, C.POINTER(Info)]),
    ("rbg_info_sized", C.c_int, [VP, C.POINTER(Info), U64]),
    ("rbg_layout_info", C.c_int, [VP, C.POINTER(LayoutInfo), U64]),
    ("rbg_get_f", C.c_int, [VP, VP]),
    ("rbg_last_run_sample", C.c_int, [VP, C.POINTER(U64)]),
    ("rbg_host_array", C.c_int, [VP, C.c_int, VP, U64, C.POINTER(U64)]),
    ("rbg_lf", C.c_int, [VP, VP, VP, VP, U64, VP, VP]),
    ("rbg_find_range", C.c_int, [VP, VP, VP, U64, VP, VP]),
    ("rbg_count", C.c_int, [VP, VP, VP, U64, VP]),
    ("rbg_find_range_w_toehold", C.c_int, [VP, VP, VP, U64, VP, VP, VP]),
    ("rbg_find_range_spans", C.c_int, [VP, VP, VP, VP, U64, VP, VP, VP]),
    ("rbg_locs_at", C.c_int, [VP, VP, VP, VP, U64, U64, VP, C.POINTER(VP)]),
    ("rbg_markers_at", C.c_int, [VP, VP, VP, U64, VP, C.POINTER(VP)]),
    ("rbg_find_range_w_markers", C.c_int, [VP, VP, VP, U64, U64, U64, VP, VP, VP, C.POINTER(VP)]),
    ("rbg_get_markers_greedy_seeding", C.c_int, [VP, VP, VP, U64, U64, U64, U64, VP, C.POINTER(VP), C.POINTER(VP)]),
    ("rbg_greedy_longest_seed", C.c_int, [VP, VP, VP, U64, U64, VP, VP, VP, VP, VP]),
    ("rbg_find_locs_greedy_seeding", C.c_int, [VP, VP, VP, U64, U64, U64, VP, C.POINTER(VP)]),
    ("rbg_free_buffer", None, [VP]),
    ("rbg_resolve_offset", C.c_int, [VP, U64, C.POINTER(C.c_char_p), C.POINTER(U64)]),
    ("rbg_find_range_dev", C.c_int, [VP, VP, VP, U64, VP, VP, VP]),
    ("rbg_find_range_w_toehold_dev", C.c_int, [VP, VP, VP, U64, VP, VP, VP, VP]),
    ("rbg_pack_ws_bytes", C.c_size_t, [U64, U64]),
    ("rbg_pack_reads_dev", C.c_int, [VP, VP, VP, U64, U64, VP, C.c_size_t, VP]),
    ("rbg_find_range_packed_dev", C.c_int, [VP, VP, VP, VP, U64, U64, VP, VP, VP]),
    ("rbg_find_range_w_toehold_packed_dev", C.c_int, [VP, VP, VP, VP, U64, U64, VP, VP, VP, VP]),
    ("rbg_locate_plan_tmp_bytes", C.c_size_t, [U64]),
    ("rbg_locate_plan_dev", C.c_int, [VP, VP, VP, U64, U64, VP, VP, C.c_size_t, VP]),
    ("rbg_locate_order_ws_bytes", C.c_size_t, [U64]),
    ("rbg_locate_order_dev", C.c_int, [VP, VP, U64, VP, C.c_size_t, VP]),
    ("rbg_locate_fill_dev", C.c_int, [VP, VP, VP, VP, U64, U64, VP, VP, VP, VP]),
    ("rbg_locate_fill_offset_dev", C.c_int, [VP, VP, VP, VP, U64, U64, VP, VP, VP, VP, VP]),
    ("rbg_greedy_longest_seed_dev", C.c_int, [VP, VP, VP, U64, U64, VP, VP, VP, VP, VP, VP]),
    ("rbg_marker_seeds_plan_dev", C.c_int, [VP, VP, VP, U64, U64, U64, U64, VP, VP, VP, C.c_size_t, VP]),
    ("rbg_marker_seeds_fill_dev", C.c_int, [VP, VP, VP, U64, U64, U64, U64, VP, VP, VP, VP, VP]),
    ("rbg_markers_plan_dev", C.c_int, [VP, VP, VP, U64, VP, VP, C.c_size_t, VP]),
    ("rbg_markers_fill_dev", C.c_int, [VP, VP, VP, U64, VP, VP, VP]),
    ("rbg_marker_seeds_log_bytes", C.c_size_t, [VP, U64, C.c_uint32]),
    ("rbg_marker_seeds_plan_log_dev", C.c_int, [VP, VP, VP, U64, U64, U64, U64, VP, VP, VP, C.c_size_t, VP, C.c_size_t, VP]),
    ("rbg_marker_seeds_fill_log_dev", C.c_int, [VP, VP, VP, U64, U64, U64, U64, VP, VP, VP, VP, VP, C.c_size_t, VP]),
    ("rbg_locate_fill_dev32", C.c_int, [VP, VP, VP, VP, U64, U64, VP, VP, VP, VP]),
    ("rbg_greedy_longest_seed_stats_dev", C.c_int, [VP, VP, VP, U64, U64, VP, VP, VP, VP, VP, VP, VP]),
    ("rbg_marker_seeds_stats_dev", C.c_int, [VP, VP, VP, U64, U64, U64, VP, VP, VP, C.c_size_t, VP, VP, VP, VP]),
    ("rbg_find_range_stats_dev", C.c_int, [VP, VP, VP, U64, VP, VP, VP, VP, VP]),
    ("rbg_locate_fill_stats_dev", C.c_int, [VP, VP, VP, VP, U64, U64, VP, VP, VP, VP, VP]),
    ("rbg_sample_reads_dev", C.c_int, [VP, U64, U64, U64, U64, U64, U64, U64, C.c_uint32, VP, VP, VP, VP]),
    ("rbg_sample_reads_pangenome_dev", C.c_int, [VP, VP, VP, VP, U64, VP, C.c_uint32, U64, U64, U64, U64, U64, U64, U64, C.c_uint32, VP, VP, VP, VP]),
    ("rbg_replicate", C.c_int, [VP, C.c_int, C.POINTER(VP)]),
    ("rbg_replicate_many", C.c_int, [VP, C.POINTER(C.c_int), C.c_int, C.POINTER(VP)]),
    ("rbg_replicate_stats", C.c_int, [VP, C.POINTER(C.c_double)]),
    ("rbg_comm_cache_clear", C.c_int, []),
    ("rbg_shard_bounds", C.c_int, [U64, C.c_int, C.c_int, C.POINTER(U64), C.POINTER(U64)]),
    ("rbg_find_range_sharded", C.c_int, [VP, C.c_int, VP, VP, U64, VP, VP, VP]),
    ("rbg_counters_allreduce", C.c_int, [VP, VP, VP, VP]),
    ("rbg_counters_allreduce_local", C.c_int, [VP, C.c_int, VP]),
    ("rbg_counters", C.c_int, [VP, VP]),
    ("rbg_counters_reset", C.c_int, [VP]),
    ("rbg_combine_stats", C.c_int, [VP, VP]),
    ("rbg_set_default_option", C.c_int, [C.c_int, C.c_int64]),
    ("rbg_get_default_option", C.c_int, [C.c_int, C.POINTER(C.c_int64)]),
]
EXPORTS = [p[0] for p in _PROTOS]

_lib = None


def _one_hip_runtime():
    """A process must run ONE HIP runtime.  PyTorch's ROCm wheels bundle their own copies of the ROCm libraries: when
    torch is imported AFTER librbg.so has pulled in /opt/rocm's, its device initialisation fails ("No HIP GPUs are
    available"); imported first, its runtime is the one librbg.so binds to as well (same sonames) -- the configuration
    bench.py and the tests run.  So, when PyTorch is installed and not loaded yet, load it first (about a second;
    RBG_NO_TORCH_PRELOAD=1 skips this for processes that never touch torch)."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("RBG_NO_TORCH_PRELOAD") == "1":
        return
    try:
        if importlib.util.find_spec("torch") is not None:
            import torch  # noqa: F401
    except Exception:  # noqa: BLE001  (a broken torch install must not keep the library from loading)
        pass


def lib():
    """Load librbg.so.  Fails loudly if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise ImportError(f"{_SO} is missing: build it with `make -C rowbowt_amd/csrc` "
                              "(or __graft_entry__.build()); there is no CPU fallback")
        _one_hip_runtime()
        L = C.CDLL(_SO)
        L.rbg_abi_version.restype = C.c_int
        L.rbg_abi_version.argtypes = []
        have = L.rbg_abi_version()
        if have != ABI_VERSION:   # (the structs above and the option numbers are this ABI's: never bind another one's symbols blindly)
            raise ImportError(f"{_SO} reports ABI {have}, this binding is written against ABI {ABI_VERSION}: rebuild it with `make -C rowbowt_amd/csrc`")
        for name, res, args in _PROTOS:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RbgError(rc, what)


def _p(a):
    return a.ctypes.data_as(VP) if a is not None else None


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def pack_reads(reads):
    """list of bytes -> (uint8 concat, uint64 offsets[N+1]): the batch layout of the C-ABI."""
    off = np.zeros(len(reads) + 1, dtype=np.uint64)
    if len(reads):
        off[1:] = np.cumsum([len(r) for r in reads], dtype=np.uint64)
    seqs = np.frombuffer(b"".join(reads), dtype=np.uint8).copy() if len(reads) else np.zeros(0, np.uint8)
    return seqs, off


def convert_index(prefix, flags, out_path):
    _check(lib().rbg_convert_index(os.fsencode(prefix), int(flags), os.fsencode(out_path)), "rbg_convert_index")


def convert_raw(bwt, ssa=None, esa=None, mab=None, docs=None, out_path=None):
    e = lambda x: os.fsencode(x) if x else None
    _check(lib().rbg_convert_raw(e(bwt), e(ssa), e(esa), e(mab), e(docs), e(out_path)), "rbg_convert_raw")


def convert_runs(heads, lens, ssa=None, esa=None, out_path=None, markers=None, docs_text=None):
    """the native cache file from a run-length BWT in memory (rbg_convert_runs[_markers]); markers = (start, end, off, vals)"""
    heads = np.ascontiguousarray(heads, dtype=np.uint8)
    lens = np.ascontiguousarray(lens, dtype=np.uint64)
    ssa = None if ssa is None else np.ascontiguousarray(ssa, dtype=np.uint64)
    esa = None if esa is None else np.ascontiguousarray(esa, dtype=np.uint64)
    mk = [None] * 4 if markers is None else [np.ascontiguousarray(a, dtype=np.uint64) for a in markers]
    _check(lib().rbg_convert_runs_markers(_p(heads), _p(lens), len(heads), _p(ssa) if ssa is not None else None, _p(esa) if esa is not None else None,
                                          _p(mk[0]) if markers is not None else None, _p(mk[1]) if markers is not None else None,
                                          len(mk[0]) if markers is not None else 0, _p(mk[2]) if markers is not None else None,
                                          _p(mk[3]) if markers is not None else None, docs_text.encode() if docs_text else None,
                                          os.fsencode(out_path)), "rbg_convert_runs_markers")


def set_default_option(opt, value):
    _check(lib().rbg_set_default_option(opt, value), "rbg_set_default_option")


def get_default_option(opt):
    v = C.c_int64(0)
    _check(lib().rbg_get_default_option(opt, C.byref(v)), "rbg_get_default_option")
    return int(v.value)


class default_option:
    """with default_option(OPT_X, v): ... -- one knob changed for the loads inside, put back afterwards"""

    def __init__(self, opt, value):
        self.opt, self.value = opt, value

    def __enter__(self):
        self.prev = get_default_option(self.opt)
        set_default_option(self.opt, self.value)
        return self

    def __exit__(self, *exc):
        set_default_option(self.opt, self.prev)
        return False


def _take(ptr, n):
    """a library-malloc'ed uint64 result as a numpy array, without copying (gigabytes of locations); the
    buffer is released (rbg_free_buffer) when the last array viewing it is gone"""
    if n == 0 or not ptr.value:
        lib().rbg_free_buffer(ptr)
        return np.zeros(0, dtype=np.uint64)
    mem = (C.c_char * (n * 8)).from_address(ptr.value)
    weakref.finalize(mem, lib().rbg_free_buffer, VP(ptr.value))
    return np.frombuffer(mem, dtype=np.uint64)   # .base is `mem`: every view keeps it alive


class RowBowt:
    """rbwt::RowBowt<ri::rle_string_sd> (rowbowt.hpp:23-24), batched over N reads."""

    def __init__(self, handle):
        self.h = handle
        self.L = lib()

    def close(self):
        if self.h:
            self.L.rbg_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @classmethod
    def from_runs(cls, heads, lens, ssa=None, esa=None, device=0):
        heads = np.ascontiguousarray(heads, dtype=np.uint8)
        lens = _u64(lens)
        ssa = _u64(ssa) if ssa is not None else None
        esa = _u64(esa) if esa is not None else None
        h = VP()
        _check(lib().rbg_build_from_runs(_p(heads), _p(lens), len(heads), _p(ssa), _p(esa), device, C.byref(h)),
               "rbg_build_from_runs")
        return cls(h)

    @classmethod
    def from_files(cls, bwt, ssa=None, esa=None, device=0):
        """rb_build's raw inputs (rb_build.cpp:83-93): <pre>.bwt, <pre>.ssa, <pre>.esa"""
        h = VP()
        _check(lib().rbg_build_from_files(os.fsencode(bwt), os.fsencode(ssa) if ssa else None,
                                          os.fsencode(esa) if esa else None, device, C.byref(h)), "rbg_build_from_files")
        return cls(h)

    def set_markers(self, run_start, run_end, mk_off, mk_vals):
        a = [_u64(v) for v in (run_start, run_end, mk_off, mk_vals)]
        _check(self.L.rbg_set_markers(self.h, _p(a[0]), _p(a[1]), len(a[0]), _p(a[2]), _p(a[3])), "rbg_set_markers")

    def set_docs(self, names, starts):
        joined = b"\0".join(n.encode() for n in names) + b"\0"
        s = _u64(starts)
        _check(self.L.rbg_set_docs(self.h, joined, _p(s), len(names)), "rbg_set_docs")

    @classmethod
    def from_cache(cls, path, flags=LoadRbwtFlag.NONE, device=0):
        """native cache file written by convert_index / convert_raw / rb_build (include/rbg.h, next-row f1)"""
        h = VP()
        _check(lib().rbg_load_cache(os.fsencode(path), int(flags), device, C.byref(h)), "rbg_load_cache")
        return cls(h)

    def write_ftab(self, k, path):
        """RowBowt::build_ftab(k) + FTab::serialize (rowbowt.hpp:726-744, ftab.hpp:29-34)"""
        _check(self.L.rbg_write_ftab(self.h, k, os.fsencode(path)), "rbg_write_ftab")

    def check_ftab(self, path):
        """-> k of a .ftab file that is exactly build_ftab(k) of this index (LoadRbwtFlag::FT), else RbgError"""
        k = U64()
        _check(self.L.rbg_check_ftab(self.h, os.fsencode(path), C.byref(k)), "rbg_check_ftab")
        return k.value

    # ---- introspection
    def info(self):
        i = Info()
        _check(self.L.rbg_info(self.h, C.byref(i)), "rbg_info")
        return i

    def layout_info(self):
        """what the load decided about the run-indexed layout (rbg_layout_info_t; all zero on the slot layout)"""
        i = LayoutInfo()
        _check(self.L.rbg_layout_info(self.h, C.byref(i), C.sizeof(i)), "rbg_layout_info")
        return i

    def get_f(self):
        out = np.zeros(256, dtype=np.uint64)
        _check(self.L.rbg_get_f(self.h, _p(out)), "rbg_get_f")
        return out

    def last_run_sample(self):
        v = U64()
        _check(self.L.rbg_last_run_sample(self.h, C.byref(v)), "rbg_last_run_sample")
        return v.value

    def host_array(self, which):
        cnt = U64()
        _check(self.L.rbg_host_array(self.h, which, None, 0, C.byref(cnt)), "rbg_host_array")
        out = np.zeros(cnt.value, dtype=np.uint64)
        _check(self.L.rbg_host_array(self.h, which, _p(out), cnt.value, C.byref(cnt)), "rbg_host_array")
        return out

    # ---- queries (host buffers)
    def LF(self, lo, hi, sym):
        """RowBowt::LF(range_t, uint8_t), rowbowt.hpp:74-88, for N triples"""
        lo, hi = _u64(lo), _u64(hi)
        sym = np.ascontiguousarray(sym, dtype=np.uint8)
        N = len(lo)
        nlo, nhi = np.zeros(N, np.uint64), np.zeros(N, np.uint64)
        _check(self.L.rbg_lf(self.h, _p(lo), _p(hi), _p(sym), N, _p(nlo), _p(nhi)), "rbg_lf")
        return nlo, nhi

    def find_range(self, seqs, off):
        N = len(off) - 1
        lo, hi = np.zeros(N, np.uint64), np.zeros(N, np.uint64)
        _check(self.L.rbg_find_range(self.h, _p(seqs), _p(off), N, _p(lo), _p(hi)), "rbg_find_range")
        return lo, hi

    def count(self, seqs, off):
        N = len(off) - 1
        cnt = np.zeros(N, np.uint64)
        _check(self.L.rbg_count(self.h, _p(seqs), _p(off), N, _p(cnt)), "rbg_count")
        return cnt

    def find_range_spans(self, buf, begin, length, toehold=False):
        """reads as spans of one host buffer (rbg_find_range_spans)"""
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        begin = _u64(begin)
        length = np.ascontiguousarray(length, dtype=np.uint32)
        N = len(begin)
        lo, hi = np.zeros(N, np.uint64), np.zeros(N, np.uint64)
        k = np.zeros(N, np.uint64) if toehold else None
        _check(self.L.rbg_find_range_spans(self.h, _p(buf), _p(begin), _p(length), N, _p(lo), _p(hi), _p(k)), "rbg_find_range_spans")
        return (lo, hi, k) if toehold else (lo, hi)

    def find_range_w_toehold(self, seqs, off):
        N = len(off) - 1
        lo, hi, k = np.zeros(N, np.uint64), np.zeros(N, np.uint64), np.zeros(N, np.uint64)
        _check(self.L.rbg_find_range_w_toehold(self.h, _p(seqs), _p(off), N, _p(lo), _p(hi), _p(k)),
               "rbg_find_range_w_toehold")
        return lo, hi, k

    def locs_at(self, lo, hi, k, max_hits=MAXU):
        lo, hi, k = _u64(lo), _u64(hi), _u64(k)
        N = len(lo)
        loc_off = np.zeros(N + 1, np.uint64)
        ptr = VP()
        _check(self.L.rbg_locs_at(self.h, _p(lo), _p(hi), _p(k), N, max_hits, _p(loc_off), C.byref(ptr)), "rbg_locs_at")
        return loc_off, _take(ptr, int(loc_off[N]))

    def markers_at(self, lo, hi):
        lo, hi = _u64(lo), _u64(hi)
        N = len(lo)
        mk_off = np.zeros(N + 1, np.uint64)
        ptr = VP()
        _check(self.L.rbg_markers_at(self.h, _p(lo), _p(hi), N, _p(mk_off), C.byref(ptr)), "rbg_markers_at")
        return mk_off, _take(ptr, int(mk_off[N]))

    def find_range_w_markers(self, seqs, off, wsize, max_range=MAXU):
        N = len(off) - 1
        lo, hi, mk_off = np.zeros(N, np.uint64), np.zeros(N, np.uint64), np.zeros(N + 1, np.uint64)
        ptr = VP()
        _check(self.L.rbg_find_range_w_markers(self.h, _p(seqs), _p(off), N, wsize, max_range & MAXU,
                                               _p(lo), _p(hi), _p(mk_off), C.byref(ptr)), "rbg_find_range_w_markers")
        return lo, hi, mk_off, _take(ptr, int(mk_off[N]))

    def get_markers_greedy_seeding(self, seqs, off, wsize, max_range=MAXU, ftab_k=0):
        """RowBowt::get_markers_greedy_seeding (rowbowt.hpp:406-482), ftab_k = k-mer size of the loaded
        ftab or 0: -> seed_off[N+1], seeds[S,6] = (lo, hi, qstart, qend, mk_begin, mk_end), mk"""
        N = len(off) - 1
        seed_off = np.zeros(N + 1, np.uint64)
        ps, pm = VP(), VP()
        _check(self.L.rbg_get_markers_greedy_seeding(self.h, _p(seqs), _p(off), N, wsize, max_range & MAXU, ftab_k, _p(seed_off),
                                                     C.byref(ps), C.byref(pm)), "rbg_get_markers_greedy_seeding")
        S = int(seed_off[N])
        seeds = _take(ps, 6 * S).reshape(S, 6)
        nmk = int(seeds[-1, 5]) if S else 0
        return seed_off, seeds, _take(pm, nmk)

    def greedy_longest_seed(self, seqs, off, min_length):
        """get_seeds_greedy_w_sample (rowbowt.hpp:222-256) -> the seed locate_from_longest_seed picks (:669-677)"""
        N = len(off) - 1
        a = [np.zeros(N, np.uint64) for _ in range(5)]
        _check(self.L.rbg_greedy_longest_seed(self.h, _p(seqs), _p(off), N, min_length, *[_p(x) for x in a]),
               "rbg_greedy_longest_seed")
        return a  # lo, hi, qstart, qend, ssamp

    def find_locs_greedy_seeding(self, seqs, off, min_length, max_hits=MAXU):
        """RowBowt::find_locs_greedy_seeding, rowbowt.hpp:633-657"""
        N = len(off) - 1
        loc_off = np.zeros(N + 1, np.uint64)
        ptr = VP()
        _check(self.L.rbg_find_locs_greedy_seeding(self.h, _p(seqs), _p(off), N, min_length, max_hits, _p(loc_off), C.byref(ptr)),
               "rbg_find_locs_greedy_seeding")
        return loc_off, _take(ptr, int(loc_off[N]))

    def resolve_offset(self, i):
        name, off = C.c_char_p(), U64()
        _check(self.L.rbg_resolve_offset(self.h, i, C.byref(name), C.byref(off)), "rbg_resolve_offset")
        return name.value.decode(), off.value

    def align_text(self, lo, hi, k, names, max_hits=MAXU, markers=False):
        """rbg_align_text: the `rb_align -s` text of a batch (bytes), made on the device; names = list of bytes"""
        lo, hi = (np.ascontiguousarray(a, dtype=np.uint64) for a in (lo, hi))
        k = None if k is None else np.ascontiguousarray(k, dtype=np.uint64)   # None: the count-only report
        blob = b"".join(names)
        nlen = np.array([len(x) for x in names], dtype=np.uint32)
        nbeg = np.zeros(len(names), dtype=np.uint64)
        if len(names) > 1:
            nbeg[1:] = np.cumsum(nlen[:-1], dtype=np.uint64)
        buf = C.create_string_buffer(blob, len(blob) + 1)
        text, n = VP(), U64()
        _check(self.L.rbg_align_text(self.h, _p(lo), _p(hi), _p(k) if k is not None else None, len(names), max_hits, 1 if markers else 0, buf, _p(nbeg), _p(nlen), C.byref(text), C.byref(n)), "rbg_align_text")
        try:
            _check(self.L.rbg_wait_text(self.h, text), "rbg_wait_text")
            return C.string_at(text, n.value)
        finally:
            self.L.rbg_release_text(self.h, text)

    def counters(self):
        out = np.zeros(4, np.uint64)
        _check(self.L.rbg_counters(self.h, _p(out)), "rbg_counters")
        return out

    # ---- several GPUs in one process
    def replicate(self, device):
        """a further replica of the device index on `device` (peer copy); free it before this one"""
        h = VP()
        _check(self.L.rbg_replicate(self.h, device, C.byref(h)), "rbg_replicate")
        r = RowBowt(h)
        r._primary = self   # keeps the primary alive
        return r

    def replicate_many(self, devices):
        """replicas on every device of `devices` at once (the peer copies overlap); free them before this one"""
        G = len(devices)
        devs = (C.c_int * G)(*devices)
        hs = (VP * G)()
        _check(self.L.rbg_replicate_many(self.h, devs, G, hs), "rbg_replicate_many")
        out = []
        for g in range(G):
            r = RowBowt(VP(hs[g]))
            r._primary = self
            out.append(r)
        return out

    def replicate_stats(self):
        """a replica's share of the fan-out: {ms of its peer copies, bytes, GB/s, peer access}"""
        out = (C.c_double * 3)()
        _check(self.L.rbg_replicate_stats(self.h, out), "rbg_replicate_stats")
        ms, nbytes, peer = float(out[0]), int(out[1]), int(out[2])
        return {"copy_ms": ms, "bytes": nbytes, "GBps": (nbytes / (ms * 1e-3) / 1e9) if ms > 0 else None,
                "peer_access": {1: "direct", 0: "staged through the host", -1: "same device"}[peer]}

    def counters_reset(self):
        _check(self.L.rbg_counters_reset(self.h), "rbg_counters_reset")

    def combine_stats(self):
        """(launches, requests) of the one-read calls that went through the micro-batching queue"""
        out = np.zeros(2, np.uint64)
        _check(self.L.rbg_combine_stats(self.h, _p(out)), "rbg_combine_stats")
        return int(out[0]), int(out[1])


def load_rowbowt(prefix, flag=LoadRbwtFlag.NONE, device=0):
    """rbwt::load_rowbowt(prefix, flag), rowbowt_io.hpp:176-189"""
    h = VP()
    _check(lib().rbg_load(os.fsencode(prefix), int(flag), device, C.byref(h)), f"rbg_load({prefix})")
    return RowBowt(h)


def shard_bounds(n_items, rank, world):
    b, e = U64(), U64()
    _check(lib().rbg_shard_bounds(n_items, rank, world, C.byref(b), C.byref(e)), "rbg_shard_bounds")
    return b.value, e.value


def find_range_sharded(replicas, seqs, off, toehold=False):
    """find_range / find_range_w_toehold with the batch sharded over the replicas (contiguous blocks, concurrently)"""
    N = len(off) - 1
    hs = (VP * len(replicas))(*[r.h for r in replicas])
    lo, hi = np.zeros(N, np.uint64), np.zeros(N, np.uint64)
    k = np.zeros(N, np.uint64) if toehold else None
    _check(lib().rbg_find_range_sharded(hs, len(replicas), _p(seqs), _p(off), N, _p(lo), _p(hi), _p(k)), "rbg_find_range_sharded")
    return (lo, hi, k) if toehold else (lo, hi)


def counters_allreduce_local(replicas):
    """one RCCL all-reduce of the four counters over replicas on distinct devices of this process"""
    hs = (VP * len(replicas))(*[r.h for r in replicas])
    out = np.zeros(4, np.uint64)
    _check(lib().rbg_counters_allreduce_local(hs, len(replicas), _p(out)), "rbg_counters_allreduce_local")
    return out
