"""CPU-only checks of the product's host side: the C-ABI library loads and exports every symbol
include/rbg.h declares, the sdsl readers + flattener reproduce what the oracle's independent
reader decodes, and query entry points refuse to run without a device (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import orc
import rowbowt_amd as ra
from rowbowt_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_match_header():
    hdr = open(os.path.join(ROOT, "include", "rbg.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rbg_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(capi.EXPORTS)
    L = ra.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.rbg_abi_version() == 3 == capi.ABI_VERSION
    assert b"CPU" in L.rbg_strerror(-3)


@pytest.fixture(scope="module")
def small_host(data_dir):
    rb = ra.load_rowbowt(os.path.join(data_dir, "small.fa"), ra.LoadRbwtFlag.SA | ra.LoadRbwtFlag.MA, device=capi.DEVICE_NONE)
    yield rb
    rb.close()


@pytest.fixture(scope="module")
def small_orc(data_dir):
    o = orc.Oracle.load(os.path.join(data_dir, "small.fa"), orc.SA | orc.MA)
    yield o
    o.close()


def test_loader_matches_oracle_reader(small_host, small_orc):
    i = small_host.info()
    assert (i.n, i.r, i.sigma, i.pos_bytes, i.device) == (30031, 7573, 5, 4, -1)
    assert i.has_tsa and i.has_markers and not i.has_docs
    heads, lens = small_orc.runs()
    assert (small_host.host_array(capi.ARR_RUN_HEADS) == heads).all()
    starts = np.concatenate(([0], np.cumsum(lens))).astype(np.uint64)
    assert (small_host.host_array(capi.ARR_RUN_START) == starts).all()
    assert (small_host.get_f() == small_orc.f()).all()
    pred_pos, samples_last, pred_to_run = small_orc.tsa()
    assert (small_host.host_array(capi.ARR_PRED_POS) == pred_pos).all()
    assert (small_host.host_array(capi.ARR_SAMPLES_LAST) == samples_last).all()
    base = np.where(pred_to_run > 0, samples_last[np.maximum(pred_to_run.astype(np.int64) - 1, 0)], 0)
    assert (small_host.host_array(capi.ARR_PHI_BASE) == base.astype(np.uint64)).all()
    assert small_host.last_run_sample() == small_orc.last_run_sample()
    s, e, o, v = small_orc.markers()
    assert (small_host.host_array(capi.ARR_MARKER_START) == s).all()
    assert (small_host.host_array(capi.ARR_MARKER_END) == e).all()
    assert (small_host.host_array(capi.ARR_MARKER_OFF) == o).all()
    assert (small_host.host_array(capi.ARR_MARKER_VALS) == v).all()
    assert i.marker_runs == 190 and i.marker_vals == 190


def test_two_step_tables_built(small_host, data_dir, monkeypatch):
    # a host-only handle answers no query, so by default it composes no k-mer table (eight depths on the host cost
    # hundreds of GB at pangenome r for tables nothing reads) ...
    i = small_host.info()
    assert i.kmer_steps == 1 and list(i.depth_runs)[1:] == [0] * 7
    # ... RBG_HOST_COMPOSE=1 keeps the host composition (the serial statement of k_compose.hip) reachable without a device
    monkeypatch.setenv("RBG_HOST_COMPOSE", "1")
    rb = ra.load_rowbowt(os.path.join(data_dir, "small.fa"), ra.LoadRbwtFlag.SA | ra.LoadRbwtFlag.MA, device=capi.DEVICE_NONE)
    i = rb.info()
    assert i.kmer_steps == 8 and i.kmer_symbols == 4   # (RBG_OPT_KMER_STEPS defaults to 8)
    assert 0 < i.pair_runs <= 2 * i.r and i.pair_runs <= i.triple_runs <= 3 * i.r and i.triple_runs <= i.quad_runs <= 4 * i.r
    assert i.quad_runs <= i.quint_runs <= 5 * i.r
    runs = list(i.depth_runs)
    assert runs[0] == i.r and runs[1:5] == [i.pair_runs, i.triple_runs, i.quad_runs, i.quint_runs]
    assert all(runs[d - 1] <= runs[d] <= (d + 1) * i.r for d in range(1, 8))   # every depth cuts each run of the one before at most once more


def test_greedy_seeding_fixture(data_dir):
    rb = ra.load_rowbowt(os.path.join(data_dir, "greedy_seeding", "ref.fa"),
                         ra.LoadRbwtFlag.SA | ra.LoadRbwtFlag.DL, device=capi.DEVICE_NONE)
    i = rb.info()
    assert (i.n, i.r) == (20047, 14949) and i.has_docs
    assert rb.resolve_offset(1234) == ("greedy_seeding", 1234)
    assert rb.resolve_offset(0) == ("greedy_seeding", 0)
    rb.close()


def test_resolve_offset_matches_oracle(small_host, small_orc):
    names, starts = ["ref", "hap1", "hap2"], [0, 10010, 20020]
    small_host.set_docs(names, starts)
    small_orc.set_docs(names, starts)
    for i in (0, 1, 9999, 10009, 10010, 10011, 20019, 20020, 25000, 30030):
        assert small_host.resolve_offset(i) == small_orc.resolve_offset(i)


def test_errors(tmp_path, data_dir):
    with pytest.raises(ra.RbgError) as e:
        ra.load_rowbowt(str(tmp_path / "nope"), device=capi.DEVICE_NONE)
    assert e.value.code == -1  # RBG_EIO: reference prints "bad file" and exit(1)s
    bad = tmp_path / "bad.rbwt"
    raw = open(os.path.join(data_dir, "small.fa.rbwt"), "rb").read()
    bad.write_bytes(raw[:-7])
    with pytest.raises(ra.RbgError) as e:
        ra.load_rowbowt(str(tmp_path / "bad"), device=capi.DEVICE_NONE)
    assert e.value.code == -2
    bad.write_bytes(raw + b"\0")
    with pytest.raises(ra.RbgError) as e:
        ra.load_rowbowt(str(tmp_path / "bad"), device=capi.DEVICE_NONE)
    assert e.value.code == -2
    # .tsa requested but absent
    (tmp_path / "only.rbwt").write_bytes(raw)
    with pytest.raises(ra.RbgError) as e:
        ra.load_rowbowt(str(tmp_path / "only"), ra.LoadRbwtFlag.SA, device=capi.DEVICE_NONE)
    assert e.value.code == -1


def test_host_side_2bit_packer(tmp_path):
    """rbg_pack2bit.hpp (the packer of the host-pointer pipeline and of the tools' FASTQ front end) against a
    bit-by-bit restatement of the device layout, under ASan + UBSan (tests/cpp/pack2bit_check.cpp)"""
    import subprocess
    exe = tmp_path / "pack2bit"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           os.path.join(ROOT, "tests", "cpp", "pack2bit_check.cpp"), "-o", str(exe)])
    p = subprocess.run([str(exe)], capture_output=True, timeout=120)
    assert p.returncode == 0 and b"pack2bit ok" in p.stdout, p.stdout[-300:] + p.stderr[-300:]


def test_host_thread_team(tmp_path):
    """rbg_thread_team.hpp (spin-then-sleep worker team of the host-pointer pipeline) under ThreadSanitizer"""
    import subprocess
    exe = tmp_path / "team"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=thread",
                           os.path.join(ROOT, "tests", "cpp", "thread_team_check.cpp"), "-o", str(exe)])
    p = subprocess.run([str(exe)], capture_output=True, timeout=300)
    assert p.returncode == 0 and b"thread team ok" in p.stdout, p.stdout[-300:] + p.stderr[-2000:]


def test_python_binding_surface():
    """the ctypes wrapper the tests and bench.py drive the ABI through keeps its methods (a misplaced edit once
    turned one into dead code)"""
    for m in ("find_range", "count", "find_range_w_toehold", "locs_at", "markers_at", "find_range_w_markers", "get_markers_greedy_seeding",
              "greedy_longest_seed", "find_locs_greedy_seeding", "LF", "counters", "counters_reset", "replicate", "info", "set_markers",
              "set_docs", "write_ftab", "check_ftab", "resolve_offset", "close"):
        assert callable(getattr(capi.RowBowt, m)), m
    for f in ("shard_bounds", "find_range_sharded", "counters_allreduce_local", "set_default_option", "lib"):
        assert callable(getattr(capi, f)), f
    assert capi.shard_bounds(10, 1, 3) == (3, 6)


def test_tuning_options_are_range_checked():
    """rbg_set_default_option (include/rbg.h "tuning"): every knob refuses values outside its documented range
    (RBG_EARG) and leaves the previous value in force; the defaults are restored afterwards."""
    L = ra.lib()
    ok = {capi.OPT_BLOCK_THREADS: [64, 128, 256], capi.OPT_RANK_BUCKET_SHIFT: [0, 8, 12, -1], capi.OPT_PHI_BUCKET_SHIFT: [0, 8, -1],
          capi.OPT_POS_BYTES: [4, 8, 0], capi.OPT_KMER_STEPS: [1, 3, 5, 8], capi.OPT_HBM_BUDGET_MB: [1, 0], capi.OPT_FTAB_K: [0, 16, -1],
          capi.OPT_PACKED_READS: [0, 2, 1], capi.OPT_DEEP_BUCKET_SHIFT: [9, 12, -1], capi.OPT_DENSE_OVERFLOW: [0, 1],
          capi.OPT_RANK_LAYOUT: [1, 2, 3, 0], capi.OPT_RUN_DEPTHS: [0x15, 31, 255, 0],
          capi.OPT_RUN_PHI: [1, 2, 0], capi.OPT_RUN_REC: [1, 2, 0], capi.OPT_RUN_REC_DEPTHS: [0x80, 0x11, 0]}
    bad = {capi.OPT_BLOCK_THREADS: [0, 100, 512], capi.OPT_RANK_BUCKET_SHIFT: [-2, 13], capi.OPT_PHI_BUCKET_SHIFT: [-2, 9],
           capi.OPT_POS_BYTES: [2, 16], capi.OPT_KMER_STEPS: [0, 9], capi.OPT_HBM_BUDGET_MB: [-1], capi.OPT_FTAB_K: [-2, 17],
           capi.OPT_PACKED_READS: [-1, 3], capi.OPT_DEEP_BUCKET_SHIFT: [-2, 13], capi.OPT_DENSE_OVERFLOW: [-1, 2],
           capi.OPT_RANK_LAYOUT: [-1, 4], capi.OPT_RUN_DEPTHS: [-1, 256],
           capi.OPT_RUN_PHI: [-1, 3], capi.OPT_RUN_REC: [-1, 3], capi.OPT_RUN_REC_DEPTHS: [-1, 256]}
    for opt, vals in ok.items():
        for v in bad[opt]:
            assert L.rbg_set_default_option(opt, v) == -4, (opt, v)   # RBG_EARG
        for v in vals:                                                 # the last value of each list is the default
            assert L.rbg_set_default_option(opt, v) == 0, (opt, v)
    assert L.rbg_set_default_option(0, 1) == -4 and L.rbg_set_default_option(19, 1) == -4
    for retired in (12, 13, 15):   # RBG_OPT_TREE_TOP_KB, _SLOT_BYTES, _RUN_FMT: gone with ABI 3, together with the kernels they selected
        assert L.rbg_set_default_option(retired, 2) == -4 and L.rbg_get_default_option(retired, C.byref(C.c_int64())) == -4
    assert all(capi.get_default_option(o) == vals[-1] for o, vals in ok.items())


def test_no_cpu_compute_path(small_host):
    """Without a device the query entry points must refuse (RBG_ENODEV), not compute on the CPU."""
    seqs, off = ra.pack_reads([b"ACGT"])
    for fn in (lambda: small_host.find_range(seqs, off), lambda: small_host.count(seqs, off),
               lambda: small_host.find_range_w_toehold(seqs, off),
               lambda: small_host.locs_at([0], [1], [5]), lambda: small_host.markers_at([0], [1]),
               lambda: small_host.find_range_w_markers(seqs, off, 2)):
        with pytest.raises(ra.RbgError) as e:
            fn()
        assert e.value.code == -3


def test_build_from_runs_layout():
    heads = np.frombuffer(b"ACAG\x01T", dtype=np.uint8)
    lens = np.array([3, 2, 1, 4, 1, 2], dtype=np.uint64)
    rb = ra.RowBowt.from_runs(heads, lens, device=capi.DEVICE_NONE)
    i = rb.info()
    assert (i.n, i.r, i.sigma) == (13, 6, 5)
    f = rb.get_f()
    assert [int(f[c]) for c in (1, 65, 67, 71, 84)] == [0, 1, 5, 7, 11]
    assert rb.host_array(capi.ARR_RUN_START).tolist() == [0, 3, 5, 6, 10, 11, 13]
    rb.close()
    with pytest.raises(ra.RbgError):  # non-maximal runs
        ra.RowBowt.from_runs(np.frombuffer(b"AAC", dtype=np.uint8), np.array([1, 1, 1], np.uint64), device=capi.DEVICE_NONE)


def _compile_shim_test(tmp_path):
    import subprocess
    exe = tmp_path / "shim_goldens"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "rowbowt_amd", "include"),
                           os.path.join(ROOT, "tests", "cpp", "shim_goldens.cpp"), "-o", str(exe),
                           "-L", os.path.join(ROOT, "rowbowt_amd"), "-lrbg", "-Wl,-rpath," + os.path.join(ROOT, "rowbowt_amd")])
    return exe


def test_cpp_shim_compiles_against_reference_style_calls(tmp_path):
    """rowbowt_gpu.hpp must accept the call shapes of the reference's tests (host compile only here)."""
    assert _compile_shim_test(tmp_path).exists()


def test_build_from_raw_files(tmp_path):
    """rb_build's raw inputs: .bwt bytes (whitespace bytes skipped, 0 -> 1), .ssa/.esa (x,y) pairs."""
    import naive
    from synth import SynthIndex
    S = SynthIndex(L=600, H=4, n_sites=12, seed=3)
    bwt = naive.bwt_from_sa(S.text, S.fm.sa).copy()
    raw = bytearray(bwt.tobytes().replace(b"\x01", b"\x00"))  # pfbwt writes the terminator as byte 0
    raw[100:100] = b"\n \t"                                    # formatted extraction skips these
    (tmp_path / "x.bwt").write_bytes(bytes(raw))
    pairs = lambda y: np.stack([np.arange(len(y), dtype=np.uint64) * 7, y.astype(np.uint64)], axis=1).tobytes()
    (tmp_path / "x.ssa").write_bytes(pairs(S.ssa))
    (tmp_path / "x.esa").write_bytes(pairs(S.esa) + b"\x01\x02\x03")  # trailing partial pair is ignored
    a = ra.RowBowt.from_files(str(tmp_path / "x.bwt"), str(tmp_path / "x.ssa"), str(tmp_path / "x.esa"), device=capi.DEVICE_NONE)
    b = ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=capi.DEVICE_NONE)
    for which in (capi.ARR_RUN_HEADS, capi.ARR_RUN_START, capi.ARR_SAMPLES_LAST, capi.ARR_PRED_POS, capi.ARR_PHI_BASE):
        assert (a.host_array(which) == b.host_array(which)).all()
    assert (a.get_f() == b.get_f()).all() and a.last_run_sample() == b.last_run_sample()
    c = ra.RowBowt.from_files(str(tmp_path / "x.bwt"), device=capi.DEVICE_NONE)
    assert c.info().r == a.info().r and not c.info().has_tsa
    (tmp_path / "short.ssa").write_bytes(pairs(S.ssa[:-1]))
    with pytest.raises(ra.RbgError) as e:
        ra.RowBowt.from_files(str(tmp_path / "x.bwt"), str(tmp_path / "short.ssa"), str(tmp_path / "x.esa"), device=capi.DEVICE_NONE)
    assert e.value.code == -2
    with pytest.raises(ra.RbgError) as e:
        ra.RowBowt.from_files(str(tmp_path / "missing.bwt"), device=capi.DEVICE_NONE)
    assert e.value.code == -1


HOST_ARRAYS = (capi.ARR_RUN_HEADS, capi.ARR_RUN_START, capi.ARR_SAMPLES_LAST, capi.ARR_PRED_POS, capi.ARR_PHI_BASE,
               capi.ARR_MARKER_START, capi.ARR_MARKER_END, capi.ARR_MARKER_OFF, capi.ARR_MARKER_VALS)


def _same_index(a, b, arrays=HOST_ARRAYS):
    for which in arrays:
        assert (a.host_array(which) == b.host_array(which)).all(), which
    assert (a.get_f() == b.get_f()).all()
    ia, ib = a.info(), b.info()
    assert (ia.n, ia.r, ia.sigma, ia.has_tsa, ia.has_markers, ia.has_docs) == (ib.n, ib.r, ib.sigma, ib.has_tsa, ib.has_markers, ib.has_docs)


def test_cache_from_runs_in_memory(tmp_path):
    """rbg_convert_runs: the cache file from a run-length BWT in memory (what bench.py's rank 0 hands the other ranks of a
    node, and what a builder that never writes the BWT as text uses) loads as the same index as rbg_build_from_runs on the
    same arrays; the parallel flatten (two passes over the runs split over threads) and the parallel radix sort of the
    samples give what the serial loops gave -- also with one thread and with more threads than runs per chunk."""
    from synth import SynthIndex
    S = SynthIndex(L=900, H=5, n_sites=20, seed=11)
    b = ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=capi.DEVICE_NONE)
    capi.convert_runs(S.heads, S.lens, S.ssa, S.esa, out_path=str(tmp_path / "m.rbgpu"))
    a = ra.RowBowt.from_cache(str(tmp_path / "m.rbgpu"), ra.LoadRbwtFlag.SA, device=capi.DEVICE_NONE)
    _same_index(a, b, HOST_ARRAYS[:5])
    assert a.last_run_sample() == b.last_run_sample()
    capi.convert_runs(S.heads, S.lens, out_path=str(tmp_path / "n.rbgpu"))
    c = ra.RowBowt.from_cache(str(tmp_path / "n.rbgpu"), device=capi.DEVICE_NONE)
    assert c.info().r == b.info().r and not c.info().has_tsa
    with pytest.raises(ra.RbgError):
        capi.convert_runs(S.heads, S.lens, S.ssa, None, out_path=str(tmp_path / "bad.rbgpu"))
    with pytest.raises(ra.RbgError):
        capi.convert_runs(S.heads, S.lens, S.ssa, S.esa, out_path=str(tmp_path / "no_such_dir" / "x.rbgpu"))
    # the file appears under its name only when complete (written as <name>.tmp.<pid>, then renamed); a failed write leaves nothing
    assert sorted(q.name for q in tmp_path.iterdir()) == ["m.rbgpu", "n.rbgpu"]
    # a larger random run list through 1 and 7 worker threads (a fresh interpreter each: the thread count is read once)
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r); import rowbowt_amd as ra; from rowbowt_amd import capi\n"
        "rng = np.random.default_rng(5); r = 300000\n"
        "step = rng.integers(1, 4, size=r); step[0] = 0; heads = np.frombuffer(b'ACGT', dtype=np.uint8)[np.cumsum(step) %% 4].copy()\n"
        "lens = rng.integers(1, 60, size=r).astype(np.uint64); heads[r // 3] = 1; lens[r // 3] = 1; n = int(lens.sum())\n"
        "vals = rng.permutation(n)[:2 * r].astype(np.uint64) + 1; vals[vals > n] = n\n"
        "rb = ra.RowBowt.from_runs(heads, lens, vals[:r], vals[r:], device=capi.DEVICE_NONE)\n"
        "import hashlib; h = hashlib.sha256()\n"
        "for w in (capi.ARR_RUN_HEADS, capi.ARR_RUN_START, capi.ARR_SAMPLES_LAST, capi.ARR_PRED_POS, capi.ARR_PHI_BASE): h.update(rb.host_array(w).tobytes())\n"
        "h.update(rb.get_f().tobytes()); print(h.hexdigest(), rb.info().pair_runs, rb.info().quint_runs)\n" % ROOT)
    outs = []
    for threads in ("1", "7", "64"):
        p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RBG_LOAD_THREADS=threads), capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(p.stdout.strip())
    assert outs[0] == outs[1] == outs[2] and len(outs[0]) > 64


def test_load_time_knobs_from_the_environment():
    """RBG_LAYOUT / RBG_RUN_DEPTHS / RBG_KMER_STEPS / RBG_HBM_BUDGET_MB / RBG_FTAB_K give the options of the same names
    their initial values (the command-line tools keep the reference's flags: include/rbg.h "Environment switches"); a value
    out of range is reported and ignored; rbg_set_default_option still overrides."""
    code = ("import sys; sys.path.insert(0, %r); from rowbowt_amd import capi\n"
            "g = capi.get_default_option\n"
            "print(g(capi.OPT_RANK_LAYOUT), g(capi.OPT_RUN_DEPTHS), g(capi.OPT_KMER_STEPS), g(capi.OPT_HBM_BUDGET_MB), g(capi.OPT_FTAB_K))\n"
            "capi.set_default_option(capi.OPT_RANK_LAYOUT, 1); print(g(capi.OPT_RANK_LAYOUT))\n" % ROOT)
    def run(**env):
        p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=120)
        assert p.returncode == 0, p.stderr[-2000:]
        return p.stdout.split(), p.stderr
    assert run()[0] == ["0", "0", "8", "0", "-1", "1"]
    out, _ = run(RBG_LAYOUT="runs", RBG_RUN_DEPTHS="0x11", RBG_KMER_STEPS="4", RBG_HBM_BUDGET_MB="50000", RBG_FTAB_K="10")
    assert out == ["2", "17", "4", "50000", "10", "1"]
    out, err = run(RBG_LAYOUT="sideways", RBG_RUN_DEPTHS="256", RBG_KMER_STEPS="9", RBG_FTAB_K="x")
    assert out == ["0", "0", "8", "0", "-1", "1"]
    assert all(("rbg: %s=" % k) in err and "ignored" in err for k in ("RBG_LAYOUT", "RBG_RUN_DEPTHS", "RBG_KMER_STEPS", "RBG_FTAB_K"))
    assert run(RBG_LAYOUT="2")[0][0] == "2"


def test_cache_file_of_several_checksum_chunks(tmp_path):
    """a cache file longer than one checksum chunk (2^20 words = 8 MB; rbg_host.cpp read_flat / FlatWriter): 1.6 M runs
    with samples make 27 MB -- written, read back as the same index, and refused when a byte of the first, a middle or the
    last chunk is flipped, or when two chunks change places"""
    rng = np.random.default_rng(19)
    r = 1_600_000
    step = rng.integers(1, 4, size=r)
    step[0] = 0
    heads = np.frombuffer(b"ACGT", dtype=np.uint8)[np.cumsum(step) % 4].copy()
    lens = rng.integers(1, 40, size=r).astype(np.uint64)
    heads[r // 3], lens[r // 3] = 1, 1
    n = int(lens.sum())
    vals = (rng.permutation(n)[:2 * r].astype(np.uint64))
    path = tmp_path / "big.rbgpu"
    capi.convert_runs(heads, lens, vals[:r], vals[r:], out_path=str(path))
    blob = path.read_bytes()
    CH = 8 << 20
    assert blob[:8] == b"RBGPUIX2" and len(blob) > 2 * CH + 4096
    a = ra.RowBowt.from_cache(str(path), ra.LoadRbwtFlag.SA, device=capi.DEVICE_NONE)
    b = ra.RowBowt.from_runs(heads, lens, vals[:r], vals[r:], device=capi.DEVICE_NONE)
    _same_index(a, b, HOST_ARRAYS[:5])
    flip = lambda at: blob[:at] + bytes([blob[at] ^ 4]) + blob[at + 1:]
    assert len(blob) > 3 * CH + 8
    swapped = blob[:CH] + blob[2 * CH:3 * CH] + blob[CH:2 * CH] + blob[3 * CH:]
    for bad in (flip(4096), flip(CH + 12345), flip(len(blob) - 16), swapped):
        (tmp_path / "bad.rbgpu").write_bytes(bad)
        with pytest.raises(ra.RbgError) as e:
            ra.RowBowt.from_cache(str(tmp_path / "bad.rbgpu"), ra.LoadRbwtFlag.SA, device=capi.DEVICE_NONE)
        assert e.value.code == -2


def test_native_cache_from_reference_index(tmp_path, data_dir, small_host):
    """next-row f1: .rbwt/.tsa/.mab/.docs -> flat .rbgpu -> the same host index; rbg_load falls back to it"""
    import shutil
    pre = os.path.join(data_dir, "small.fa")
    for suf in (".rbwt", ".tsa", ".mab"):
        shutil.copy(pre + suf, tmp_path / ("idx" + suf))
    (tmp_path / "idx.docs").write_text("ref 0\nhap1 10010\nhap2 20020\n")
    ALL = ra.LoadRbwtFlag.SA | ra.LoadRbwtFlag.MA | ra.LoadRbwtFlag.DL
    cache = tmp_path / "only" / "idx.rbgpu"
    cache.parent.mkdir()
    capi.convert_index(str(tmp_path / "idx"), ALL, str(cache))
    ref = ra.load_rowbowt(str(tmp_path / "idx"), ALL, device=capi.DEVICE_NONE)
    a = ra.RowBowt.from_cache(str(cache), ALL, device=capi.DEVICE_NONE)
    _same_index(a, ref)
    assert a.resolve_offset(20306) == ref.resolve_offset(20306) == ("hap2", 286)
    # prefix with no .rbwt next to it: rbg_load picks the cache up (what rb_align does after rb_build)
    b = ra.load_rowbowt(str(tmp_path / "only" / "idx"), ALL, device=capi.DEVICE_NONE)
    _same_index(b, ref)
    # parts are selectable; a part the file lacks is a missing file
    c = ra.RowBowt.from_cache(str(cache), ra.LoadRbwtFlag.NONE, device=capi.DEVICE_NONE)
    assert not c.info().has_tsa and not c.info().has_markers and c.info().r == ref.info().r
    capi.convert_index(str(tmp_path / "idx"), ra.LoadRbwtFlag.NONE, str(tmp_path / "bare.rbgpu"))
    with pytest.raises(ra.RbgError) as e:
        ra.RowBowt.from_cache(str(tmp_path / "bare.rbgpu"), ra.LoadRbwtFlag.SA, device=capi.DEVICE_NONE)
    assert e.value.code == -1
    # the cache is smaller than the decoded arrays and a torn one is refused
    blob = cache.read_bytes()
    assert blob[:8] == b"RBGPUIX2" and len(blob) % 8 == 0
    for bad in (blob[:-8], blob[:1000], blob[:40] + bytes([blob[40] ^ 1]) + blob[41:], b"RBGPUIX2" + bytes(200), b"RBGPUIX1" + blob[8:],
                blob[:-9] + bytes([blob[-9] ^ 0x80]) + blob[-8:]):
        (tmp_path / "bad.rbgpu").write_bytes(bad)
        with pytest.raises(ra.RbgError) as e:
            ra.RowBowt.from_cache(str(tmp_path / "bad.rbgpu"), ra.LoadRbwtFlag.NONE, device=capi.DEVICE_NONE)
        assert e.value.code == -2
    # version 1 of the file (one checksum chain over all words; written until round 3) is still read
    M = (1 << 64) - 1

    def flat_sum(words):
        h = 0x9E3779B97F4A7C15
        for w in words:
            h = ((h ^ int(w)) * 0xFF51AFD7ED558CCD) & M
            h = ((h << 29) | (h >> 35)) & M
        return h
    body = np.frombuffer(b"RBGPUIX1" + blob[8:-8], dtype="<u8")
    (tmp_path / "v1.rbgpu").write_bytes(body.tobytes() + int(flat_sum(body)).to_bytes(8, "little"))
    _same_index(ra.RowBowt.from_cache(str(tmp_path / "v1.rbgpu"), ALL, device=capi.DEVICE_NONE), ref)
    # ... and version 2's last word is that sum over the sums of its chunks of 2^20 words
    words2 = np.frombuffer(blob[:-8], dtype="<u8")
    assert flat_sum([flat_sum(words2[i:i + (1 << 20)]) for i in range(0, len(words2), 1 << 20)]) == int.from_bytes(blob[-8:], "little")
    with pytest.raises(ra.RbgError) as e:
        ra.RowBowt.from_cache(str(tmp_path / "nope.rbgpu"), device=capi.DEVICE_NONE)
    assert e.value.code == -1


def test_rb_build_cli_raw_inputs(tmp_path):
    """rb_build on rb_build's raw inputs (.bwt/.ssa/.esa/.docs, rb_build.cpp:83-93) -> .rbgpu == building from runs"""
    import subprocess
    import naive
    from synth import SynthIndex
    S = SynthIndex(L=700, H=5, n_sites=15, seed=8)
    bwt = naive.bwt_from_sa(S.text, S.fm.sa).copy()
    (tmp_path / "x.bwt").write_bytes(bwt.tobytes().replace(b"\x01", b"\x00"))
    pairs = lambda y: np.stack([np.zeros(len(y), dtype=np.uint64), y.astype(np.uint64)], axis=1).tobytes()
    (tmp_path / "x.ssa").write_bytes(pairs(S.ssa))
    (tmp_path / "x.esa").write_bytes(pairs(S.esa))
    (tmp_path / "x.docs").write_text("a 0\nb 701\n")
    exe = os.path.join(ROOT, "rowbowt_amd", "rb_build")
    (tmp_path / "out").mkdir()
    p = subprocess.run([exe, "-s", "-l", "-o", str(tmp_path / "out" / "y"), str(tmp_path / "x")], capture_output=True, timeout=60)
    assert p.returncode == 0, p.stderr.decode()
    assert (tmp_path / "out" / "y.docs").read_text() == "a 0\nb 701\n"   # copied, rowbowt_io.hpp:73-80
    got = ra.load_rowbowt(str(tmp_path / "out" / "y"), ra.LoadRbwtFlag.SA | ra.LoadRbwtFlag.DL, device=capi.DEVICE_NONE)
    want = ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=capi.DEVICE_NONE)
    for which in HOST_ARRAYS[:5]:
        assert (got.host_array(which) == want.host_array(which)).all()
    assert got.resolve_offset(705) == ("b", 4)
    # without -s no toehold SA is stored; asking for it later is a missing file
    p = subprocess.run([exe, "-o", str(tmp_path / "out" / "z"), str(tmp_path / "x")], capture_output=True, timeout=60)
    assert p.returncode == 0
    with pytest.raises(ra.RbgError):
        ra.load_rowbowt(str(tmp_path / "out" / "z"), ra.LoadRbwtFlag.SA, device=capi.DEVICE_NONE)
    # error paths: missing input ("does not exist", rowbowt_io.hpp:23-26), no argument, raw .ma, fbb
    p = subprocess.run([exe, "-s", str(tmp_path / "nope")], capture_output=True, timeout=60)
    assert p.returncode == 1 and b"does not exist" in p.stderr
    p = subprocess.run([exe], capture_output=True, timeout=60)
    assert p.returncode == 1 and b"no argument provided" in p.stderr
    p = subprocess.run([exe, "-m", str(tmp_path / "x")], capture_output=True, timeout=60)
    assert p.returncode == 1 and b".mab" in p.stderr
    p = subprocess.run([exe, "-x", str(tmp_path / "x")], capture_output=True, timeout=60)
    assert p.returncode == 1
    # --from-index converts a reference-built index
    data = os.path.join(ROOT, "tests", "data", "small.fa")
    p = subprocess.run([exe, "--from-index", "-s", "-m", "-o", str(tmp_path / "out" / "small"), data], capture_output=True, timeout=60)
    assert p.returncode == 0, p.stderr.decode()
    a = ra.load_rowbowt(str(tmp_path / "out" / "small"), ra.LoadRbwtFlag.SA | ra.LoadRbwtFlag.MA, device=capi.DEVICE_NONE)
    b = ra.load_rowbowt(data, ra.LoadRbwtFlag.SA | ra.LoadRbwtFlag.MA, device=capi.DEVICE_NONE)
    _same_index(a, b)


def test_build_from_runs_rejects_inconsistent_input():
    """garbage in the caller's arrays is RBG_EARG, not a table indexed out of bounds later"""
    from synth import SynthIndex
    S = SynthIndex(L=300, H=3, n_sites=5, seed=4)
    ok = ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=capi.DEVICE_NONE)
    n = ok.info().n
    ok.close()
    cases = []
    bad = S.lens.copy(); bad[3] = 0; cases.append((S.heads, bad, S.ssa, S.esa))                    # empty run
    bad = S.heads.copy(); bad[5] = bad[4]; cases.append((bad, S.lens, S.ssa, S.esa))               # not maximal
    bad = S.ssa.copy(); bad[2] = n + 1; cases.append((S.heads, S.lens, bad, S.esa))                # sample beyond the text
    bad = S.esa.copy(); bad[7] = 2**63; cases.append((S.heads, S.lens, S.ssa, bad))
    bad = S.ssa.copy(); bad[1] = bad[0]; cases.append((S.heads, S.lens, bad, S.esa))               # two runs starting at one text position
    for heads, lens, ssa, esa in cases:
        with pytest.raises(ra.RbgError) as e:
            ra.RowBowt.from_runs(heads, lens, ssa, esa, device=capi.DEVICE_NONE)
        assert e.value.code == -4
    rng = np.random.default_rng(0)
    for _ in range(200):   # random garbage never crashes: either rejected or a (meaningless) index
        R = int(rng.integers(1, 40))
        heads = rng.integers(0, 6, R).astype(np.uint8)
        lens = rng.integers(0, 5, R).astype(np.uint64)
        ssa = rng.integers(0, 200, R).astype(np.uint64)
        esa = rng.integers(0, 200, R).astype(np.uint64)
        try:
            ra.RowBowt.from_runs(heads, lens, ssa, esa, device=capi.DEVICE_NONE).close()
        except ra.RbgError as e:
            assert e.code in (-2, -4)


def _compile_c_example(tmp_path):
    import subprocess
    exe = tmp_path / "abi_usage"
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_usage.c"), "-o", str(exe),
                           "-L", os.path.join(ROOT, "rowbowt_amd"), "-lrbg", "-Wl,-rpath," + os.path.join(ROOT, "rowbowt_amd")])
    return exe


def test_header_is_plain_c_and_links(tmp_path):
    """include/rbg.h is consumed by a C11 compiler with -pedantic -Werror, and the example of INTEGRATION.md
    section 2 links against librbg.so (it runs in the GPU suite)"""
    assert _compile_c_example(tmp_path).exists()


def test_no_run_indexed_kernel_spills_and_no_kernel_shifts_by_its_last_vgpr():
    """(a) No kernel of the run-indexed layout may use scratch: the one instantiation that spilled (the instrumented search at 8-byte
    positions, after the bucket records had raised its register need) faulted on the device (profiles/r04_fault_note.txt).
    (b) No kernel at all may take the amount of a 64-bit shift from the last VGPR of its allocation: on gfx950 such a shift reads its
    amount from elsewhere (profiles/r05_shift64_last_vgpr.md: k_lf_runs answered from other buckets' records, or faulted, depending on
    what had run before).  tools/check_isa.py cross-compiles every kernel file to gfx950 assembly (no GPU needed) and reads both off it."""
    import re
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_isa
    per_file = check_isa.scan(workers=4)
    assert set(per_file) >= {"k_runs.hip", "k_runs_seeds.hip", "k_search.hip", "k_locate.hip", "k_markers.hip", "k_build.hip", "k_compose.hip", "k_text.hip"}
    bad = [(f,) + h for f, ks in per_file.items() for h in check_isa.hazards(ks)]
    assert not bad, bad
    res = per_file["k_runs.hip"] + per_file["k_runs_seeds.hip"]
    names = subprocess.run(["c++filt"], input="\n".join(k[0] for k in res), capture_output=True, text=True).stdout.splitlines()
    v2 = [(n, k[2]) for n, k in zip(names, res) if re.search(r"(k_find_range_runs<|_runs<|runs2<)", n)]
    assert len(v2) >= 35, len(v2)   # (12 + 5 of k_runs.hip, 2 + 4 + 2 + 6 + 6 of k_runs_seeds.hip)
    spilled = [(n, s) for n, s in v2 if s]
    assert not spilled, spilled
    # the checker itself: a kernel of 72 VGPRs shifting by v71 is flagged, one shifting by v70 or with 73 VGPRs is not
    assert check_isa.hazards([("k", 72, 0, [71, 3])]) == [("k", 72, 1)] and not check_isa.hazards([("k", 72, 0, [70]), ("k", 73, 0, [71])])
    assert check_isa.hazards([("k", 70, 0, [71])]) == [("k", 70, 1)]   # (allocated in granules of eight: 70 -> 72)


def test_info_sized_and_layout_info_sizes(small_host):
    """rbg_info writes sizeof(rbg_info_t) of the header it was built with, so a caller built against another layout of the struct uses rbg_info_sized
    (fills at most out_bytes); rbg_layout_info refuses a size that is neither this ABI's nor a documented prefix of it (an ABI-2 caller would get shifted fields)."""
    import ctypes as C
    L = ra.lib()
    full = capi.Info()
    assert L.rbg_info(small_host.h, C.byref(full)) == 0
    # a smaller struct (an older client): only its bytes are written
    buf = (C.c_uint8 * (C.sizeof(capi.Info) + 16))(*([0xAB] * (C.sizeof(capi.Info) + 16)))
    part = 64
    assert L.rbg_info_sized(small_host.h, C.cast(buf, C.POINTER(capi.Info)), part) == 0
    assert bytes(buf[:part]) == bytes(C.string_at(C.byref(full), part)) and all(b == 0xAB for b in buf[part:part + 32])
    assert L.rbg_info_sized(small_host.h, C.cast(buf, C.POINTER(capi.Info)), C.sizeof(capi.Info) + 16) == 0      # a larger one: sizeof(rbg_info_t) bytes, no more
    assert bytes(buf[:C.sizeof(capi.Info)]) == bytes(C.string_at(C.byref(full), C.sizeof(capi.Info))) and all(b == 0xAB for b in buf[C.sizeof(capi.Info):])
    assert L.rbg_info_sized(small_host.h, C.cast(buf, C.POINTER(capi.Info)), 4) != 0
    li = capi.LayoutInfo()
    assert L.rbg_layout_info(small_host.h, C.byref(li), C.sizeof(li)) == 0
    assert L.rbg_layout_info(small_host.h, C.byref(li), C.sizeof(li) - 8) == 0            # the struct before its last field: a prefix at a field boundary
    abi2 = 8 * 4 + 3 * 5 * 8 + 6 * 8 + 2 * 5 * 8      # ABI 2: five-entry arrays
    assert L.rbg_layout_info(small_host.h, C.byref(li), abi2) != 0
