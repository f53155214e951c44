"""GPU parity (through the C-ABI, bit-exact against the oracle; needs an MI355X): the command-line tools, the C++ shim, the C program, rb_align -s text made on the device."""
import json
import os
import re
import sys

import numpy as np
import pytest

import golden_values as G
import orc
import rowbowt_amd as ra
from rowbowt_amd.shard import shard_bounds
from rowbowt_amd import capi
from synth import SynthIndex
from gpu_common import *  # noqa: F401,F403  (helpers shared by the GPU parity files)

pytestmark = pytest.mark.gpu
MAXU = G.MAXU
ALL = ra.LoadRbwtFlag.SA | ra.LoadRbwtFlag.MA


from kseq_model import kseq_model as _kseq_model  # noqa: E402


def test_cli_count_stdout(data_dir, simple_reads):
    rc, out, err = _run_cli([os.path.join(data_dir, "small.fa"), os.path.join(data_dir, "simple_query.fq")])
    assert rc == 0, err
    names = ["r1.ref", "r1.sample0.0", "r2.ref", "r2.sample0.0", "r3.ref", "r3.sample0.0"]
    want = "".join(f"{n} ({lo},{hi}), count={hi - lo + 1}\n" for n, (lo, hi) in zip(names, G.SIMPLE_RANGES))
    assert out == want
    assert len(err.strip().splitlines()[-1].split()) == 2  # "<load_s> <query_s>", rb_align.cpp:192
    # empty ranges print the unsigned wrap of 0-1+1 (rb_align.cpp:122)
    rc, out, _ = _run_cli([os.path.join(data_dir, "small.fa"), os.path.join(data_dir, "error_query.fq")])
    lines = out.splitlines()
    assert rc == 0 and lines[0] == "r1.ref (1,0), count=0" and lines[2] == "r2.ref (27430,27432), count=3"


def test_cli_layout_from_the_environment(data_dir, tmp_path):
    """rb_align keeps the reference's flags; the library's load-time knobs reach it by environment (include/rbg.h):
    RBG_LAYOUT=runs answers from the run-indexed layout, all depths or depths 1 and 4 only -- the same text."""
    import shutil
    for suf in (".rbwt", ".tsa"):
        shutil.copy(os.path.join(data_dir, "small.fa" + suf), tmp_path / ("idx" + suf))
    (tmp_path / "idx.docs").write_text("ref 0\nhap1 10010\nhap2 20020\n")
    args = ["-s", str(tmp_path / "idx"), os.path.join(data_dir, "simple_query.fq")]
    rc0, out0, err0 = _run_cli(args, env={"RBG_VERBOSE": "1"})
    assert rc0 == 0 and "run-indexed layout" not in err0, err0
    for env in ({"RBG_LAYOUT": "runs"}, {"RBG_LAYOUT": "runs", "RBG_RUN_DEPTHS": "0x1F"}, {"RBG_LAYOUT": "runs", "RBG_RUN_DEPTHS": "9", "RBG_FTAB_K": "0"}):
        rc, out, err = _run_cli(args, env=dict(env, RBG_VERBOSE="1"))
        assert rc == 0 and out == out0 and "run-indexed layout" in err, err
        mask = {None: "0x8b", "0x1F": "0x1f", "9": "0x9"}[env.get("RBG_RUN_DEPTHS")]
        assert f"k-mer depths with run lists: mask {mask}" in err, err


def test_cli_locs_and_markers_stdout(data_dir, tmp_path, small, simple_reads):
    import gzip
    import shutil
    rb, o = small
    for suf in (".rbwt", ".tsa", ".mab"):
        shutil.copy(os.path.join(data_dir, "small.fa" + suf), tmp_path / ("idx" + suf))
    (tmp_path / "idx.docs").write_text("ref 0\nhap1 10010\nhap2 20020\n")  # no .docs is shipped
    o.set_docs(["ref", "hap1", "hap2"], [0, 10010, 20020])
    fq = tmp_path / "q.fq.gz"  # gz + true FASTQ syntax + descriptions after the name
    with gzip.open(fq, "wt") as f:
        for i, q in enumerate(simple_reads):
            f.write(f"@read{i} some description\n{q.decode()}\n+\n{'~' * len(q)}\n")
    rc, out, err = _run_cli(["-s", "-m", str(tmp_path / "idx"), str(fq)])
    assert rc == 0, err
    want = ""
    for i, q in enumerate(simple_reads):
        lo, hi, k = o.find_range_w_toehold(q)
        want += f"read{i} ({lo},{hi}), count={hi - lo + 1}\n\tlocs: "
        for l in o.locs_at(lo, hi, k):
            name, off = o.resolve_offset(l)
            want += f"{l}/{name}:{off} "
        want += "\n\tmarkers: "
        mk = o.markers_at(lo, hi)
        if not mk:
            want += "no markers (consider building the marker array with a larger window size)"
        for m_ in mk:
            want += f"{G.get_pos(m_)}/{G.get_allele(m_)} "
        want += "\n"
    assert out == want
    assert "20306/hap2:286 286/ref:286" in out
    # (-s -m is made on the device too, markers line included; the host formatter gives the same bytes)
    rc, out_h, _ = _run_cli(["-s", "-m", str(tmp_path / "idx"), str(fq)], env={"RB_ALIGN_HOST_TEXT": "1"})
    assert rc == 0 and out_h == want
    # batching is invisible: one read per GPU batch gives the same bytes
    rc, out1, _ = _run_cli(["-s", "-m", "--batch", "1", str(tmp_path / "idx"), str(fq)])
    assert rc == 0 and out1 == want
    # replicas are invisible too: every batch sharded over three replicas (on this box's one GPU the same device
    # three times; `--gpus G` puts them on G devices), with batches smaller than, equal to and larger than the shards
    for extra in (["--devices", "0,0,0"], ["--devices", "0,0", "--batch", "3"], ["--gpus", "1", "--batch", "2"]):
        rc, outg, err = _run_cli(["-s", "-m"] + extra + [str(tmp_path / "idx"), str(fq)])
        assert rc == 0 and outg == want, err
    # a truncated record ends the run like kseq's -2 without being reported; the reads before it are
    # (rb_align.cpp:176-185: the loop stops at the failing kseq_read)
    part = tmp_path / "part.fq"
    part.write_text("".join(f"@read{i}\n{q.decode()}\n+\n{'~' * len(q)}\n" for i, q in enumerate(simple_reads[:3])) + "@bad\nACGT\n+\n~~\n")
    rc, outp, err = _run_cli([str(tmp_path / "idx"), str(part)])
    assert rc == 1 and "truncated quality string" in err
    assert outp.count("\n") == 3 and "bad" not in outp and outp.startswith("read0 ")
    # missing index -> "bad file", exit(1) (rowbowt_io.hpp:166-169)
    rc, _, err = _run_cli([str(tmp_path / "nope"), str(fq)])
    assert rc == 1 and "bad file" in err
    # truncated quality string -> error like kseq's -2 (rb_align.cpp:183-185)
    bad = tmp_path / "bad.fq"
    bad.write_text("@r\nACGT\n+\n~~\n")
    rc, _, err = _run_cli([str(tmp_path / "idx"), str(bad)])
    assert rc == 1 and "truncated quality string" in err


def test_cli_locs_text_made_on_the_device(data_dir, tmp_path, small, simple_reads, error_reads, synth):
    """`rb_align -s` (no -m): the text comes from rbg_align_text -- locs_at, resolve_offset and the decimals on the device
    (k_text.hip) -- and is byte-identical to the oracle's rendering of rb_report (rb_align.cpp:118-139) and to the host
    formatter (RB_ALIGN_HOST_TEXT=1): reads without a match, names of 1 and of 700 characters (beyond what a workgroup
    stages in LDS), descriptions, one read per batch, three replicas, and a synthetic pangenome whose reads have tens of
    locations in 50 documents, in batches that do not divide the input."""
    import shutil
    rb, o = small
    for suf in (".rbwt", ".tsa"):
        shutil.copy(os.path.join(data_dir, "small.fa" + suf), tmp_path / ("idx" + suf))
    long_doc = "hap2_" + "x" * 90
    (tmp_path / "idx.docs").write_text(f"ref 0\nhap1 10010\n{long_doc} 20020\n")
    o.set_docs(["ref", "hap1", long_doc], [0, 10010, 20020])
    reads = list(simple_reads) + list(error_reads) + [b"ACGT", b"A", simple_reads[0][:30]]
    names = [f"read{i}" for i in range(len(reads))]
    names[1] = "r"
    names[2] = "n" * 700
    names[4] = "q" * 300
    fq = tmp_path / "q.fq"
    fq.write_text("".join(f"@{n} desc {i}\n{q.decode()}\n+\n{'~' * len(q)}\n" for i, (n, q) in enumerate(zip(names, reads))))
    want = ""
    for n, q in zip(names, reads):
        lo, hi, k = o.find_range_w_toehold(q)
        want += f"{n} ({lo},{hi}), count={(hi - lo + 1) % 2**64}\n\tlocs: "
        if lo <= hi:
            for l in o.locs_at(lo, hi, k):
                dn, off = o.resolve_offset(l)
                want += f"{l}/{dn}:{off} "
        want += "\n"
    for extra in ([], ["--batch", "1"], ["--devices", "0,0,0"], ["--devices", "0,0", "--batch", "3"]):
        rc, out, err = _run_cli(["-s"] + extra + [str(tmp_path / "idx"), str(fq)])
        assert rc == 0 and out == want, err
    rc, out_h, err = _run_cli(["-s", str(tmp_path / "idx"), str(fq)], env={"RB_ALIGN_HOST_TEXT": "1"})
    assert rc == 0 and out_h == want, err
    # a pangenome: many locations per read, 50 documents
    S = synth
    unit = len(S.text) // 50
    docs = "".join(f"hap{h} {h * unit}\n" for h in range(50))
    capi.convert_runs(S.heads, S.lens, S.ssa, S.esa, out_path=str(tmp_path / "pg.rbgpu"), docs_text=docs)
    rs = S.sample_reads(5000, 70, seed=31, sub_rate=0.1, ragged=True)
    fq2 = tmp_path / "pg.fq"
    fq2.write_text("".join(f"@pg.{i}/{i % 7}\n{q.decode()}\n+\n{'I' * len(q)}\n" for i, q in enumerate(rs) if len(q)))
    outs = []
    for extra, env in (([], None), (["--batch", "700", "--devices", "0,0"], None), ([], {"RB_ALIGN_HOST_TEXT": "1"})):
        rc, out, err = _run_cli(["-s"] + extra + [str(tmp_path / "pg"), str(fq2)], env=env)
        assert rc == 0, err
        outs.append(out)
    assert outs[0] == outs[2] and outs[1] == outs[2] and outs[0].count("\n") == 2 * sum(1 for q in rs if len(q))
    assert len(outs[0]) > 200 * len(rs)


def test_align_text_markers_line(small, simple_reads, error_reads):
    """RBG_TEXT_MARKERS through the ABI on the reference's fixture: with and without the locations' line, reads with and
    without markers and without a match (rb_align.cpp:118-145)"""
    rb, o = small
    o.set_docs(["ref", "hap1", "hap2"], [0, 10010, 20020])
    rb.set_docs(["ref", "hap1", "hap2"], [0, 10010, 20020])
    reads = list(simple_reads) + list(error_reads) + [b"ACGT", simple_reads[0][:25]]
    seqs, off = ra.pack_reads(reads)
    lo, hi, k = rb.find_range_w_toehold(seqs, off)
    names = [f"q{i}".encode() for i in range(len(reads))]
    for with_locs in (True, False):
        got = rb.align_text(lo, hi, k if with_locs else None, names, markers=True).decode()
        want = ""
        for n, a, b, kk in zip(names, lo, hi, k):
            a, b, kk = int(a), int(b), int(kk)
            want += f"{n.decode()} ({a},{b}), count={(b - a + 1) % 2**64}\n"
            if with_locs:
                want += "\tlocs: "
                if a <= b:
                    for l in o.locs_at(a, b, kk):
                        dn, offs = o.resolve_offset(l)
                        want += f"{l}/{dn}:{offs} "
                want += "\n"
            want += "\tmarkers: "
            mk = o.markers_at(a, b) if a <= b else []
            if not mk:
                want += "no markers (consider building the marker array with a larger window size)"
            for m_ in mk:
                want += f"{G.get_pos(m_)}/{G.get_allele(m_)} "
            want += "\n"
        assert got == want


def test_align_text_through_the_abi(synth):
    """rbg_align_text called directly: max_hits caps the locations per read like locs_at's (rowbowt.hpp:613-621); without the
    document list the call says RBG_ENOTLOADED; texts of several calls may be out at once"""
    S = synth
    rb = ra.RowBowt.from_runs(S.heads, S.lens, S.ssa, S.esa, device=0)
    o = orc.Oracle.from_runs(S.heads, S.lens, S.ssa, S.esa)
    reads = S.sample_reads(300, 50, seed=3, sub_rate=0.1)
    seqs, off = ra.pack_reads(reads)
    lo, hi, k = rb.find_range_w_toehold(seqs, off)
    names = [f"r{i}".encode() for i in range(len(reads))]
    with pytest.raises(ra.RbgError):
        rb.align_text(lo, hi, k, names)
    # k = NULL: the report without -s (rb_align.cpp:120-122), no document list needed; empty ranges print count=0
    got = rb.align_text(lo, hi, None, names)
    assert got.decode() == "".join(f"{n.decode()} ({int(a)},{int(b)}), count={(int(b) - int(a) + 1) % 2**64}\n" for n, a, b in zip(names, lo, hi))
    unit = len(S.text) // 4
    starts = [0, unit, 2 * unit, 3 * unit]
    rb.set_docs([f"d{j}" for j in range(4)], starts)
    o.set_docs([f"d{j}" for j in range(4)], starts)
    for max_hits in (MAXU, 3, 1, 0):
        got = rb.align_text(lo, hi, k, names, max_hits)
        want = ""
        for n, a, b, kk in zip(names, lo, hi, k):
            a, b, kk = int(a), int(b), int(kk)
            want += f"{n.decode()} ({a},{b}), count={(b - a + 1) % 2**64}\n\tlocs: "
            if a <= b:
                for l in o.locs_at(a, b, kk, max_hits):
                    dn, offs = o.resolve_offset(l)
                    want += f"{l}/{dn}:{offs} "
            want += "\n"
        assert got.decode() == want
    rb.close()
    o.close()


def test_cli_rb_markers_stdout(data_dir, tmp_path, small):
    import rb_markers_model as RM
    rb, o = small
    idx = os.path.join(data_dir, "small.fa")
    text = open(idx, "rb").read().split(b"\n", 1)[1].replace(b"\n", b"")
    rng = np.random.default_rng(77)
    recs = []
    for fn in ("simple_query.fq", "error_query.fq"):
        names, seqs = orc.read_fastx(os.path.join(data_dir, fn))
        recs += list(zip(names, seqs))
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    for i in range(300):   # 101 bp reads from either strand, some with errors, lower case and Ns
        p = int(rng.integers(0, len(text) - 101))
        q = bytearray(text[p:p + 101])
        if i % 2:
            q = bytearray(bytes(q).translate(comp)[::-1])
        for _ in range(int(rng.integers(0, 3))):
            q[int(rng.integers(0, 101))] = b"ACGTN"[int(rng.integers(0, 5))]
        if i % 7 == 0:
            q = bytearray(bytes(q).lower())
        recs.append((f"syn{i}".encode(), bytes(q)))
    recs.append((b"short", b"ACG"))
    recs.append((b"empty", b""))
    fq = tmp_path / "reads.fq"
    with open(fq, "wb") as f:
        for name, seq in recs:
            f.write(b"@" + name + b" x\n" + seq + b"\n+\n" + b"I" * len(seq) + b"\n")
    rc, out, err = _run_rb_markers([idx, str(fq)])
    assert rc == 0, err
    want = RM.expected_stdout(o, recs)
    assert out == want
    assert " + 0 20 0/289/0\n" in out and out.count("\n") > 2 * len(recs) - 10
    assert "loading rowbowt + markers took" in err and "counting markers took" in err
    for args, kw in ((["--wsize", "10", "--max-range", "3", "--min-range", "2"], dict(wsize=10, max_range=3, min_range=2)),
                     (["-w", "5", "--batch", "7", "--threads", "3"], dict(wsize=5)),
                     (["--heuristic"], dict(heuristic=True)),
                     (["--heuristic", "--best-strand-only", "--min-seed-length", "30", "--read-len", "101"],
                      dict(heuristic=True, best_strand=True, min_seed_len=30, read_len=101)),
                     (["--heuristic", "-y", "25", "--clear-conflicting", "--clear-identical", "-l", "50", "-w", "8"],
                      dict(heuristic=True, min_seed_len=25, clear_conflicting=True, clear_identical=True, read_len=50, wsize=8))):
        rc, out, err = _run_rb_markers(args + [idx, str(fq)])
        assert rc == 0, err
        assert out == RM.expected_stdout(o, recs, **kw), args
    # --ftab: the index prefix needs its .ftab (rb_build -f); seeds then go through search_ftab
    import shutil
    for suf in (".rbwt", ".mab"):
        shutil.copy(idx + suf, tmp_path / ("fx" + suf))
    rc, _, err = _run_rb_markers(["--ftab", str(tmp_path / "fx"), str(fq)])
    assert rc == 1 and "bad file" in err                      # no .ftab yet (rowbowt_io.hpp:166-169)
    rb.write_ftab(6, str(tmp_path / "fx.ftab"))
    long_recs = [r for r in recs if len(r[1]) >= 6]
    fq2 = tmp_path / "long.fq"
    with open(fq2, "wb") as f:
        for name, seq in long_recs:
            f.write(b"@" + name + b"\n" + seq + b"\n+\n" + b"I" * len(seq) + b"\n")
    rc, out, err = _run_rb_markers(["--ftab", "-w", "8", str(tmp_path / "fx"), str(fq2)])
    assert rc == 0, err
    assert out == RM.expected_stdout(o, long_recs, wsize=8, ftab_k=6)
    assert out != RM.expected_stdout(o, long_recs, wsize=8)
    rc, out, err = _run_rb_markers(["-f", "--heuristic", "--best-strand-only", "-y", "20", str(tmp_path / "fx"), str(fq2)])
    assert rc == 0 and out == RM.expected_stdout(o, long_recs, heuristic=True, best_strand=True, min_seed_len=20, ftab_k=6)
    rc, _, err = _run_rb_markers(["--ftab", "-w", "4", str(tmp_path / "fx"), str(fq2)])
    assert rc == 1 and "wsize cannot be greater" in err       # rowbowt.hpp:423-426 (k - 1 > wsize)
    rc, _, err = _run_rb_markers(["--ftab", str(tmp_path / "fx"), str(fq)])
    assert rc == 1 and "shorter than the ftab" in err         # the reference dies in substr (rowbowt.hpp:431)
    text_ftab = (tmp_path / "fx.ftab").read_text().splitlines()
    (tmp_path / "fx.ftab").write_text("\n".join(text_ftab[:-1] + [text_ftab[-1].rsplit(" ", 1)[0] + " 0"]) + "\n")
    rc, _, err = _run_rb_markers(["--ftab", str(tmp_path / "fx"), str(fq2)])
    assert rc == 1 and "ftab" in err                          # a table that is not this index's is refused
    # modes the reference itself refuses or that this engine does not build
    for flag in ("--overlap", "--lmem", "--fbb"):
        rc, _, err = _run_rb_markers([flag, idx, str(fq)])
        assert rc == 1 and err
    rc, _, err = _run_rb_markers([idx])
    assert rc == 1 and "no argument provided" in err
    rc, _, err = _run_rb_markers([str(tmp_path / "nope"), str(fq)])
    assert rc == 1 and "bad file" in err


def test_cpp_shim_reference_goldens(tmp_path, data_dir):
    """The reference's own test assertions, issued through the C++ shim with the reference's signatures."""
    import shutil
    import subprocess
    from test_capi_host import _compile_shim_test
    exe = _compile_shim_test(tmp_path)
    for suf in (".rbwt", ".tsa"):
        shutil.copy(os.path.join(data_dir, "small.fa" + suf), tmp_path / ("idx" + suf))
    (tmp_path / "idx.docs").write_text("ref 0\nhap1 10010\nhap2 20020\n")
    p = subprocess.run([str(exe), data_dir, str(tmp_path / "idx")], capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert b"shim goldens ok" in p.stdout


def test_cpp_threaded_one_read_caller(tmp_path, data_dir):
    """tests/cpp/shim_threads.cpp: a thread pool calling the reference's one-query methods through the shim, unmodified
    (rb_markers.cpp:318-535's shape); answers equal the batch forms, and the library's micro-batching queue serves the
    calls with fewer launches than calls (printed: rate with and without it)"""
    import subprocess
    from test_capi_host import ROOT
    exe = tmp_path / "shim_threads"
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "rowbowt_amd", "include"),
                           os.path.join(ROOT, "tests", "cpp", "shim_threads.cpp"), "-o", str(exe),
                           "-L", os.path.join(ROOT, "rowbowt_amd"), "-lrbg", "-Wl,-rpath," + os.path.join(ROOT, "rowbowt_amd")])
    prefix, fasta = os.path.join(data_dir, "small.fa"), os.path.join(data_dir, "small.fa")
    for combine, threads in (("1", 16), ("0", 16), ("1", 1)):
        env = dict(os.environ, RBG_HOST_COMBINE=combine)
        p = subprocess.run([str(exe), prefix, fasta, str(threads), "4000"], capture_output=True, timeout=600, env=env)
        assert p.returncode == 0 and b"shim threads ok" in p.stdout, p.stdout.decode()[-500:] + p.stderr.decode()[-1500:]
        print(f"RBG_HOST_COMBINE={combine}:", p.stdout.decode().splitlines()[0])
        if combine == "1" and threads == 16:
            m = re.search(r"(\d+) one-read calls in (\d+) launches", p.stdout.decode())
            assert m and int(m.group(1)) == 9000 and int(m.group(2)) < int(m.group(1))


def test_c_abi_example_program(tmp_path, data_dir):
    """tests/c/abi_usage.c (plain C11 over include/rbg.h): the reference's golden values through the C-ABI"""
    import subprocess
    from test_capi_host import _compile_c_example
    exe = _compile_c_example(tmp_path)
    p = subprocess.run([str(exe), os.path.join(data_dir, "small.fa")], capture_output=True, timeout=120)
    assert p.returncode == 0 and b"abi_usage ok" in p.stdout, p.stderr.decode()


def test_cli_parser_matches_kseq(data_dir, tmp_path, small):
    rb, o = small
    t = open(os.path.join(data_dir, "small.fa"), "rb").read().split(b"\n", 1)[1].replace(b"\n", b"")
    s1, s2, s3 = t[100:160], t[500:530], t[900:1000]
    blob = (b"junk before the first header\n>multi line\tcomment here\n" + s1[:20] + b"\n" + s1[20:45] + b"\n\n" + s1[45:] + b"\n"
            b"@fq1 desc\r\n" + s2 + b"\r\n+\r\n" + b"I" * len(s2) + b"\r\n"
            b">with space in seq\n" + s3[:10] + b" " + s3[10:] + b"\n"
            b"@fq2\n" + s3[:50] + b"\n" + s3[50:] + b"\n+fq2\n" + b">" * 50 + b"\n" + b"@" * 50 + b"\n"
            b">empty\n>last_no_newline\n" + s1)
    fq = tmp_path / "weird.fx"
    fq.write_bytes(blob)
    recs, err = _kseq_model(blob)
    assert err == -1 and [r[0] for r in recs] == [b"multi", b"fq1", b"with", b"fq2", b"empty", b"last_no_newline"]
    assert recs[0][1] == s1 and recs[1][1] == s2 and recs[3][1] == s3 and recs[4][1] == b""
    rc, out, errtxt = _run_cli([os.path.join(data_dir, "small.fa"), str(fq)])
    assert rc == 0, errtxt
    want = ""
    for name, seq in recs:
        lo, hi = o.find_range(seq)
        want += f"{name.decode()} ({lo},{hi}), count={(hi - lo + 1) % 2**64}\n"
    assert out == want
    assert "(1,0), count=0" in out.splitlines()[2]  # the blank inside the sequence is kept, as kseq does


@pytest.mark.parametrize("fmt", ["fasta", "fastq"])
@pytest.mark.parametrize("gz", [False, True])
def test_cli_many_windows(data_dir, tmp_path, small, fmt, gz):
    """an input several windows long (rb_align --window-mb 1; more than 3 MB of records), plain (memory-mapped) and gzip (zlib,
    with the unfinished record carried from window to window): every record answered once, in order, as
    rb_align.cpp:176-191 prints it.  FASTA is the case where a window ends right after the next record's '>' has been
    consumed (kseq.h:195-199), which the zlib path once mishandled."""
    import gzip
    rb, o = small
    t = open(os.path.join(data_dir, "small.fa"), "rb").read().split(b"\n", 1)[1].replace(b"\n", b"")
    rng = np.random.default_rng(12)
    recs, blob = [], bytearray()
    for i in range(36000):
        a, m = int(rng.integers(0, len(t) - 160)), int(rng.integers(20, 150))
        seq = bytearray(t[a:a + m])
        if rng.random() < 0.2:
            seq[int(rng.integers(m))] = ord("ACGT"[int(rng.integers(4))])
        seq = bytes(seq)
        recs.append((b"r%d" % i, seq))
        if fmt == "fasta":
            blob += b">r%d some text\n" % i + seq[:60] + b"\n" + (seq[60:] + b"\n" if len(seq) > 60 else b"")
        else:
            blob += b"@r%d\n" % i + seq + b"\n+\n" + b"I" * len(seq) + b"\n"
    assert len(blob) > (3 << 20)
    path = tmp_path / ("reads." + fmt + (".gz" if gz else ""))
    if gz:
        with gzip.open(path, "wb") as f:
            f.write(bytes(blob))
    else:
        path.write_bytes(bytes(blob))
    rc, out, errtxt = _run_cli(["--window-mb", "1", os.path.join(data_dir, "small.fa"), str(path)])
    assert rc == 0, errtxt
    seqs, off = ra.pack_reads([r[1] for r in recs])
    wlo, whi = o.find_range_batch(seqs, off) if hasattr(o, "find_range_batch") else o.find_range_w_toehold_batch(seqs, off)[:2]
    lines = out.splitlines()
    assert len(lines) == len(recs)
    for i in (0, 1, 5000, 11000, 35999):
        assert lines[i] == f"r{i} ({int(wlo[i])},{int(whi[i])}), count={(int(whi[i]) - int(wlo[i]) + 1) % 2**64}"
    want = "".join(f"r{i} ({int(wlo[i])},{int(whi[i])}), count={(int(whi[i]) - int(wlo[i]) + 1) % 2**64}\n" for i in range(len(recs)))
    assert out == want
