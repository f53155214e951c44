"""The command-line tools' FASTA/FASTQ reader (rowbowt_amd/csrc/fastx.hpp) against kseq_read's behaviour
(reference include/kseq.h:178-219, restated in kseq_model.py), CPU only: well-formed input, truncated
quality strings (-2: the failing record is never reported, rb_align.cpp:176) and a broken gzip stream (-3)."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from kseq_model import kseq_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dump(tmp_path_factory):
    exe = tmp_path_factory.mktemp("fx") / "fastx_dump"
    subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "fastx_dump.cpp"), "-o", str(exe), "-lz"])

    def run(path, batch=3):
        p = subprocess.run([str(exe), str(path), str(batch)], capture_output=True, timeout=60)
        assert p.returncode == 0, p.stdout[-300:]
        lines = p.stdout.split(b"\n")
        assert lines[-1] == b"" and lines[-2].startswith(b"rc=")
        recs = [tuple(l.split(b"\t", 1)) for l in lines[:-2]]
        return recs, int(lines[-2][3:])
    return run


CASES = {
    "fastq": b"@r1 c\nACGT\n+\nIIII\n@r2\nGGCC\nAA\n+r2\nIII\nIII\n",
    "fasta_multi": b"junk\n>a x y\nAC\nGT\n\nAA\n>b\n>c\nTT",
    "crlf": b"@q\r\nACGTA\r\n+\r\nIIIII\r\n>f\r\nAC\r\n",
    "trunc_qual_short": b"@r1\nACGT\n+\nIIII\n@r2\nACGT\n+\nII\n",
    "trunc_no_qual": b">a\nACGT\n@b\nAC\n+\n",
    "trunc_no_plus_newline": b"@a\nACGT\n+\nIIII\n@b\nAC\n+",
    "trunc_long_qual": b"@a\nACGT\n+\nIIIIII\n@b\nAC\n+\nII\n",
    "header_only_eof": b">x",
    "empty": b"",
    "no_header": b"ACGT\nACGT\n",
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_matches_kseq_model(dump, tmp_path, name):
    data = CASES[name]
    want, want_rc = kseq_model(data)
    for gz in (False, True):
        f = tmp_path / (name + (".gz" if gz else ".fx"))
        f.write_bytes(gzip.compress(data) if gz else data)
        for batch in (1, 3, 100):
            got, rc = dump(f, batch)
            assert rc == want_rc, (name, gz, batch, got)
            assert got == [(n, s) for n, s in want], (name, gz, batch)
    if name.startswith("trunc"):
        assert want_rc == -2


def test_random_records_and_truncations(dump, tmp_path):
    rng = np.random.default_rng(5)
    recs = []
    for i in range(40):
        seq = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), int(rng.integers(0, 90))))
        if rng.random() < 0.5:
            recs.append(b"@q%d some comment\n" % i + seq + b"\n+\n" + b"I" * len(seq) + b"\n")
        else:
            k = int(rng.integers(1, 4))
            recs.append(b">f%d\n" % i + b"\n".join(seq[j::k] for j in range(k)) + b"\n")
    blob = b"".join(recs)
    for cut in [len(blob)] + [int(c) for c in rng.integers(1, len(blob), 60)]:
        data = blob[:cut]
        f = tmp_path / "r.fx"
        f.write_bytes(data)
        want, want_rc = kseq_model(data)
        got, rc = dump(f, 4)
        assert (got, rc) == ([(n, s) for n, s in want], want_rc), cut


def test_broken_gzip_stream_is_minus_3(dump, tmp_path):
    """gzread fails in the middle of the stream: the records that were complete are reported, then -3
    ("ERROR: error reading stream", rb_align.cpp:186-188)"""
    rng = np.random.default_rng(6)
    recs = [(b"r%d" % i, bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 100))) for i in range(20000)]
    blob = b"".join(b"@" + n + b"\n" + s + b"\n+\n" + b"I" * 100 + b"\n" for n, s in recs)
    z = bytearray(gzip.compress(blob, 1))
    mid = len(z) // 2
    z[mid:mid + 64] = bytes(rng.integers(0, 256, 64, dtype=np.uint8))  # corrupt the deflate stream
    f = tmp_path / "broken.fq.gz"
    f.write_bytes(bytes(z))
    got, rc = dump(f, 1000)
    assert rc in (-3, -2)   # an error inside a quality string surfaces as -2 (kseq.h:214-217, see fastx.hpp)
    assert 0 < len(got) < len(recs)
    assert got[:5000] == recs[:5000]   # (zlib hands out some garbled bytes before it notices the damage)
    # a gzip file cut short: this zlib's gzread hands out the good bytes and then reports plain end of file
    # (kseq sees the same calls), so the run ends like a short file: complete records, then -1 or -2
    f2 = tmp_path / "cut.fq.gz"
    f2.write_bytes(gzip.compress(blob, 1)[:mid])
    got, rc = dump(f2, 1000)
    import zlib
    prefix = zlib.decompressobj(31).decompress(f2.read_bytes())   # the bytes a reader gets out of the cut file
    want, want_rc = kseq_model(prefix)
    assert 0 < len(got) < len(recs) and got[:-1] == recs[:len(got) - 1]
    assert (got, rc) == ([(n, s_) for n, s_ in want], want_rc) or rc == -3


# ---- rb_align's record scanner (fastx_index.hpp): the same contract, input left in place ---------------------------
@pytest.fixture(scope="module")
def scan(tmp_path_factory):
    exe = tmp_path_factory.mktemp("fxi") / "fastx_index_dump"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-pthread",
                           os.path.join(ROOT, "tests", "cpp", "fastx_index_dump.cpp"), "-o", str(exe)])

    def run(path, block, threads, minseg):
        p = subprocess.run([str(exe), str(path), str(block), str(threads), str(minseg)], capture_output=True, timeout=120)
        assert p.returncode == 0, (p.stdout[-300:], p.stderr[-600:])
        lines = p.stdout.split(b"\n")
        assert lines[-1] == b"" and lines[-2].startswith(b"rc="), lines[-3:]
        return [tuple(l.split(b"\t", 1)) for l in lines[:-2]], int(lines[-2][3:])
    return run


GEOMETRIES = [(1 << 20, 1, 1 << 20), (7, 1, 1), (64, 3, 1), (1000, 4, 16), (1 << 20, 8, 1), (333, 2, 100)]


@pytest.fixture(scope="module")
def cli_input(tmp_path_factory):
    """the tools' InputSource itself (cli_input.hpp): mapped plain files and the zlib path, windows of any size"""
    exe = tmp_path_factory.mktemp("cin") / "cli_input_dump"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-pthread",
                           os.path.join(ROOT, "tests", "cpp", "cli_input_dump.cpp"), "-o", str(exe), "-lz"])

    def run(path, window, threads, minseg=None, stats=False, mode=False):
        p = subprocess.run([str(exe), str(path), str(window), str(threads)] + ([str(minseg)] if minseg else []), capture_output=True, timeout=300)
        assert p.returncode == 0, (p.stdout[-300:], p.stderr[-600:])
        lines = p.stdout.split(b"\n")
        assert lines[-1] == b"" and lines[-2].startswith(b"rc="), lines[-3:]
        res = [tuple(l.split(b"\t", 1)) for l in lines[:-2]], int(lines[-2][3:])
        if stats or mode:   # "windows=<n> parallel=<n> mode=<m>" on stderr
            st = dict(kv.split(b"=") for kv in p.stderr.split(b"\n")[-2].split())
            if mode:
                return res + (int(st[b"windows"]), st[b"mode"].decode())
            return res + (int(st[b"windows"]), int(st[b"parallel"]))
        return res
    return run


@pytest.mark.parametrize("name", sorted(CASES))
def test_input_source_plain_and_gzip_match_kseq_model(cli_input, tmp_path, name):
    """what rb_align / rb_markers read: the same records and end code as kseq_read, from the mapped file and through
    zlib, whatever the window size (records longer than a window, windows of a few bytes, carry-over between them)"""
    import gzip
    data = CASES[name]
    want, want_rc = kseq_model(data)
    plain = tmp_path / (name + ".fx")
    plain.write_bytes(data)
    gz = tmp_path / (name + ".fx.gz")
    with gzip.open(gz, "wb") as f:
        f.write(data)
    for path in (plain, gz):
        for window, threads in ((1 << 20, 1), (5, 1), (64, 3), (1000, 4)):
            got, rc = cli_input(path, window, threads)
            assert (got, rc) == ([(n, s) for n, s in want], want_rc), (name, path.name, window, threads)


def bgzf_bytes(data, block=65280, level=6):
    """`data` as a BGZF file (htslib's blocked gzip): independent members of at most 64 KB, each with the extra subfield
    'BC' holding its compressed size - 1, closed by the empty end-of-file block"""
    import struct
    import zlib
    out = bytearray()
    for a in list(range(0, len(data), block)) + [None]:
        chunk = b"" if a is None else data[a:a + block]
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        payload = c.compress(chunk) + c.flush()
        bsize = 12 + 6 + len(payload) + 8
        out += b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
        out += payload + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))
    return bytes(out)


def test_input_source_bgzf_blocks_inflated_in_parallel(cli_input, tmp_path):
    """a BGZF file (bgzip / htslib output) is mapped and its blocks inflated by the worker threads: same records and end
    code as kseq_read over the decompressed bytes, whatever the window; gzip.open reads the same file (it IS gzip); a
    corrupted block ends the stream with -3 after the records that came before it, like a failing gzread"""
    import gzip
    rng = np.random.default_rng(33)
    seqs = [bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), int(rng.integers(1, 400)))) for _ in range(3000)]
    fq = b"".join(b"@r%d c\n" % i + s + b"\n+\n" + b"@" * len(s) + b"\n" for i, s in enumerate(seqs))
    fa = b"".join(b">r%d\n" % i + s + b"\n" for i, s in enumerate(seqs))
    for name, data in (("b.fq.gz", fq), ("b.fa.gz", fa), ("trunc.fq.gz", fq[:len(fq) - 57])):
        path = tmp_path / name
        path.write_bytes(bgzf_bytes(data, block=int(rng.integers(500, 65280))))
        assert gzip.open(path, "rb").read() == data
        want, want_rc = kseq_model(data)
        for window, threads, minseg in ((1 << 20, 4, 1000), (7000, 3, 300), (100, 2, 50), (1 << 20, 1, 1 << 20)):   # (1 thread: zlib's stream)
            got, rc = cli_input(path, window, threads, minseg)
            assert (got, rc) == ([(n, s) for n, s in want], want_rc), (name, window, threads)
    # corruption: flip a byte inside the payload of a block in the middle
    blob = bytearray(bgzf_bytes(fq, block=4000))
    pos, k = 0, 0
    while k < 40:
        pos += (blob[pos + 16] | (blob[pos + 17] << 8)) + 1
        k += 1
    blob[pos + 30] ^= 0x5A
    bad = tmp_path / "bad.fq.gz"
    bad.write_bytes(bytes(blob))
    want, _ = kseq_model(fq)
    for window, threads in ((1 << 20, 4), (9000, 3)):
        got, rc = cli_input(bad, window, threads, 500)
        # (the record the bad block cuts may come out shortened, as from kseq_read when gzread fails inside a sequence)
        assert rc == -3 and 0 < len(got) < len(want) and got[:-1] == [(n, s) for n, s in want][:len(got) - 1]
        assert len(got) >= 40 * 4000 // 900 - 30      # everything before the bad block was delivered


# ten bytes that pass even the strict test for a speculative member start (magic, deflate, no flags, mtime 0, XFL 0, OS 3 = Unix)
GZ_TRAP = b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03"


def _gzread_model(raw):
    """what zlib's gzread delivers (the reference reads through it: kseq.h over gzFile): member after member while the next bytes are a
    gzip header; anything else -- zero padding too -- is trailing garbage and ends the file"""
    import zlib
    data, at = b"", 0
    while raw[at:at + 2] == b"\x1f\x8b":
        d = zlib.decompressobj(31)
        data += d.decompress(raw[at:])
        at = len(raw) - len(d.unused_data)
    return data


def _trap_fastq(rng, n, clean):
    """FASTQ whose quality strings (from record `clean` on) hold the bytes of a gzip member header: they survive verbatim in stored
    (level 0) members and look like member starts to a scan of the file"""
    seqs = [bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), int(rng.integers(1, 300)))) for _ in range(n)]
    return b"".join(b"@r%d\n" % i + s + b"\n+\n" + ((GZ_TRAP * len(s))[:len(s)] if i >= clean else b"I" * len(s)) + b"\n" for i, s in enumerate(seqs))


def test_input_source_gzip_members_inflated_in_parallel(cli_input, tmp_path):
    """a gzip file of several members (`cat a.gz b.gz`; not BGZF) is mapped and its members are inflated by the worker threads,
    speculatively from every place that looks like a member's start and checked as a chain: same records and end code as kseq_read over
    what gzip.open reads from the same file -- with members of every size (long ones are streamed across windows), members stored
    uncompressed whose DATA holds gzip header bytes (false starts inside a member), zero padding between members, garbage after the
    last one (zlib stops there), and a corrupt member (-3 after the records before it)"""
    import gzip
    rng = np.random.default_rng(44)
    fq = _trap_fastq(rng, 4000, clean=700)   # (member 0 is free of traps: the check at open() passes and the parallel mode is entered)

    def members(data, sizes, levels, pad=b""):
        out, at = bytearray(), 0
        k = 0
        while at < len(data):
            n = sizes[k % len(sizes)]
            out += gzip.compress(data[at:at + n], compresslevel=levels[k % len(levels)]) + pad
            at += n
            k += 1
        return bytes(out), k
    for name, sizes, levels, pad, tail in (("m1.fq.gz", [50000], [6], b"", b""), ("m2.fq.gz", [1, 70000, 333, 20000], [0, 6, 1], b"", b""),
                                           ("m3.fq.gz", [9000], [0], b"\0" * 7, b""), ("m4.fq.gz", [30000, 100], [6, 0], b"", b"garbage after the last member"),
                                           ("m5.fq.gz", [3000, 400000, 50, 90000], [6, 0, 0, 1], b"", b"")):
        blob, nm = members(fq, sizes, levels, pad)
        path = tmp_path / name
        path.write_bytes(blob + tail)
        assert nm > 3
        data = _gzread_model(blob + tail)
        assert data == (fq if not pad else fq[:sizes[0]])
        want, want_rc = kseq_model(data)
        for window, threads, minseg in ((1 << 20, 4, 1000), (5000, 3, 300), (200, 2, 50), (1 << 20, 1, 1 << 20)):   # (1 thread: zlib's stream)
            got, rc, _, mode = cli_input(path, window, threads, minseg, mode=True)
            assert (got, rc) == ([(n, s) for n, s in want], want_rc), (name, window, threads)
            assert mode == ("stream" if threads == 1 or pad else "members"), (name, threads, mode)   # (padding: member 0 is followed by no header)
    # a corrupt member in the middle: -3, everything before it delivered
    blob, nm = members(fq, [20000], [6])
    blob = bytearray(blob)
    starts = [i for i in range(len(blob) - 3) if blob[i:i + 3] == b"\x1f\x8b\x08" and (i == 0 or True)]
    mid = starts[len(starts) // 2]
    blob[mid + 40] ^= 0x77
    bad = tmp_path / "badm.fq.gz"
    bad.write_bytes(bytes(blob))
    want, _ = kseq_model(fq)
    for window, threads in ((1 << 20, 4), (9000, 3)):
        got, rc, _, mode = cli_input(bad, window, threads, 500, mode=True)
        assert mode == "members"
        assert rc == -3 and 0 < len(got) < len(want) and got[:-1] == [(n, s) for n, s in want][:len(got) - 1]
        assert len(got) >= (len(starts) // 2 - 1) * 20000 // 330      # the members before the damaged one were delivered


def test_input_source_single_member_gzip_with_false_member_starts_stays_a_stream(cli_input, tmp_path):
    """ the bytes of a gzip member header turn up by chance inside deflate data -- about once per 134 MB -- so an ordinary
    single-member .fq.gz has `candidates`.  Such a file must stay zlib's single stream: windows of the size asked for (not the whole
    file as one window), and, cut short, every record before the cut followed by -3 -- what kseq over gzread delivers.  Here the false
    starts are planted (a stored member whose quality strings hold header bytes that pass even the strict test)."""
    import gzip
    rng = np.random.default_rng(45)
    fq = _trap_fastq(rng, 6000, clean=40)
    blob = gzip.compress(fq, compresslevel=0)
    assert blob.count(GZ_TRAP) > 1000
    path = tmp_path / "single.fq.gz"
    path.write_bytes(blob)
    want, want_rc = kseq_model(fq)
    for window, threads in ((100000, 4), (30000, 2), (100000, 1)):
        got, rc, windows, mode = cli_input(path, window, threads, 500, mode=True)
        assert mode == "stream"
        assert (got, rc) == ([(n, s) for n, s in want], want_rc)
        assert windows >= len(fq) // window          # never the whole file as one window
    # the same file cut short: gzread fails at the cut; the records before it are delivered, then -3
    cut = tmp_path / "cut.fq.gz"
    cut.write_bytes(blob[:len(blob) * 9 // 10])
    ref = cli_input(cut, 100000, 1, 500)             # one thread: zlib's stream, as in the base behaviour
    assert ref[1] in (-1, -2) and len(want) * 8 // 10 < len(ref[0]) < len(want)
    for window, threads in ((100000, 4), (30000, 3)):
        got, rc, _, mode = cli_input(cut, window, threads, 500, mode=True)
        assert mode == "stream" and (got, rc) == ref
    # a real two-member file whose SECOND member is cut short: members mode, the first member and the text before the cut, the same end code
    two = gzip.compress(fq[:200000], compresslevel=6) + gzip.compress(fq[200000:], compresslevel=0)[:-20000]
    p2 = tmp_path / "two_cut.fq.gz"
    p2.write_bytes(two)
    ref = cli_input(p2, 100000, 1, 500)
    assert ref[1] in (-1, -2) and len(want) * 8 // 10 < len(ref[0]) < len(want)
    for window, threads in ((100000, 4), (1 << 20, 3), (3000, 2)):
        got, rc, _, mode = cli_input(p2, window, threads, 500, mode=True)
        assert mode == "members" and (got, rc) == ref


def test_input_source_every_window_is_scanned_in_parallel(cli_input, tmp_path):
    """FASTA and FASTQ over many windows with several scanning threads, mapped and through zlib: same records as
    kseq_read, and EVERY window goes through the multi-threaded scan -- also the FASTA windows that begin right after an
    already consumed '>' (ScanState::last_char set), which used to fall back to one thread"""
    import gzip
    rng = np.random.default_rng(21)
    seqs = [bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), int(rng.integers(1, 300)))) for _ in range(4000)]
    fa = b"".join(b">r%d c\n" % i + b"\n".join(s[j:j + 60] for j in range(0, len(s), 60)) + b"\n" for i, s in enumerate(seqs))
    fq = b"".join(b"@r%d c\n" % i + s + b"\n+\n" + b"@" * len(s) + b"\n" for i, s in enumerate(seqs))
    for name, data in (("multi.fa", fa), ("multi.fq", fq)):
        want, want_rc = kseq_model(data)
        plain = tmp_path / name
        plain.write_bytes(data)
        gz = tmp_path / (name + ".gz")
        with gzip.open(gz, "wb") as f:
            f.write(data)
        for path in (plain, gz):
            for window, threads, minseg in ((60000, 4, 2000), (17000, 3, 500)):
                got, rc, windows, parallel = cli_input(path, window, threads, minseg, stats=True)
                assert (got, rc) == ([(n, s) for n, s in want], want_rc), (path.name, window, threads)
                assert windows >= 8, windows
                # the last window may be shorter than two segments; every other one must have taken the threaded path
                assert parallel >= windows - 1, (path.name, window, windows, parallel)


@pytest.mark.parametrize("name", sorted(CASES))
def test_scanner_matches_kseq_model(scan, tmp_path, name):
    data = CASES[name]
    want, want_rc = kseq_model(data)
    f = tmp_path / (name + ".fx")
    f.write_bytes(data)
    for block, threads, minseg in GEOMETRIES:
        got, rc = scan(f, block, threads, minseg)
        assert (got, rc) == ([(n, s) for n, s in want], want_rc), (name, block, threads, minseg)


def test_scanner_random_records_truncations_and_adversarial_lines(scan, tmp_path):
    """well-formed mixes, every kind of cut, and inputs built to fool a boundary guess: quality lines that start
    with '@' and look like headers, '>' and '@' inside sequences' neighbours, blank lines, CRLF"""
    rng = np.random.default_rng(8)
    recs = []
    for i in range(60):
        seq = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), int(rng.integers(0, 90))))
        kind = rng.random()
        if kind < 0.4:
            q = bytes(rng.choice(np.frombuffer(b"@+>I5#", dtype=np.uint8), len(seq)))   # qualities full of header characters
            recs.append(b"@q%d some comment\n" % i + seq + b"\n+\n" + q + b"\n")
        elif kind < 0.55:
            recs.append(b"@c%d\r\n" % i + seq + b"\r\n+\r\n" + b"I" * len(seq) + b"\r\n")
        elif kind < 0.7:   # a multi-line FASTQ record whose quality lines begin with '@'
            h = len(seq) // 2
            recs.append(b"@m%d\n" % i + seq[:h] + b"\n" + seq[h:] + b"\n+\n" + b"@" * h + b"\n" + b"@" * (len(seq) - h) + b"\n")
        else:
            k = int(rng.integers(1, 4))
            recs.append(b">f%d\n" % i + b"\n".join(seq[j::k] for j in range(k)) + b"\n\n")
    blob = b"".join(recs)
    cuts = [len(blob)] + [int(c) for c in rng.integers(1, len(blob), 50)]
    for cut in cuts:
        data = blob[:cut]
        f = tmp_path / "r.fx"
        f.write_bytes(data)
        want, want_rc = kseq_model(data)
        for block, threads, minseg in ((1 << 20, 4, 1), (97, 3, 8), (1 << 20, 1, 1 << 20)):
            got, rc = scan(f, block, threads, minseg)
            assert (got, rc) == ([(n, s) for n, s in want], want_rc), (cut, block, threads, minseg)


def test_scanner_large_wellformed_many_threads(scan, tmp_path):
    rng = np.random.default_rng(9)
    recs = [(b"read%d" % i, bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 100))) for i in range(30000)]
    blob = b"".join(b"@" + n + b" x\n" + s + b"\n+\n" + b"@" * 100 + b"\n" for n, s in recs)   # worst-case qualities: all '@'
    f = tmp_path / "big.fq"
    f.write_bytes(blob)
    for block, threads, minseg in ((1 << 30, 8, 1 << 16), (1 << 18, 4, 1 << 12)):
        got, rc = scan(f, block, threads, minseg)
        assert rc == -1 and got == recs
