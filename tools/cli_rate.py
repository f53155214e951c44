#!/usr/bin/env python3
"""End-to-end rate of the rb_align-compatible CLI (scan + GPU + text output) on the toy index with a large
synthetic plain FASTQ in the page cache.  GPU box only.  usage: cli_rate.py [reads = 10000000] [markers_reads = 2000000]"""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fa = b"".join(open(os.path.join(ROOT, "tests/data/small.fa"), "rb").read().split(b"\n")[1:])
rng = np.random.default_rng(1)
N, m = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, 100
NM = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000


def write_fastq(path, n):
    """'@r%08d\\n' + 100 bases + '\\n+\\n' + 100 x '~' + '\\n' per record, built as one byte matrix"""
    arr = np.frombuffer(fa, dtype=np.uint8)
    rec = np.empty((n, 2 + 8 + 1 + m + 3 + m + 1), dtype=np.uint8)
    rec[:, 0], rec[:, 1] = ord("@"), ord("r")
    ids = np.arange(n)
    for d in range(8):
        rec[:, 2 + 7 - d] = ord("0") + (ids // 10**d) % 10
    rec[:, 10] = 10
    starts = rng.integers(0, len(fa) - m, n)
    for a in range(0, n, 1 << 20):
        b = min(n, a + (1 << 20))
        rec[a:b, 11:11 + m] = arr[starts[a:b, None] + np.arange(m)[None, :]]
    rec[:, 11 + m], rec[:, 12 + m], rec[:, 13 + m] = 10, ord("+"), 10
    rec[:, 14 + m:14 + 2 * m] = ord("~")
    rec[:, 14 + 2 * m] = 10
    rec.tofile(path)


path = "/tmp/cli_reads.fq"
t0 = time.perf_counter()
write_fastq(path, N)
print(f"wrote {path} in {time.perf_counter() - t0:.1f} s", flush=True)
os.makedirs("/tmp/cli_idx", exist_ok=True)
for suf in (".rbwt", ".tsa", ".mab"):
    subprocess.check_call(["cp", os.path.join(ROOT, "tests/data/small.fa" + suf), "/tmp/cli_idx/idx" + suf])
open("/tmp/cli_idx/idx.docs", "w").write("ref 0\nhap1 10010\nhap2 20020\n")
exe = os.path.join(ROOT, "rowbowt_amd", "rb_align")
subprocess.run(["cat", path], stdout=subprocess.DEVNULL)   # page cache
print(f"plain FASTQ, {N} x {m} bp, {os.path.getsize(path) / 1e6:.0f} MB in the page cache; toy index (tests/data/small.fa); "
      f"{os.cpu_count()} logical CPUs")
for flags, out in (([], "/dev/null"), (["--threads", "16"], "/dev/null"), (["--threads", "32"], "/dev/null"), (["--threads", "16"], "/tmp/cli_out.txt"),
                   (["-s", "--threads", "16"], "/tmp/cli_out.txt"), (["-s", "-m", "--threads", "16"], "/tmp/cli_out.txt")):
    t0 = time.perf_counter()
    try:
        p = subprocess.run([exe] + flags + ["/tmp/cli_idx/idx", path], stdout=open(out, "wb"), stderr=subprocess.PIPE, timeout=180)
    except subprocess.TimeoutExpired:
        print(f"rb_align {' '.join(flags)} -> {out}: TIMEOUT after 180 s", flush=True)
        continue
    dt = time.perf_counter() - t0
    if p.returncode != 0:
        print(f"rb_align {' '.join(flags)} -> {out}: exit {p.returncode}: {p.stderr.decode()[-300:]}", flush=True)
        continue
    load_s, query_s = (float(x) for x in p.stderr.decode().strip().splitlines()[-1].split())
    sz = os.path.getsize(out) if out != "/dev/null" else 0
    print(f"rb_align {' '.join(flags) or '(count, default threads)':24s} -> {out:16s}: process {dt:.2f} s (index load {load_s:.2f} s); query loop {query_s:.3f} s = "
          f"{N / query_s:.3e} reads/s" + (f"; {sz / 1e6:.0f} MB of text" if sz else ""), flush=True)

if NM:
    pathm = "/tmp/cli_reads_m.fq"
    write_fastq(pathm, NM)
    exe2 = os.path.join(ROOT, "rowbowt_amd", "rb_markers")
    for flags in (["--threads", "16"], ["--heuristic", "--best-strand-only", "--min-seed-length", "30", "--threads", "16"]):
        t0 = time.perf_counter()
        p = subprocess.run([exe2] + flags + ["/tmp/cli_idx/idx", pathm], stdout=open("/tmp/cli_out.txt", "wb"), stderr=subprocess.PIPE, timeout=300)
        dt = time.perf_counter() - t0
        sz = os.path.getsize("/tmp/cli_out.txt")
        last = p.stderr.decode().strip().splitlines()[-1]
        try:
            loop_s = float(last.split("took:")[1].split()[0])
            loop = f"query loop {loop_s:.3f} s = {NM / loop_s:.3e} reads/s"
        except Exception:  # noqa: BLE001
            loop = last
        print(f"rb_markers {' '.join(flags)}: {NM} x {m} bp FASTQ -> {sz / 1e6:.0f} MB of text; process {dt:.2f} s = {NM / dt:.3e} reads/s; {loop}")
