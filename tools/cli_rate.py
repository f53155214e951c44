#!/usr/bin/env python3
"""End-to-end rate of the rb_align-compatible CLI (parse + GPU + text output) on the toy index with a large
synthetic read file.  GPU box only."""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fa = b"".join(open(os.path.join(ROOT, "tests/data/small.fa"), "rb").read().split(b"\n")[1:])
rng = np.random.default_rng(1)
N, m = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000, 100
starts = rng.integers(0, len(fa) - m, N)
arr = np.frombuffer(fa, dtype=np.uint8)
idx = starts[:, None] + np.arange(m)[None, :]
reads = arr[idx]
path = "/tmp/cli_reads.fq"
with open(path, "wb") as f:
    qual = b"~" * m
    for i in range(N):
        f.write(b"@r%d\n" % i + reads[i].tobytes() + b"\n+\n" + qual + b"\n")
os.makedirs("/tmp/cli_idx", exist_ok=True)
for suf in (".rbwt", ".tsa", ".mab"):
    subprocess.check_call(["cp", os.path.join(ROOT, "tests/data/small.fa" + suf), "/tmp/cli_idx/idx" + suf])
open("/tmp/cli_idx/idx.docs", "w").write("ref 0\nhap1 10010\nhap2 20020\n")
exe = os.path.join(ROOT, "rowbowt_amd", "rb_align")
for flags in ([], ["-s"], ["-s", "-m"]):
    t0 = time.perf_counter()
    p = subprocess.run([exe] + flags + ["/tmp/cli_idx/idx", path], stdout=open("/tmp/cli_out.txt", "wb"), stderr=subprocess.PIPE)
    dt = time.perf_counter() - t0
    sz = os.path.getsize("/tmp/cli_out.txt")
    print(f"rb_align {' '.join(flags) or '(count)':8s}: {N} x {m} bp FASTQ ({os.path.getsize(path)/1e6:.0f} MB) -> {sz/1e6:.0f} MB of text in {dt:.2f} s = {N/dt:.3e} reads/s"
          f"   [stderr: {p.stderr.decode().strip().splitlines()[-1]}]")

exe2 = os.path.join(ROOT, "rowbowt_amd", "rb_markers")
for flags in ([], ["--threads", "16"], ["--heuristic", "--best-strand-only", "--min-seed-length", "30", "--threads", "16"]):
    t0 = time.perf_counter()
    p = subprocess.run([exe2] + flags + ["/tmp/cli_idx/idx", path], stdout=open("/tmp/cli_out.txt", "wb"), stderr=subprocess.PIPE)
    dt = time.perf_counter() - t0
    sz = os.path.getsize("/tmp/cli_out.txt")
    print(f"rb_markers {' '.join(flags) or '(defaults)'}: {N} x {m} bp FASTQ -> {sz/1e6:.0f} MB of text in {dt:.2f} s = {N/dt:.3e} reads/s"
          f"   [stderr: {p.stderr.decode().strip().splitlines()[-1]}]")
