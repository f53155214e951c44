#!/usr/bin/env python3
"""End-to-end rates of the command-line tools ON THE BENCH-SCALE INDEX (profiles/r03_cli_rate.txt): rb_align count /
-s (about 41 locations per read, ten times the text of the toy index) / -s -m, plain FASTQ in the page cache and .gz
(many gzip members), and rb_markers on both strands with the synthetic marker array.  GPU box only.

The index is the bench's synthetic chr22-scale pangenome (rowbowt_amd/tools/synth_pangenome.py), written once as the native
cache file with its marker array and a .docs text (rbg_convert_runs_markers): the tools load "<prefix>.rbgpu" when there is
no "<prefix>.rbwt" (include/rbg.h).  Replaces nothing in the reference; what it times is rb_align.cpp:162-193 /
rb_markers.cpp:318-535 as this engine runs them.

usage: cli_rate_bench.py [--reads 10000000] [--gz-reads 4000000] [--marker-reads 2000000] [--L ..] [--H ..]"""
import argparse
import gzip
import os
import subprocess
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def fastq_bytes(reads):
    """'@r%08d\\n' + bases + '\\n+\\n' + '~' * m + '\\n' per record, as one byte matrix"""
    n, m = reads.shape
    rec = np.empty((n, 2 + 8 + 1 + m + 3 + m + 1), dtype=np.uint8)
    rec[:, 0], rec[:, 1] = ord("@"), ord("r")
    ids = np.arange(n)
    for d in range(8):
        rec[:, 2 + 7 - d] = ord("0") + (ids // 10**d) % 10
    rec[:, 10] = 10
    rec[:, 11:11 + m] = reads
    rec[:, 11 + m], rec[:, 12 + m], rec[:, 13 + m] = 10, ord("+"), 10
    rec[:, 14 + m:14 + 2 * m] = ord("~")
    rec[:, 14 + 2 * m] = 10
    return rec


def _bgzf_part(args):
    """bytes [a, b) of the file as BGZF blocks (htslib's blocked gzip: members of at most 64 KB that carry their compressed
    size in the 'BC' extra subfield), without the end-of-file block"""
    import struct
    import zlib
    path, a, b = args
    with open(path, "rb") as f:
        f.seek(a)
        data = f.read(b - a)
    out = bytearray()
    for o in range(0, len(data), 65280):
        chunk = data[o:o + 65280]
        c = zlib.compressobj(1, zlib.DEFLATED, -15)
        payload = c.compress(chunk) + c.flush()
        out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(payload) + 8 - 1)
        out += payload + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))
    return bytes(out)


def numa_probe(exe, prefix, fq, n, th, reps):
    """rb_align -s -m in fresh processes: (a) pinned buffers wherever the process was started (RBG_PIN_NUMA=0), the process started on each NUMA
    node in turn (the child sets its own affinity to that node's CPUs before anything else runs: what the scheduler's choice amounts to);
    (b) the same with the library's default (buffers allocated from the GPU's node)."""
    import re
    nodes = sorted(int(d[4:]) for d in os.listdir("/sys/devices/system/node") if re.fullmatch(r"node\d+", d))

    def cpus_of(node):
        out = []
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            out += list(range(int(a), int(b or a) + 1))
        return out
    print(f"NUMA nodes {nodes}: " + "; ".join(f"node {k}: {len(cpus_of(k))} CPUs" for k in nodes), flush=True)
    res = {}
    for pin in ("0", "1"):
        for rep in range(reps):
            node = nodes[rep % len(nodes)]
            cpus = cpus_of(node)
            env = dict(os.environ, RB_ALIGN_TRACE="1", RBG_NUMA_TRACE="1", RBG_PIN_NUMA=pin)
            t0 = time.perf_counter()
            p = subprocess.run([exe, "-s", "-m"] + th + [prefix, fq], stdout=open("/dev/null", "wb"), stderr=subprocess.PIPE, timeout=600, env=env,
                               preexec_fn=lambda c=cpus: os.sched_setaffinity(0, c))
            dt = time.perf_counter() - t0
            err = p.stderr.decode().strip().splitlines()
            if p.returncode != 0:
                print(f"exit {p.returncode}: {err[-3:]}", flush=True)
                continue
            load_s, query_s = (float(x) for x in err[-1].split())
            pinned = [l for l in err if l.startswith("rbg: pinned buffer")]
            where = sorted({(int(re.search(r"first page on node (-?\d+)", l).group(1)), int(re.search(r"GPU \d+ on node (-?\d+)", l).group(1))) for l in pinned})
            trace = [l for l in err if l.startswith("rb_align loop:")]
            rate = n / query_s
            res.setdefault((pin, node), []).append(rate)
            sizes = [int(re.search(r"pinned buffer of (\d+) MB", l).group(1)) for l in pinned]
            print(f"RBG_PIN_NUMA={pin} process on node {node}: {len(pinned)} pinned buffers {sizes} MB on (buffer node, GPU node) {where}; query loop {query_s:6.3f} s = {rate:.3e} reads/s "
                  f"(process {dt:.2f} s)" + (f"   [{trace[-1][15:]}]" if trace else ""), flush=True)
    print("# summary: reads/s by (RBG_PIN_NUMA, node the process ran on): min / median / max")
    for key in sorted(res):
        v = sorted(res[key])
        print(f"#   pin={key[0]} node={key[1]}: {v[0]:.3e} / {v[len(v) // 2]:.3e} / {v[-1]:.3e}  ({len(v)} processes)")
    for pin in ("0", "1"):
        allv = sorted(x for k, v in res.items() if k[0] == pin for x in v)
        if allv:
            print(f"#   pin={pin}, all nodes: min {allv[0]:.3e}, max {allv[-1]:.3e}, spread {(allv[-1] - allv[0]) / allv[-1] * 100:.0f} %")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--gz-reads", type=int, default=4_000_000)
    ap.add_argument("--marker-reads", type=int, default=2_000_000)
    ap.add_argument("--L", type=int, default=40_000_000)
    ap.add_argument("--H", type=int, default=50)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--dir", default="/tmp/cli_big")
    ap.add_argument("--only-s", action="store_true", help="only the rb_align -s rows on plain FASTQ (tuning runs)")
    ap.add_argument("--only-markers", action="store_true",
                    help="only the rb_markers rows, each under the layout the tool gets by default (RBG_LAYOUT_AUTO: the run-indexed replica) and under "
                         "RBG_LAYOUT=prefer-slots (what the tool forced until round 5), three processes each (profiles/r05_rb_markers_layout.txt)")
    ap.add_argument("--only-sm", type=int, default=0, help="only the rb_align -s -m row on plain FASTQ, this many times (its spread)")
    ap.add_argument("--numa-probe", type=int, default=0,
                    help="the process-to-process spread of rb_align -s -m and its cause: this many fresh processes with the pinned result buffers left where "
                         "the scheduler put the process (RBG_PIN_NUMA=0) and as many with them on the GPU's NUMA node (the library's default), each started on "
                         "a CPU of alternating sockets; prints where process, GPU and buffers were, the rates, and the summary (profiles/r04_numa_probe.txt)")
    args = ap.parse_args()
    import torch
    from rowbowt_amd import capi
    from rowbowt_amd.tools import synth_pangenome as sp

    os.makedirs(args.dir, exist_ok=True)
    prefix = os.path.join(args.dir, "idx")
    dev = torch.device("cuda", 0)
    t0 = time.time()
    text, info = sp.make_text(args.L, args.H, 0.01, 20240229, dev)
    sa = sp.suffix_array(text)
    inp = sp.index_inputs(text, sa)
    markers = sp.marker_array(text, info, sa, w=10)
    del sa
    docs = "".join(f"hap{h} {h * info['unit']}\n" for h in range(args.H))
    capi.convert_runs(inp["heads"], inp["lens"], inp["ssa"], inp["esa"], out_path=prefix + ".rbgpu", markers=markers, docs_text=docs)
    m = 100
    reads, _ = sp.sample_reads(text, info, args.reads, m, seed=20240231, sub_rate=0.1)
    h_reads = reads.cpu().numpy()
    del text, reads
    torch.cuda.empty_cache()
    fq = os.path.join(args.dir, "reads.fq")
    fastq_bytes(h_reads).tofile(fq)
    rec_bytes = 2 + 8 + 1 + m + 3 + m + 1
    fqm = os.path.join(args.dir, "reads_m.fq")
    fastq_bytes(h_reads[:args.marker_reads]).tofile(fqm)
    gz = os.path.join(args.dir, "reads.bgzf.gz")     # BGZF: blocks the tools inflate in parallel
    nparts = 64
    cuts = [(args.gz_reads * i // nparts) * rec_bytes for i in range(nparts + 1)]
    with ProcessPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex, open(gz, "wb") as out:
        for blob in ex.map(_bgzf_part, [(fq, cuts[i], cuts[i + 1]) for i in range(nparts)]):
            out.write(blob)
        out.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))   # the end-of-file block
    gz1 = os.path.join(args.dir, "reads.1m.gz")      # plain gzip, one member: zlib's single stream
    n1 = min(1_000_000, args.gz_reads)
    with open(fq, "rb") as f, gzip.open(gz1, "wb", compresslevel=1) as g:
        g.write(f.read(n1 * rec_bytes))
    subprocess.run(["cat", fq, fqm, gz, gz1], stdout=subprocess.DEVNULL)   # page cache
    print(f"bench-scale index n = {inp['n']}, r = {inp['r']}, {len(markers[0])} marker runs; cache file {os.path.getsize(prefix + '.rbgpu') / 1e6:.0f} MB; "
          f"{args.reads} x {m} bp reads sampled from the text (10 % with one substitution), FASTQ {os.path.getsize(fq) / 1e6:.0f} MB in the page cache, "
          f"BGZF of the first {args.gz_reads}: {os.path.getsize(gz) / 1e6:.0f} MB, plain one-member gzip of the first {n1}: {os.path.getsize(gz1) / 1e6:.0f} MB; "
          f"{os.cpu_count()} logical CPUs; prepared in {time.time() - t0:.0f} s",
          flush=True)
    del inp, markers, h_reads
    time.sleep(8)   # (the memory the synthesis freed is cleared in the background: tools/alloc_probe.py)
    exe = os.path.join(ROOT, "rowbowt_amd", "rb_align")
    th = ["--threads", str(args.threads)]
    out_txt = os.path.join(args.dir, "out.txt")
    rows = ((th, fq, args.reads, "/dev/null"), (["-s"] + th, fq, args.reads, "/dev/null"), (["-s"] + th, fq, args.reads, out_txt),
            (["-s", "-m"] + th, fq, args.reads, "/dev/null"), (th, gz, args.gz_reads, "/dev/null"), (["-s"] + th, gz, args.gz_reads, "/dev/null"),
            (th, gz1, n1, "/dev/null"), (["-s"] + th, gz1, n1, "/dev/null"))
    if args.only_s:
        rows = tuple(r for r in rows if r[0][:1] == ["-s"] and "-m" not in r[0] and r[1] == fq)
    if args.only_sm:
        rows = tuple(r for r in rows if "-m" in r[0]) * args.only_sm
    if args.numa_probe:
        numa_probe(exe, prefix, fq, args.reads, th, args.numa_probe)
        return
    if args.only_markers:
        rows = ()
    for flags, path, n, out in rows:
        t0 = time.perf_counter()
        env = dict(os.environ, RB_ALIGN_TRACE="1")
        try:
            p = subprocess.run([exe] + flags + [prefix, path], stdout=open(out, "wb"), stderr=subprocess.PIPE, timeout=600, env=env)
        except subprocess.TimeoutExpired:
            print(f"rb_align {' '.join(flags)} {os.path.basename(path)}: TIMEOUT", flush=True)
            continue
        dt = time.perf_counter() - t0
        err = p.stderr.decode().strip().splitlines()
        if p.returncode != 0:
            print(f"rb_align {' '.join(flags)} {os.path.basename(path)}: exit {p.returncode}: {err[-3:]}", flush=True)
            continue
        load_s, query_s = (float(x) for x in err[-1].split())
        trace = [l for l in err if l.startswith("rb_align loop:")]
        sz = os.path.getsize(out) if out != "/dev/null" else 0
        print(f"rb_align {' '.join(flags):22s} {os.path.basename(path):14s} -> {os.path.basename(out):8s}: process {dt:6.2f} s (index load {load_s:5.2f} s); query loop "
              f"{query_s:6.3f} s = {n / query_s:.3e} reads/s" + (f"; {sz / 1e6:.0f} MB of text" if sz else "") + (f"   [{trace[-1][15:]}]" if trace else ""), flush=True)
    if args.only_s or args.only_sm:
        return
    exe2 = os.path.join(ROOT, "rowbowt_amd", "rb_markers")
    variants = [("", {})]
    if args.only_markers:
        variants = [("[default layout] ", {}), ("[RBG_LAYOUT=prefer-slots] ", {"RBG_LAYOUT": "prefer-slots"})] * 3
    for label, extra_env in variants:
      for flags in (th, ["--heuristic", "--best-strand-only", "--min-seed-length", "30"] + th):
        t0 = time.perf_counter()
        p = subprocess.run([exe2] + flags + [prefix, fqm], stdout=open(out_txt, "wb"), stderr=subprocess.PIPE, timeout=600, env=dict(os.environ, RB_ALIGN_TRACE="1", **extra_env))
        dt = time.perf_counter() - t0
        sz = os.path.getsize(out_txt)
        err = p.stderr.decode().strip().splitlines()
        last = err[-1] if err else ""
        tr = [l for l in err if l.startswith("rb_markers loop:")]
        try:
            loop_s = float(last.split("took:")[1].split()[0])
            loop = f"query loop {loop_s:.3f} s = {args.marker_reads / loop_s:.3e} reads/s"
        except Exception:  # noqa: BLE001
            loop = last
        print(f"{label}rb_markers {' '.join(flags)}: {args.marker_reads} x {m} bp (both strands) -> {sz / 1e6:.0f} MB of text; process {dt:.2f} s (exit {p.returncode}); {loop}" + (f"   [{tr[-1][17:]}]" if tr else ""), flush=True)


if __name__ == "__main__":
    main()
