#!/usr/bin/env python3
"""Compiles every kernel file of rowbowt_amd/csrc to gfx950 assembly (hipcc cross-compiles without a GPU) and reports, per kernel:
scratch bytes per lane, VGPRs, and the 64-bit shifts whose AMOUNT is in the kernel's last allocated VGPR.

Why the last: on gfx950 a v_lshlrev_b64 / v_lshrrev_b64 / v_ashrrev_b64 that takes its amount from the last VGPR of the wave's
allocation (allocation granule: 8 registers) reads the amount from elsewhere -- profiles/r05_shift64_last_vgpr.md (k_lf_runs, 72 VGPRs,
amount in v71: results shifted by the contents of v0; seen as a memory fault whose appearance depended on what ran before).

usage: tools/check_isa.py [files...]   (default: every k_*.hip)     exit 1 when a kernel has such a shift (scratch is listed: the slot
layout's search kernels and rocPRIM's sorts use some by design; no kernel of the run-indexed layout may -- tests/test_capi_host.py)
"""
import glob
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rowbowt_amd", "csrc")
GRANULE = 8


def kernels_of(hip_file):
    """[(mangled name, vgprs, scratch bytes, [amount registers of its VGPR-amount 64-bit shifts])]"""
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        p = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-S", "--cuda-device-only", hip_file, "-o", out],
                           cwd=CSRC, capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError(p.stderr[-2000:])
        text = open(out).read()
    res, name, shifts = [], None, []
    for line in text.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, shifts = m.group(1), []
            continue
        m = re.search(r"\bv_(?:lshl|lshr|ashr)rev_b64\S*\s+v\[\d+:\d+\], v(\d+),", line)
        if m and name:
            shifts.append(int(m.group(1)))
        m = re.search(r"\.amdhsa_next_free_vgpr (\d+)", line)
        if m and name:
            res.append([name, int(m.group(1)), None, shifts])
            name = None
        m = re.search(r"^; ScratchSize: (\d+)", line)
        if m and res and res[-1][2] is None:
            res[-1][2] = int(m.group(1))
    return [tuple(r) for r in res]


def hazards(kernels):
    """kernels whose 64-bit shifts take the amount from the last register of the allocation"""
    bad = []
    for name, nv, _scratch, shifts in kernels:
        alloc = (nv + GRANULE - 1) // GRANULE * GRANULE
        hit = [s for s in shifts if s == alloc - 1]
        if hit:
            bad.append((name, nv, len(hit)))
    return bad


def scan(files=None, workers=4):
    files = files or sorted(os.path.basename(f) for f in glob.glob(os.path.join(CSRC, "k_*.hip")))
    with ThreadPoolExecutor(workers) as ex:
        return dict(zip(files, ex.map(kernels_of, files)))


def main():
    per_file = scan(sys.argv[1:] or None)
    rc = 0
    for f, ks in per_file.items():
        spilled = [(n, s) for n, _v, s, _sh in ks if s]
        bad = hazards(ks)
        print(f"{f}: {len(ks)} kernels, max VGPRs {max((v for _n, v, _s, _sh in ks), default=0)}, {len(spilled)} with scratch, {len(bad)} shifting by their last VGPR")
        names = subprocess.run(["c++filt"], input="\n".join(n for n, *_ in spilled + bad), capture_output=True, text=True).stdout.splitlines()
        for (n, *rest), dn in zip(spilled + bad, names):
            print("   ", dn[:140], rest)
        rc = rc or (1 if bad else 0)
    return rc


if __name__ == "__main__":
    sys.exit(main())
