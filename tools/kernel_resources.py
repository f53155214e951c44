#!/usr/bin/env python3
"""Registers / scratch / occupancy / LDS of this repo's kernels as the compiler reports them
(-Rpass-analysis=kernel-resource-usage), one line per kernel.  usage: tools/kernel_resources.py [k_search.hip ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "rowbowt_amd", "csrc")


def main():
    files = sys.argv[1:] or [f for f in sorted(os.listdir(SRC)) if f.startswith("k_") and f.endswith(".hip")]
    for f in files:
        p = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-c", f, "-o", "/dev/null",
                            "-Rpass-analysis=kernel-resource-usage"], cwd=SRC, capture_output=True, text=True)
        cur = None
        for line in p.stderr.splitlines():
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
                cur = {"name": name}
                continue
            if cur is None:
                continue
            for key, pat in (("vgpr", r"\bVGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"SGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                             ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
                m = re.search(pat, line)
                if m:
                    cur[key] = int(m.group(1))
            if "lds" in cur:
                if "rbg::" in cur["name"]:
                    short = re.sub(r"\(rbg::DevIndex.*|\((unsigned|void|rbg::Run|rbg::Phi).*", "", cur["name"].replace("rbg::(anonymous namespace)::", "").replace("void ", ""))
                    print(f"{f:14s} {short:60s} vgpr {cur.get('vgpr'):3d} agpr {cur.get('agpr', 0):3d} scratch {cur.get('scratch'):4d} B occ {cur.get('occ')} lds {cur.get('lds')}")
                cur = None


if __name__ == "__main__":
    main()
