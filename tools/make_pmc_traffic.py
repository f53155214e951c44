#!/usr/bin/env python3
"""rocprofv3 --pmc passes (tools/pmc_passes.sh) -> profiles/pmc_traffic.json, the file bench.py reads for
roofline.traffic / request_roof.tcc_miss_per_launch.  Stamped with the sha256 of the librbg.so that was
profiled: bench.py drops the numbers when it runs a different build.
usage: make_pmc_traffic.py <gpurun_out/pmc_TAG> <source note> > profiles/pmc_traffic.json"""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def label(kernel):
    k = kernel.replace("rbg::(anonymous namespace)::", "")
    if "k_find_range<" in k and "packed" not in k:
        args = k.split("<", 1)[1].split(">", 1)[0].replace(" ", "").split(",")
        # <P, TOEHOLD, USE_FTAB, STATS>: the timed launches are the table-using, non-instrumented ones
        if len(args) >= 3 and args[2] == "true" and (len(args) < 4 or args[3] == "false"):
            return "k_find_range<toehold>" if args[1] == "true" else "k_find_range<count>"
        return None
    if "k_locate_fill<" in k:
        args = k.split("<", 1)[1].split(">", 1)[0].replace(" ", "").split(",")
        return "k_locate_fill" if len(args) < 2 or args[1] == "false" else None
    return None


def main():
    d = sys.argv[1]
    per = collections.defaultdict(lambda: collections.defaultdict(float))  # (label, counter) -> dispatch -> value
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            lab = label(r["Kernel_Name"])
            if lab:
                per[(lab, r["Counter_Name"])][(f, r["Dispatch_Id"])] += float(r["Counter_Value"])
    out = {"_source": sys.argv[2] if len(sys.argv) > 2 else d,
           "_librbg_sha256": hashlib.sha256(open(os.path.join(ROOT, "rowbowt_amd", "librbg.so"), "rb").read()).hexdigest(),
           "_units": "FETCH_SIZE / WRITE_SIZE are reported in KiB; bytes = value x 1024, counted as reported (an isolated random 16-byte "
                     "gather reports 64.0 B: profiles/r01_calib_gather_roof_pmc.txt); per launch = the largest dispatch (the full batch)"}
    for (lab, ctr), disp in sorted(per.items()):
        out.setdefault(lab, {})
        v = max(disp.values())
        if ctr == "FETCH_SIZE":
            out[lab]["fetch_bytes"] = v * 1024
        elif ctr == "WRITE_SIZE":
            out[lab]["write_bytes"] = v * 1024
        elif ctr == "TCC_MISS_sum":
            out[lab]["tcc_miss_per_launch"] = v
        elif ctr == "TCC_REQ_sum":
            out[lab]["tcc_req_per_launch"] = v
        elif ctr == "TCC_HIT_sum":
            out[lab]["tcc_hit_per_launch"] = v
    for lab, v in out.items():
        if isinstance(v, dict) and "fetch_bytes" in v and "write_bytes" in v:
            v["hbm_bytes_per_launch"] = v["fetch_bytes"] + v["write_bytes"]
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
