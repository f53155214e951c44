#!/usr/bin/env python3
"""rocprofv3 --pmc passes (tools/pmc_passes.sh) -> profiles/pmc_traffic.json, the file bench.py reads for
roofline.traffic / request_roof.tcc_miss_per_launch.  Stamped with the sha256 of the librbg.so that was
profiled: bench.py drops the numbers when it runs a different build.
usage: make_pmc_traffic.py <gpurun_out/pmc_TAG> <source note> [<more pass directories>] > profiles/pmc_traffic.json"""
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


def label_runs(kernel):
    """k_find_range_runs<P, TOEHOLD, PACKED, STATS, GLDS, STAGE>: the byte-read, non-instrumented launches with LDS-direct record fetches (what bench.py times;
    staged or not: a default run launches the staged form only)"""
    k = kernel.replace("rbg::(anonymous namespace)::", "")
    if "k_find_range_runs<" not in k:
        return None
    a = k.split("<", 1)[1].split(">", 1)[0].replace(" ", "").split(",")
    if len(a) < 5 or a[2] != "false" or a[3] != "false" or a[4] != "true":
        return None
    pb = 8 if a[0] == "unsignedlong" else 4
    return ("k_find_range<toehold>" if a[1] == "true" else "k_find_range<count>") + f" [runs, pos_bytes {pb}]"


def merge_runs(target, d):
    """--merge-runs <pmc_traffic.json> <run_indexed_pmc dir>: add the run-indexed kernels' counters (tools/pmc_run_indexed.sh) to the file"""
    out = json.load(open(target))
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            lab = label_runs(r["Kernel_Name"])
            if lab:
                per[(lab, r["Counter_Name"])][(f, r["Dispatch_Id"])] += float(r["Counter_Value"])
    names = {"FETCH_SIZE": ("fetch_bytes", 1024), "WRITE_SIZE": ("write_bytes", 1024), "TCC_MISS_sum": ("tcc_miss_per_launch", 1),
             "TCC_REQ_sum": ("tcc_req_per_launch", 1), "TCC_HIT_sum": ("tcc_hit_per_launch", 1)}
    for (lab, ctr), disp in sorted(per.items()):
        if ctr in names:
            out.setdefault(lab, {})[names[ctr][0]] = max(disp.values()) * names[ctr][1]
    for lab, v in out.items():
        if isinstance(v, dict) and "fetch_bytes" in v and "write_bytes" in v:
            v["hbm_bytes_per_launch"] = v["fetch_bytes"] + v["write_bytes"]
    out["_source_runs"] = "keys with [runs, ...]: tools/pmc_run_indexed.sh (tools/tune.py, RBG_TUNE_LAYOUT=runs, the same index and batch), same library build"
    json.dump(out, open(target, "w"), indent=1)


NAMES = {"FETCH_SIZE": ("fetch_bytes", 1024), "WRITE_SIZE": ("write_bytes", 1024), "TCC_MISS_sum": ("tcc_miss_per_launch", 1),
         "TCC_REQ_sum": ("tcc_req_per_launch", 1), "TCC_HIT_sum": ("tcc_hit_per_launch", 1)}


def _targs(kernel):
    k = kernel.replace("rbg::(anonymous namespace)::", "")
    head = k.split("<", 1)[0].split("(", 1)[0].split("::")[-1].split(" ")[-1]   # ("void rbg::...::name<args>(params)" as rocprofv3 prints it)
    return head, (k.split("<", 1)[1].split(">(", 1)[0].rsplit(">", 1)[0].replace(" ", "").split(",") if "<" in k else [])


def collect(d, label_fn):
    """per (label, counter): the largest dispatch's value (the full batch), over every pass directory under d"""
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            lab = label_fn(r["Kernel_Name"])
            if lab:
                per[(lab, r["Counter_Name"])][(f, r["Dispatch_Id"])] += float(r["Counter_Value"])
    out = collections.defaultdict(dict)
    for (lab, ctr), disp in sorted(per.items()):
        if ctr in NAMES:
            out[lab][NAMES[ctr][0]] = max(disp.values()) * NAMES[ctr][1]
    for v in out.values():
        if "fetch_bytes" in v and "write_bytes" in v:
            v["hbm_bytes_per_launch"] = v["fetch_bytes"] + v["write_bytes"]
    return out


def merge_seeds(target, d):
    """--merge-seeds <pmc_traffic.json> <dir>: the seeding kernels of the run-indexed layout (bench.py --markers; round 6): the marker seeds' two walks
    (k_marker_seeds_runs<P, FILL, LOG, STATS>: count walk <.., false, false, false> + fill walk <.., true, false, false>) summed, and the greedy seeds"""
    def lab(kernel):
        name, a = _targs(kernel)
        if name == "k_marker_seeds_runs" and len(a) >= 4 and a[2] == "false" and a[3] == "false":
            return "ms_fill" if a[1] == "true" else "ms_plan"
        if name == "k_greedy_seed_runs" and len(a) >= 3 and a[2] == "false":
            return "k_greedy_seed_runs"
        return None
    got = collect(d, lab)
    out = json.load(open(target))
    if "ms_plan" in got and "ms_fill" in got:
        out["k_marker_seeds_runs (plan + fill)"] = {k: got["ms_plan"].get(k, 0) + got["ms_fill"].get(k, 0) for k in set(got["ms_plan"]) | set(got["ms_fill"])}
        out["k_marker_seeds_runs (plan + fill)"]["parts"] = {"count_walk": got["ms_plan"], "fill_walk": got["ms_fill"]}
    if "k_greedy_seed_runs" in got:
        out["k_greedy_seed_runs"] = got["k_greedy_seed_runs"]
    out["_source_seeds"] = "seeding kernels: tools/run_profiles_r06.sh (rocprofv3 --pmc passes over bench.py --markers), same library build"
    json.dump(out, open(target, "w"), indent=1)


def merge_pangenome(target, d, key):
    """--merge-pangenome <pmc_traffic.json> <dir> <key>: K2 / K3 of tools/pangenome_stream.py --preset driver (key = 'pangenome_shape L=.. H=.. m=..': what the tool
    looks its roofline.traffic up under)"""
    def lab(kernel):
        name, a = _targs(kernel)
        if name == "k_find_range_runs" and len(a) >= 4 and a[1] == "true" and a[2] == "false" and a[3] == "false":
            return "find_range_w_toehold"
        if name in ("k_locate_fill", "k_locate_fill_runs2") and "true" not in a[1:3]:
            return "locate_fill"
        return None
    out = json.load(open(target))
    out[key] = dict(collect(d, lab))
    out["_source_pangenome"] = "pangenome_shape: rocprofv3 --pmc passes over tools/pangenome_stream.py --preset driver (tools/run_profiles_r06.sh), same library build"
    json.dump(out, open(target, "w"), indent=1)


def main():
    if sys.argv[1] == "--merge-runs":
        return merge_runs(sys.argv[2], sys.argv[3])
    if sys.argv[1] == "--merge-seeds":
        return merge_seeds(sys.argv[2], sys.argv[3])
    if sys.argv[1] == "--merge-pangenome":
        return merge_pangenome(sys.argv[2], sys.argv[3], sys.argv[4])
    d = sys.argv[1]
    per = collections.defaultdict(lambda: collections.defaultdict(float))  # (label, counter) -> dispatch -> value
    files = []
    for dd in [d] + sys.argv[3:]:                                           # (further directories: the passes over another headline replica)
        files += sorted(glob.glob(dd + "/**/*counter_collection.csv", recursive=True))
    for f in files:
        for r in csv.DictReader(open(f)):
            lab = label(r["Kernel_Name"]) or label_runs(r["Kernel_Name"])   # (the default bench's headline replica is run-indexed since round 4)
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
