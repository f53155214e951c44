#!/usr/bin/env python3
"""Load time of an index from its native cache file in a FRESH process, stage by stage (profiles/r03_load_time.txt).

  python tools/load_time.py --make /dev/shm/x.rbgpu [--L ... --H ...]        # synthesise the bench index, write the cache, exit
  python tools/load_time.py --make /dev/shm/x.rbgpu --pangenome --L 250000000 --H 200   # a true BWT at n = 5e10 (pangenome_bwt.py)
  RBG_VERBOSE=1 python tools/load_time.py --load /dev/shm/x.rbgpu            # rbg_load_cache, timed; a few reads searched

Two processes on purpose: the loading process has allocated and freed nothing on the GPU before, so what it measures is
the library's own work plus what the platform charges for a large hipMalloc (tools/alloc_probe.py) -- not the clearing
of memory a synthesis step has just freed.  Replaces nothing in the reference; its load is rowbowt_io.hpp:176-189."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--make")
    ap.add_argument("--load")
    ap.add_argument("--pangenome", action="store_true")
    ap.add_argument("--L", type=int, default=40_000_000)
    ap.add_argument("--H", type=int, default=50)
    ap.add_argument("--site-rate", type=float, default=0.01)
    ap.add_argument("--seed", type=int, default=20240229)
    ap.add_argument("--kmer-steps", type=int, default=0)
    ap.add_argument("--layout", choices=("auto", "slots", "runs"), default="auto")
    ap.add_argument("--run-depths", type=lambda v: int(v, 0), default=0, help="RBG_OPT_RUN_DEPTHS (run-indexed layout; 0 = all depths)")
    args = ap.parse_args()
    import numpy as np
    import rowbowt_amd as ra
    from rowbowt_amd import capi

    if args.make:
        import torch
        dev = torch.device("cuda", 0)
        if args.pangenome:
            from rowbowt_amd.tools import pangenome_bwt as pb
            pg = pb.make_pangenome(args.L, args.H, args.site_rate, args.seed, dev)
            inp = pb.build_runs(pg, log=lambda *a: None)
        else:
            from rowbowt_amd.tools import synth_pangenome as sp
            text, info = sp.make_text(args.L, args.H, args.site_rate, args.seed, dev)
            inp = sp.index_inputs(text, sp.suffix_array(text))
        t0 = time.time()
        capi.convert_runs(inp["heads"], inp["lens"], inp["ssa"], inp["esa"], out_path=args.make)
        print(json.dumps({"made": args.make, "n": int(inp["n"]), "r": int(inp["r"]), "file_bytes": os.path.getsize(args.make),
                          "convert_runs_s": time.time() - t0}))
        return
    if args.kmer_steps:
        capi.set_default_option(capi.OPT_KMER_STEPS, args.kmer_steps)
    if args.layout != "auto":
        capi.set_default_option(capi.OPT_RANK_LAYOUT, {"slots": 1, "runs": 2}[args.layout])
    if args.run_depths:
        capi.set_default_option(capi.OPT_RUN_DEPTHS, args.run_depths)
    ra.lib()
    t0 = time.time()
    rb = ra.RowBowt.from_cache(args.load, ra.LoadRbwtFlag.SA, device=0)
    t_load = time.time() - t0
    ix = rb.info()
    seqs, off = ra.pack_reads([b"ACGTACGTACGTAGCTAGCTAGCATCGATCGATCAGCTAGCTAGCATCGATCGACTAGCTAGCTAGC", b"ACGT", b"TTTTTTTTTTTT"])
    lo, hi, k = rb.find_range_w_toehold(seqs, off)
    print(json.dumps({"loaded": args.load, "load_s": t_load, "n": int(ix.n), "r": int(ix.r), "hbm_bytes": int(ix.hbm_bytes),
                      "symbols_per_gather": int(ix.kmer_steps), "rank_layout": int(ix.rank_layout), "pos_bytes": int(ix.pos_bytes),
                      "file_bytes": os.path.getsize(args.load), "sanity_ranges": [[int(a), int(b)] for a, b in zip(lo, hi)]}))
    rb.close()


if __name__ == "__main__":
    main()
