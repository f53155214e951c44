#!/usr/bin/env python3
"""DESIGN.md 2c's table "what a default rbg_load builds", generated from what the loads themselves reported (rbg_info / rbg_layout_info as the
committed JSON lines of bench.py and tools/pangenome_stream.py carry them): input (n, r, free HBM) -> budget -> depths kept, records per depth, phi form,
replica bytes, and the rate measured from that replica.  A rule edit (capi/load.ipp options_for / upload(), capi/upload_runs.ipp) shows in the driver-run sizes
-- the bench line's config.index.layout_info and tests/test_gpu_scale.py::test_pangenome_shape_r_above_1e8_default_load -- without a builder-side full-size run;
the two largest rows need one (60 s loads of 100-200 GB).

usage: tools/layout_rules_table.py [json ...]     (default: the round's files under profiles/; prints the markdown table)
tests/test_layout_rules_table.py checks that DESIGN.md holds exactly this output."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = ["profiles/r06_bench.json", "profiles/r06_bench.json#pangenome_shape", "profiles/r06_pangenome_stream_n5e10_default.json",
           "profiles/r06_pangenome_stream_r1e9_default.json"]


def load(path):
    """the JSON line of a file; 'file#pangenome_shape' = that block of a bench.py line (tools/pangenome_stream.py --preset driver run by bench.py)"""
    f, _, block = path.partition("#")
    d = json.loads(open(os.path.join(ROOT, f)).read().strip().splitlines()[-1])
    if block:
        b = d[block]
        d = {"value": b["value"], "config": {"index": b["index"]}}
    return d


def row_of(path):
    d = load(path)
    ix = d["config"]["index"]
    li = ix.get("layout_info") or {}
    kept = li.get("depths_kept") or [i + 1 for i in range(8) if li.get("depth_mask_kept", 0) >> i & 1]
    recs = li.get("depths_with_records") or [i + 1 for i, b in enumerate(li.get("rec_bytes", [])) if b]
    rec_gb = sum(li.get("rec_bytes", [])) / 1e9
    phi = f"slots ({li['phi_slot_bytes'] / 1e9:.1f} GB)" if li.get("phi_slots") else "list + directory"
    raised = li.get("budget_raised", 0)
    return (f"| {ix['n']:.2e} | {ix['r']:.2e} | {ix.get('H', '')} | {ix['hbm_free_at_load'] / 1e9:.0f} | {ix['hbm_budget'] / 1e9:.0f}{' (raised to 3/4)' if raised else ' (1/4)'} | "
            f"{ix.get('symbols_per_gather', '')} | {', '.join(map(str, kept))} | {', '.join(map(str, recs)) or 'none'} ({rec_gb:.1f} GB) | {phi} | {ix['hbm_bytes'] / 1e9:.1f} | "
            f"{d['value']:.3g} | `{path}` |")


def table(paths):
    out = ["| n | r | haplotypes | free HBM (GB) | budget (GB) | symbols per step | depths kept | depths with bucket records | phi | replica (GB) | reads/s from it | file |",
           "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for p in paths:
        if os.path.exists(os.path.join(ROOT, p.partition("#")[0])):
            out.append(row_of(p))
    return "\n".join(out)


if __name__ == "__main__":
    print(table(sys.argv[1:] or DEFAULT))
