#!/usr/bin/env python3
"""Regenerates the files of tests/golden/ that are NOT transcribed from the reference's own tests:
outputs of the oracle (oracle/rb_oracle.c) and of the rb_markers model on the reference's shipped
fixture (tests/data/small.fa.*, simple_query.fq, error_query.fq).  They pin the oracle's behaviour:
tests/test_oracle_golden.py checks the oracle still reproduces them, the GPU tests check the HIP path
against them without the oracle in the loop.

reference_rb_tests.json is different: it holds the values asserted by the reference's tests/rb_tests.cpp
(file:line in its "source" fields) and is edited by hand only.

usage: python tests/golden/make_golden.py      (CPU only)
"""
import itertools
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
sys.path.insert(0, TESTS)
import orc  # noqa: E402
import rb_markers_model as RM  # noqa: E402

DATA = os.path.join(TESTS, "data")


def main(out=HERE):
    o = orc.Oracle.load(os.path.join(DATA, "small.fa"), orc.SA | orc.MA)
    recs = []
    for fn in ("simple_query.fq", "error_query.fq"):
        names, seqs = orc.read_fastx(os.path.join(DATA, fn))
        recs += [(fn, n, s) for n, s in zip(names, seqs)]
    # get_markers_greedy_seeding (rowbowt.hpp:406-482) without and with an ftab
    seeds = {"source": "oracle/rb_oracle.c orc_markers_greedy_seeding_ftab on tests/data/small.fa", "cases": []}
    for wsize, max_range, ftab_k in ((19, 1000, 0), (5, 1000, 0), (8, 1000, 6), (10, 2, 0), (10, 1000, 9)):
        case = {"wsize": wsize, "max_range": max_range, "ftab_k": ftab_k, "reads": []}
        for fn, name, seq in recs:
            case["reads"].append({"file": fn, "name": name.decode(), "seeds": [[lo, hi, qs, qe, mk] for lo, hi, qs, qe, mk in
                                                                               o.markers_greedy_seeding(seq, wsize, max_range, ftab_k)]})
        seeds["cases"].append(case)
    json.dump(seeds, open(os.path.join(out, "toy_marker_seeds.json"), "w"), indent=0, separators=(",", ":"))
    # RowBowt::build_ftab(4) + FTab::serialize (rowbowt.hpp:726-744, ftab.hpp:29-34)
    with open(os.path.join(out, "toy_k4.ftab"), "w") as f:
        for kmer in sorted("".join(t) for t in itertools.product("ACGT", repeat=4)):
            lo, hi = o.find_range(kmer.encode())
            if lo <= hi:
                f.write(f"{kmer} {lo} {hi}\n")
    # rb_markers' stdout (tests/rb_markers_model.py over the oracle), defaults and the heuristic worker
    both = [(n, s) for _, n, s in recs]
    open(os.path.join(out, "toy_rb_markers_default.txt"), "w").write(RM.expected_stdout(o, both))
    open(os.path.join(out, "toy_rb_markers_heuristic.txt"), "w").write(
        RM.expected_stdout(o, both, wsize=8, heuristic=True, best_strand=True, min_seed_len=5, read_len=20))
    # count + locate per read of error_query.fq (the reference only asserts simple_query.fq)
    loc = {"source": "oracle find_range_w_toehold + locs_at on tests/data/error_query.fq", "reads": []}
    names, seqs = orc.read_fastx(os.path.join(DATA, "error_query.fq"))
    for n, s in zip(names, seqs):
        lo, hi, k = o.find_range_w_toehold(s)
        loc["reads"].append({"name": n.decode(), "lo": lo, "hi": hi, "toehold": k, "locs": o.locs_at(lo, hi, k) if hi >= lo else []})
    json.dump(loc, open(os.path.join(out, "toy_error_query_locate.json"), "w"), indent=0, separators=(",", ":"))
    o.close()


if __name__ == "__main__":
    main()
