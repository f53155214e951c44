"""DESIGN.md 2c's table "what a default rbg_load builds" is GENERATED (tools/layout_rules_table.py) from what the loads themselves reported in the committed
JSON lines (rbg_info / rbg_layout_info): the document must hold exactly the script's output, and every row must obey the rules it illustrates
(a rule edit shows its effect at every measured size).  The structures being sized: rle_string.hpp:131-161, toehold_sa.hpp:56-72."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import layout_rules_table as T  # noqa: E402


def test_design_holds_the_generated_table():
    table = T.table(T.DEFAULT)
    assert table.count("\n") >= 4, "a profile the table is made from is missing"
    assert table in open(os.path.join(ROOT, "DESIGN.md")).read()


def test_every_row_obeys_the_rules():
    for p in T.DEFAULT:
        d = T.load(p)
        ix = d["config"]["index"]
        li = ix["layout_info"]
        raised = li.get("budget_raised", 0)
        # the budget: a quarter of the free HBM, or -- an index of about 1e9 runs on a device the load has to itself -- three quarters
        frac = ix["hbm_budget"] / ix["hbm_free_at_load"]
        assert abs(frac - (0.75 if raised else 0.25)) < 0.01, (p, frac)
        assert bool(raised) == (ix["r"] > 8e8), p
        assert ix["hbm_bytes"] <= ix["hbm_budget"], p
        kept = li.get("depths_kept") or [i + 1 for i in range(8) if li.get("depth_mask_kept", 0) >> i & 1]
        recs = li.get("depths_with_records") or [i + 1 for i, b in enumerate(li["rec_bytes"]) if b]
        # depth 1 and the deepest are always kept; records for every kept depth come before the depths in between; phi slots while they fit
        assert kept[0] == 1 and kept[-1] == ix["symbols_per_gather"] and recs == kept and li["phi_slots"] > 0, p
        if len(kept) > 2:   # the depths in between stay only where records for all of them fit the budget too
            assert ix["hbm_bytes"] <= ix["hbm_budget"]
