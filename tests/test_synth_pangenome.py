"""The torch-based index synthesiser used by bench.py, checked on the CPU against the naive
numpy suffix sorter and through the oracle."""
import numpy as np
import torch

import naive
import orc
from rowbowt_amd.tools import synth_pangenome as sp


def test_suffix_array_matches_naive():
    for seed, L, H in ((1, 300, 3), (2, 1000, 5), (3, 64, 8)):
        text, info = sp.make_text(L, H, 0.02, seed, "cpu")
        sa = sp.suffix_array(text)
        want = naive.suffix_array(text.numpy())
        assert (sa.numpy() == want).all()
    # degenerate: long identical haplotypes (deep doubling)
    text, info = sp.make_text(500, 6, 0.0001, 4, "cpu")
    assert (sp.suffix_array(text).numpy() == naive.suffix_array(text.numpy())).all()


def test_index_inputs_and_reads_through_oracle():
    text, info = sp.make_text(2000, 6, 0.01, 7, "cpu")
    sa = sp.suffix_array(text)
    inp = sp.index_inputs(text, sa)
    assert int(inp["lens"].sum()) == info["n"] and inp["r"] == len(inp["heads"])
    o = orc.Oracle.from_runs(inp["heads"], inp["lens"], inp["ssa"], inp["esa"])
    reads, start = sp.sample_reads(text, info, 400, 50, seed=9, sub_rate=0.2)
    seqs = reads.numpy().reshape(-1)
    off = (np.arange(401) * 50).astype(np.uint64)
    lo, hi, k = o.find_range_w_toehold_batch(seqs, off)
    fm = naive.NaiveFM(text.numpy())
    matched = 0
    for i in range(400):
        q = reads[i].numpy().tobytes()
        assert (int(lo[i]), int(hi[i])) == fm.find_range(q)
        if hi[i] >= lo[i]:
            matched += 1
            assert o.locs_at(int(lo[i]), int(hi[i]), int(k[i])) == fm.locs(int(lo[i]), int(hi[i]))
    assert 250 < matched < 400
    o.close()


def test_marker_array_builder_matches_loop_version():
    """The vectorised marker-array synthesiser against a plain-loop construction (same rule as
    tests/test_gpu_parity.py::test_midscale_pangenome_all_queries)."""
    text, info = sp.make_text(1500, 5, 0.02, 11, "cpu")
    sa = sp.suffix_array(text)
    rs, re_, mo, mv = sp.marker_array(text, info, sa, w=6)
    n, unit, H, L = info["n"], info["unit"], info["H"], info["L"]
    t = text.numpy()
    isa = np.empty(n, dtype=np.int64)
    isa[sa.numpy()] = np.arange(n)
    base = t[:L]
    sites = np.flatnonzero((t[: H * unit].reshape(H, unit)[:, :L] != base[None, :]).any(axis=0))
    tags = {}
    for h in range(H):
        for s in sites:
            a = int(t[h * unit + s] != base[s])
            for d in range(6):
                if s - d >= 0:
                    tags.setdefault(int(isa[h * unit + s - d]), set()).add(int(s) | (a << 60))
    ws, we, wo, wv = [], [], [0], []
    for r in sorted(tags):
        vals = sorted(tags[r])
        if ws and we[-1] == r - 1 and len(vals) == 1 and wv[wo[-2]:wo[-1]] == vals:
            we[-1] = r
        else:
            ws.append(r); we.append(r); wv += vals; wo.append(len(wv))
    assert rs.tolist() == ws and re_.tolist() == we and mo.tolist() == wo and mv.tolist() == wv
    assert (rs[1:] > re_[:-1]).all()
