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
