"""The N>1 path on CPU: two gloo ranks shard a read batch exactly like bench.py does across GPUs,
each computes its shard (with the oracle here: no GPU in this container), the global counters are
all-reduced, and the rank-ordered concatenation must equal the unsharded answer."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from rowbowt_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import json, os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import orc
from rowbowt_amd import shard
from test_oracle_vs_naive import sample_reads
import naive

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
o = orc.Oracle.load(os.path.join({root!r}, "tests", "data", "small.fa"), orc.SA)
heads, lens = o.runs()
text = naive.invert_bwt(naive.expand_bwt(heads, lens))
reads = sample_reads(text, 1001, 60, np.random.default_rng(99), spans=[(0, 10000), (10010, 20010), (20020, 30020)])
seqs, off = orc.pack_reads(reads)
my_seqs, my_off, (b, e) = shard.shard_reads(seqs, off, rank, world)
lo, hi, k = o.find_range_w_toehold_batch(my_seqs, my_off)
loc_off, locs = o.locs_at_batch(lo, hi, k)
occ = np.where(hi >= lo, hi - lo + 1, 0)
mine = [len(lo), int((hi >= lo).sum()), int(occ.sum()), int(len(locs))]
total = shard.reduce_counters(mine)
gathered = [None] * world
dist.all_gather_object(gathered, (b, e, lo.tolist(), hi.tolist(), k.tolist(), locs.tolist()))
if rank == 0:
    json.dump({{"total": total, "shards": gathered}}, open({out!r}, "w"))
dist.barrier()
dist.destroy_process_group()
"""


def test_shard_bounds_cover_and_order():
    for n in (0, 1, 7, 1000, 10_000_001):
        for world in (1, 2, 3, 8):
            edges = [shard.shard_bounds(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[r][1] == edges[r + 1][0] for r in range(world - 1))
            sizes = [e - b for b, e in edges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard.shard_bounds(10, 2, 2)


def test_reduce_counters_identity_without_group():
    assert shard.reduce_counters(np.array([1, 2, 3, 4], np.uint64)) == [1, 2, 3, 4]


@pytest.mark.parametrize("world", [2, 3])
def test_two_rank_gloo(tmp_path, world):
    import orc
    import naive
    from test_oracle_vs_naive import sample_reads
    out = tmp_path / "res.json"
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, out=str(out)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + world + os.getpid() % 500), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=180) == 0
    res = json.load(open(out))
    # unsharded answer
    o = orc.Oracle.load(os.path.join(ROOT, "tests", "data", "small.fa"), orc.SA)
    heads, lens = o.runs()
    text = naive.invert_bwt(naive.expand_bwt(heads, lens))
    reads = sample_reads(text, 1001, 60, np.random.default_rng(99), spans=[(0, 10000), (10010, 20010), (20020, 30020)])
    seqs, off = orc.pack_reads(reads)
    lo, hi, k = o.find_range_w_toehold_batch(seqs, off)
    loc_off, locs = o.locs_at_batch(lo, hi, k)
    occ = np.where(hi >= lo, hi - lo + 1, 0)
    assert res["total"] == [1001, int((hi >= lo).sum()), int(occ.sum()), len(locs)]
    shards = sorted(res["shards"], key=lambda s: s[0])
    assert shards[0][0] == 0 and shards[-1][1] == 1001
    cat = lambda j: [v for s in shards for v in s[j]]
    assert cat(2) == lo.tolist() and cat(3) == hi.tolist() and cat(4) == k.tolist() and cat(5) == locs.tolist()
    o.close()
