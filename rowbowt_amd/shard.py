"""Read-stream sharding for the multi-GPU run: the index is replicated per GPU,
read i of a batch of N goes to rank i*G//N (contiguous blocks, so concatenating the per-rank
outputs in rank order restores the input order), there is NO data-path collective, and the four
global counters {reads, matched, sum occ, sum locs} are summed with one all-reduce
(RCCL on the GPU box: torch.distributed's "nccl" backend; gloo in the CPU tests)."""
import numpy as np


def shard_bounds(n_items, rank, world):
    """Contiguous block [begin, end) of rank `rank` out of `world`; sizes differ by at most one."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return (n_items * rank) // world, (n_items * (rank + 1)) // world


def shard_reads(seqs, off, rank, world):
    """Slice a packed batch (uint8 concat, uint64 offsets[N+1]) to this rank's reads."""
    n = len(off) - 1
    b, e = shard_bounds(n, rank, world)
    lo, hi = int(off[b]), int(off[e])
    return seqs[lo:hi], (off[b:e + 1] - off[b]).astype(np.uint64), (b, e)


def reduce_counters(counters, device=None, group=None):
    """Sum the 4 x u64 per-rank counters over all ranks (the only collective of the run).
    Returns a list of Python ints.  Without an initialised process group this is the identity."""
    import torch
    import torch.distributed as dist

    vals = [int(v) for v in counters]
    if not (dist.is_available() and dist.is_initialized()):
        return vals
    t = torch.tensor(vals, dtype=torch.int64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return [int(v) for v in t.cpu().tolist()]
