"""Multi-GPU sharding of the per-chunk work items.

The reference shards by `(contig, chunk_id, chunk_num)` rows of tmp/CHUNK_LIST fanned out with GNU parallel
(run_clair3_rna:441-449, :681-706); chunks are independent (they overlap +-33 bp on purpose) so there is no exchange
step and no collective on the data path.  Here: one process per GPU, a static largest-first (LPT) assignment of chunks
to ranks by read count, and torch.distributed used only for the barrier / max-over-ranks timing of bench.py.
"""
import numpy as np


def lpt_assign(costs, world):
    """Largest-processing-time-first greedy: returns `world` lists of item indices (each ascending)."""
    loads = [0.0] * world
    out = [[] for _ in range(world)]
    for i in sorted(range(len(costs)), key=lambda k: (-costs[k], k)):
        r = min(range(world), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += costs[i]
    return [sorted(x) for x in out]


def chunk_costs(read_pos, read_len, chunks):
    """Reads overlapping each chunk (ctg_start, ctg_end], from the sorted read starts — the LPT cost."""
    read_pos = np.asarray(read_pos)
    ends = read_pos + np.asarray(read_len)
    return [int(np.count_nonzero((read_pos < b + 33) & (ends > a - 33))) for a, b in chunks]


def reduce_max(dist, value, device="cpu"):
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_sum(dist, value, device="cpu"):
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
