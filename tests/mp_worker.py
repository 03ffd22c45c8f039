"""Worker for tests/test_multiproc.py: 2 ranks, gloo, CPU.  Each rank processes its LPT share of the chunks of one
small synthetic contig with the CPU oracle (tests may use the oracle) and the ranks agree on totals and timing through
the same reduce helpers bench.py uses."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch.distributed as dist  # noqa: E402

from clair3_rna_amd import shard, synth  # noqa: E402
from tests import helpers as H  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ref, rs, _ = synth.small_case(seed=51, ref_len=36000, n_genes=8, depth=15)
    chunks = [(6000 * i, 6000 * (i + 1)) for i in range(6)]
    import numpy as np
    span = np.array([sum((int(c) >> 4) for c in rs.cigar[int(r["cigar_off"]):int(r["cigar_off"]) + int(r["n_cigar"])] if (int(c) & 15) in (0, 2, 3))
                     for r in rs.reads])
    costs = shard.chunk_costs(rs.reads["pos"], span, chunks)
    plan = shard.lpt_assign(costs, world)
    dist.barrier()
    t0 = time.perf_counter()
    mine = {}
    for ci in plan[rank]:
        a, b = chunks[ci]
        mine[ci] = [l.split("\t")[1] for l in H.oracle_chunk(rs, ref, 1, a, b)["lines"]]
    elapsed = time.perf_counter() - t0 + 0.01 * rank
    n_local = sum(len(v) for v in mine.values())
    total = shard.reduce_sum(dist, n_local)
    tmax = shard.reduce_max(dist, elapsed)
    gathered = [None] * world
    dist.all_gather_object(gathered, (rank, elapsed, mine))
    if rank == 0:
        merged = {}
        for _r, _e, m in gathered:
            merged.update(m)
        print("MPRESULT " + json.dumps(dict(world=world, plan=plan, costs=costs, total=total, tmax=tmax,
                                            elapsed=[g[1] for g in gathered], per_chunk={str(k): v for k, v in merged.items()})), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
