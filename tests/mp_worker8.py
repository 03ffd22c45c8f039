"""Worker for tests/test_multiproc.py::test_eight_ranks_*: world_size 8, gloo, CPU.  The whole-genome shard of BASELINE.json
configs[2]: the 24 GRCh38 contigs dealt to 8 ranks largest-first; every rank checks its host budget and the ranks agree on the plan."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch.distributed as dist  # noqa: E402

from clair3_rna_amd import shard  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    costs = [l for _n, l in shard.GRCH38]
    plan = shard.lpt_assign(costs, world)
    n_thr, cpus = shard.host_budget(apply=True)
    mine = plan[rank]
    load = float(sum(costs[i] for i in mine))
    tot = shard.reduce_sum(dist, load)
    worst = shard.reduce_max(dist, load)
    gathered = [None] * world
    dist.all_gather_object(gathered, (rank, mine, n_thr, sorted(cpus), os.environ.get("C3R_THREADS")))
    if rank == 0:
        print("MPRESULT " + json.dumps(dict(world=world, total=tot, worst=worst, ranks=gathered)), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
