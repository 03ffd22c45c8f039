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


def contig_costs(bam_fn, contigs, lengths):
    """What each contig costs a rank, for the LPT deal of a whole sample: mapped reads per contig from the BAM index (its metadata
    pseudo-bin), else the compressed bytes the contig's records span, else the contig's length — ONE basis for all contigs, so that the
    costs are comparable.  -> (costs, name of the basis).  lengths: {contig: bp}."""
    weights = {}
    if str(bam_fn).endswith(".bam"):
        try:
            from . import bamio
            with bamio.BamFile(bam_fn) as b:
                if b.has_index:
                    weights = b.contig_weights()
        except Exception:
            weights = {}
    for k, basis in ((0, "mapped reads (BAM index)"), (1, "compressed bytes (BAM index)")):
        # (a contig of the calling list that the BAM header does not have holds no read: 0, not "unknown")
        vals = [weights.get(c, (0, 0))[k] for c in contigs] if weights else []
        if vals and all(v >= 0 for v in vals) and sum(vals) > 0:
            # (a contig without reads still costs its fetch and an empty scan: a small floor keeps them from piling up on one rank)
            floor = max(1, sum(vals) // (200 * len(vals)))
            return [max(int(v), floor) for v in vals], basis
    return [int(lengths[c]) for c in contigs], "length"


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


# GRCh38 primary contig lengths (chr1..22, X, Y): the genome BASELINE.json configs[2..3] shard by contig
GRCH38 = [("chr1", 248956422), ("chr2", 242193529), ("chr3", 198295559), ("chr4", 190214555), ("chr5", 181538259), ("chr6", 170805979),
          ("chr7", 159345973), ("chr8", 145138636), ("chr9", 138394717), ("chr10", 133797422), ("chr11", 135086622), ("chr12", 133275309),
          ("chr13", 114364328), ("chr14", 107043718), ("chr15", 101991189), ("chr16", 90338345), ("chr17", 83257441), ("chr18", 80373285),
          ("chr19", 58617616), ("chr20", 64444167), ("chr21", 46709983), ("chr22", 50818468), ("chrX", 156040895), ("chrY", 57227415)]


def imbalance(costs, plan):
    """max rank load / mean rank load of an assignment (1.0 = perfect)."""
    loads = [sum(costs[i] for i in p) for p in plan]
    mean = sum(loads) / float(len(loads))
    return max(loads) / mean if mean > 0 else 1.0


def local_world():
    """(local_rank, local_world_size) of this process under torch.distributed.run (1 process per GPU); (0, 1) otherwise.
    C3R_HOST_SLICE=r/w overrides it: a single process then takes the host slice rank r of w would get on this node — how the
    8-GPU host budget is tried out on a 1-GPU box (tools/host_slice.py)."""
    import os
    ov = os.environ.get("C3R_HOST_SLICE")
    if ov:
        r, _, w = ov.partition("/")
        return int(r), max(1, int(w))
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    lw = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    return lr, max(1, lw)


_BASE_AFFINITY = None


def _cpu_lists():
    """[(numa node, [cpus])] from /sys; one pseudo-node with every allowed CPU when the topology is not exposed."""
    import glob
    import os
    global _BASE_AFFINITY
    if _BASE_AFFINITY is None:
        # the CPUs the process started with: host_budget(apply=True) narrows the mask, and a second call in the same process (a driver
        # that runs several samples) must slice what the process was given, not its own earlier slice (32 -> 4 -> 1 CPUs otherwise)
        _BASE_AFFINITY = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    allowed = _BASE_AFFINITY
    nodes = []
    for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*"), key=lambda x: int(x.rsplit("node", 1)[1])):
        try:
            txt = open(os.path.join(d, "cpulist")).read().strip()
        except OSError:
            continue
        cpus = []
        for part in txt.split(","):
            if not part:
                continue
            a, _, b = part.partition("-")
            cpus.extend(range(int(a), int(b or a) + 1))
        cpus = [c for c in cpus if c in set(allowed)]
        if cpus:
            nodes.append((int(d.rsplit("node", 1)[1]), cpus))
    return nodes or [(0, allowed)]


def _gpu_numa_node(local_rank):
    """NUMA node of the local_rank-th GPU (KFD topology order = HIP device order), or None."""
    import glob
    import os
    gpus = []
    for d in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*"), key=lambda x: int(os.path.basename(x))):
        try:
            props = dict(l.split(None, 1) for l in open(os.path.join(d, "properties")).read().splitlines() if " " in l)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0:          # (CPU nodes have no SIMDs)
            gpus.append(d)
    if local_rank >= len(gpus):
        return None
    try:
        links = glob.glob(os.path.join(gpus[local_rank], "io_links", "*", "properties"))
        for fn in links:
            props = dict(l.split(None, 1) for l in open(fn).read().splitlines() if " " in l)
            to = int(props.get("node_to", "-1"))
            if 0 <= to < 16 and int(props.get("type", "0")) == 2:   # PCIe link to a CPU node
                return to
    except (OSError, ValueError):
        pass
    return None


def host_budget(local_rank=None, local_world_size=None, apply=False):
    """Host side of one rank on a node shared with its peers (one process per GPU): the CPUs this rank may use and its thread
    count.  Every rank gets an equal share of its GPU's NUMA node when the topology says which node that is, else an equal slice
    of all CPUs (rank order = CPU order, which matches the usual GPU-to-socket wiring).  Without this, 8 ranks each spawn the
    32 decode + fetch threads a single-process run would (~400 runnable threads on 256 hardware threads).
    apply=True pins the process (sched_setaffinity) and exports C3R_THREADS / OMP_NUM_THREADS unless they are set already.
    Returns (n_threads, cpus)."""
    import os
    if local_rank is None or local_world_size is None:
        local_rank, local_world_size = local_world()
    nodes = _cpu_lists()
    every = [c for _n, cpus in nodes for c in cpus]
    cpus = None
    if local_world_size > 1:
        # (C3R_HOST_SLICE pretends to be one of N ranks on a box with fewer GPUs: the plain equal slice)
        node = None if os.environ.get("C3R_HOST_SLICE") else _gpu_numa_node(local_rank)
        by_node = dict(nodes)
        if node in by_node and len(nodes) > 1:
            # ranks whose GPUs hang off the same node share that node's CPUs evenly
            peers = [r for r in range(local_world_size) if _gpu_numa_node(r) == node]
            k = peers.index(local_rank) if local_rank in peers else 0
            share = by_node[node]
            n = max(1, len(share) // max(1, len(peers)))
            cpus = share[k * n:(k + 1) * n] or share
        else:
            n = max(1, len(every) // local_world_size)
            cpus = every[local_rank * n:(local_rank + 1) * n] or every
    else:
        cpus = every
    n_threads = max(1, min(32, len(cpus)))
    if apply:
        if local_world_size > 1 and hasattr(os, "sched_setaffinity"):
            try:
                os.sched_setaffinity(0, cpus)
            except OSError:
                pass
        os.environ.setdefault("C3R_THREADS", str(n_threads))
        os.environ.setdefault("OMP_NUM_THREADS", str(n_threads))
        # (the native libraries take their default thread counts from the affinity mask set above: BGZF inflate, compression,
        # FASTA slices, row decode — libc3r_io.so / libc3r.so, usable_cpus())
    return n_threads, cpus
