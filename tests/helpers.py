"""Shared helpers for the parity tests: run the CPU oracle and the HIP engine on the same inputs."""
import numpy as np

from oracle import oracle as orc

CTG = "chr20"


def oracle_chunk(rs, ref, ref_start, ctg_start, ctg_end, channels=18, lbed=None, **pk):
    """Oracle A1..A5 for one chunk.  Returns dict(rows, lines, X, depth)."""
    es, ee = max(1, ctg_start - 33), ctg_end + 33
    rows = orc.mpileup(rs.reads, rs.cigar, rs.seq, CTG, es, ee, min_mq=pk.pop("min_mq", 5), excl_flags=pk.pop("excl_flags", 2316),
                       bed=lbed, with_hp=(channels == 30), max_depth=pk.pop("max_depth", 8000), compat=pk.pop("mpileup_compat", 0))
    P = orc.make_params(phased=(channels == 30), **pk)
    # the reference upper-cases the whole slice when it loads it ("uppercase for masked sequences", shared/utils.py:186-187)
    lines = orc.create_tensor(rows, CTG, ref.upper(), ref_start, P)
    X, depth = orc.batch_from_lines(lines, channels)
    return dict(rows=rows, lines=lines, X=X, depth=depth)


def engine_chunk(eng, rs, ref, ref_start, ctg_start, ctg_end):
    """HIP A1..A5 for one chunk through the C-ABI.  Returns dict(lines, X, raw, sites, tokens)."""
    from clair3_rna_amd import altinfo
    eng.load_reads(rs)
    eng.set_reference(ref_start, ref)
    n = eng.scan(ctg_start, ctg_end)
    raw = eng.tensors(rescaled=False)
    X = eng.tensors(rescaled=True)
    sites, toks = eng.sites(), eng.tokens()
    lines = altinfo.format_lines(CTG, sites, raw, toks, rs, ref.upper(), ref_start, padins=eng.pad_insertions())
    return dict(n=n, lines=lines, X=X, raw=raw, sites=sites, tokens=toks)


def first_diff(a, b):
    for i, (x, y) in enumerate(zip(a, b)):
        if x != y:
            fx, fy = x.split("\t"), y.split("\t")
            for k in range(min(len(fx), len(fy))):
                if fx[k] != fy[k]:
                    if k == 3:
                        vx, vy = np.array(fx[3].split(), int), np.array(fy[3].split(), int)
                        d = np.nonzero(vx != vy)[0]
                        return "line %d pos %s field 3 idx %s got %s exp %s" % (i, fx[1], d[:8], vx[d[:8]], vy[d[:8]])
                    return "line %d pos %s/%s field %d: %r vs %r" % (i, fx[1], fy[1], k, fx[k][:120], fy[k][:120])
    return "length %d vs %d" % (len(a), len(b))


def merge_readsets(a, b):
    """ReadSet holding the reads of both, in position order (stable: a's reads before b's at equal positions)."""
    from clair3_rna_amd.reads import ReadSet
    rb = b.reads.copy()
    rb["cigar_off"] += len(a.cigar)
    rb["seq_off"] += len(a.seq)
    reads = np.concatenate([a.reads, rb])
    order = np.argsort(reads["pos"], kind="stable")
    return ReadSet(reads[order], np.concatenate([a.cigar, b.cigar]), np.concatenate([a.seq, b.seq]))


def indel_next_to_indel_reads(ref, pos0, n=6, seed=7):
    """Hand-made reads at 0-based pos0 whose columns differ between the two samtools printers: `40M2I1D40M` (an insertion with a
    deletion right behind it) and `40M1I1P1I40M` (a pad inside the run of I ops), both strands."""
    import random
    from clair3_rna_amd.reads import ReadSet
    rng = random.Random(seed)
    recs = []
    for k in range(n):
        flag = 16 if k % 2 else 0
        left, right = ref[pos0:pos0 + 40].upper(), ref[pos0 + 41:pos0 + 81].upper()
        recs.append(dict(pos=pos0, cigar="40M2I1D40M", seq=left + "TG" + right, flag=flag))
        recs.append(dict(pos=pos0, cigar="40M1I1P1I40M", seq=left + "CA" + ref[pos0 + 40:pos0 + 80].upper(), flag=flag))
    return ReadSet.from_records(recs)


def fake_samtools(path, version):
    """An executable that answers `--version` like samtools <version> (the drivers' --mpileup_compat auto asks it)."""
    import os
    with open(path, "w") as f:
        f.write("#!/bin/sh\necho 'samtools %s'\necho 'Using htslib %s'\n" % (version, version))
    os.chmod(path, 0o755)
    return path
