"""Deterministic synthetic long-read RNA alignments — small-scale Python generator.

Used by the parity tests and smoke(); bench.py uses the C++ generator in csrc/synth.cpp for the
chr20-scale workload.  Reads carry N (intron) ops, mismatches, insertions, deletions, soft clips,
both strands, a MAPQ mix, filtered flags and optional HP tags, so every CIGAR/flag branch of the
tensor-build kernel is exercised.
"""
import random

import numpy as np

from .reads import ReadSet

BASES = "ACGT"


def random_reference(length, seed):
    rng = np.random.RandomState(seed)
    return "".join(np.array(list(BASES))[rng.randint(0, 4, size=length)])


def make_transcripts(rng, ref_len, n_genes, exon_mean=150, exon_sd=80, intron_lo=100, intron_hi=3000, start=200):
    """Each transcript: list of (exon_start0, exon_end0) on the reference, left to right."""
    genes = []
    pos = start
    for _ in range(n_genes):
        n_ex = rng.randint(2, 8)
        exons = []
        for e in range(n_ex):
            L = max(30, int(rng.gauss(exon_mean, exon_sd)))
            if pos + L >= ref_len - 200:
                break
            exons.append((pos, pos + L))
            pos += L
            if e + 1 < n_ex:
                pos += int(np.exp(rng.uniform(np.log(intron_lo), np.log(intron_hi))))
        if len(exons) >= 1:
            genes.append(exons)
        pos += rng.randint(50, 3000)
        if pos >= ref_len - 500:
            break
    return genes


def simulate_reads(ref, genes, depth, seed, platform="ont", phased=False, mean_len=900, variants=True,
                   err_mismatch=None, err_ins=None, err_del=None):
    """Return (ReadSet, truth dict).  depth = mean exonic depth per gene (scaled by a per-gene factor)."""
    rng = random.Random(seed)
    if platform == "ont":
        em, ei, ed = 0.03, 0.015, 0.025
    else:
        em, ei, ed = 0.003, 0.001, 0.001
    em = em if err_mismatch is None else err_mismatch
    ei = ei if err_ins is None else err_ins
    ed = ed if err_del is None else err_del
    records = []
    truth = {}
    for exons in genes:
        tx = []   # transcript coordinate -> reference position
        for a, b in exons:
            tx.extend(range(a, b))
        tlen = len(tx)
        # per-gene variants in transcript coordinates: kind, allele fraction
        var = {}
        if variants:
            for i in range(tlen):
                r = rng.random()
                if r < 1 / 1000.0:
                    var[i] = ("snp", 0.5, rng.choice([b for b in BASES if b != ref[tx[i]]]))
                elif r < 1 / 1000.0 + 1 / 3000.0:
                    var[i] = ("snp", 1.0, rng.choice([b for b in BASES if b != ref[tx[i]]]))
                elif r < 1 / 1000.0 + 1 / 3000.0 + 1 / 8000.0:
                    var[i] = (rng.choice(["ins", "del"]), 0.5, rng.randint(1, 3))
                elif r < 1 / 1000.0 + 1 / 3000.0 + 1 / 8000.0 + 1 / 5000.0 and ref[tx[i]] == "A":
                    var[i] = ("edit", rng.uniform(0.1, 0.3), "G")
        for i, v in var.items():
            truth[tx[i] + 1] = v
        level = depth * np.exp(rng.gauss(0, 0.4))
        n_reads = max(1, int(level * tlen / float(min(mean_len, tlen))))
        for _ in range(n_reads):
            L = int(np.exp(rng.gauss(np.log(mean_len), 0.6))) if platform == "ont" else int(rng.gauss(2500, 800))
            L = max(200, min(8000, L))
            L = min(L, tlen)
            s = rng.randint(0, tlen - L)
            hap = rng.randint(0, 1)
            rev = rng.random() < 0.5
            ops = []      # list of [op, len]
            seq = []

            def add(op, n=1):
                if n <= 0:
                    return
                if ops and ops[-1][0] == op:
                    ops[-1][1] += n
                else:
                    ops.append([op, n])

            i = s
            first = True
            while i < s + L:
                if not first and tx[i] != tx[i - 1] + 1:
                    add("N", tx[i] - tx[i - 1] - 1)
                first = False
                v = var.get(i)
                base = ref[tx[i]]
                if v is not None:
                    if v[0] == "snp" and (v[1] >= 1.0 or hap == 1):
                        base = v[2]
                    elif v[0] == "edit" and rng.random() < v[1]:
                        base = v[2]
                r = rng.random()
                if r < ed and i > s and i + 1 < s + L and tx[i] == tx[i - 1] + 1:
                    n = 1
                    while rng.random() < 0.4 and n < 10 and i + n + 1 < s + L and tx[i + n] == tx[i + n - 1] + 1:
                        n += 1
                    add("D", n)
                    i += n
                    continue
                if r < ed + em:
                    base = rng.choice([b for b in BASES if b != base])
                elif r < ed + em + 0.002:
                    base = "N"
                seq.append(base)
                add("M")
                if v is not None and v[0] == "del" and hap == 1 and i + v[2] + 1 < s + L and \
                        all(tx[i + k + 1] == tx[i + k] + 1 for k in range(v[2])):
                    add("D", v[2])
                    i += v[2] + 1
                    continue
                if (v is not None and v[0] == "ins" and hap == 1) or rng.random() < ei:
                    n = v[2] if (v is not None and v[0] == "ins" and hap == 1) else 1
                    if not (v is not None and v[0] == "ins" and hap == 1):
                        while rng.random() < 0.4 and n < 10:
                            n += 1
                    if i + 1 < s + L:
                        for _k in range(n):
                            seq.append(rng.choice(BASES))
                        add("I", n)
                i += 1
            # a CIGAR must start and end on M for a sane aligner; trim leading/trailing non-M
            while ops and ops[0][0] != "M":
                op, n = ops.pop(0)
                if op == "I":
                    seq = seq[n:]
            while ops and ops[-1][0] != "M":
                op, n = ops.pop()
                if op == "I":
                    seq = seq[:len(seq) - n]
            if not ops:
                continue
            pos0 = tx[s]
            # leading ops removed may have shifted the start: recompute from the first kept M
            # (leading D/N trimmed => start moves right)
            # simple and exact: walk tx until the first base that produced an M
            # (we only trim I at the ends in practice because D/N need both neighbours)
            if rng.random() < 0.10:
                n = rng.randint(5, 50)
                seq = [rng.choice(BASES) for _ in range(n)] + seq
                ops.insert(0, ["S", n])
            if rng.random() < 0.10:
                n = rng.randint(5, 50)
                seq = seq + [rng.choice(BASES) for _ in range(n)]
                ops.append(["S", n])
            if rng.random() < 0.03:
                ops.insert(0, ["H", rng.randint(10, 200)])
            r = rng.random()
            mapq = 60 if r < 0.92 else (rng.randint(0, 4) if r < 0.97 else rng.randint(5, 59))
            flag = 16 if rev else 0
            r = rng.random()
            if r < 0.02:
                flag |= 256
            elif r < 0.04:
                flag |= 2048
            elif r < 0.045:
                flag |= 8
            elif r < 0.055:
                flag |= 1024
            elif r < 0.06:
                flag |= 512
            hp = 0
            if phased:
                hp = (hap + 1) if rng.random() < 0.9 else 0
            records.append(dict(pos=pos0, cigar="".join("%d%s" % (n, op) for op, n in ops), seq="".join(seq),
                                flag=flag, mapq=mapq, hp=hp))
    return ReadSet.from_records(records), truth


def small_case(seed=1, ref_len=60000, n_genes=12, depth=20, platform="ont", phased=False, **kw):
    """Reference + reads for a small region; returns (ref_seq, ReadSet, truth)."""
    ref = random_reference(ref_len, seed)
    rng = random.Random(seed + 17)
    genes = make_transcripts(rng, ref_len, n_genes, **{k: v for k, v in kw.items() if k in ("intron_hi", "intron_lo", "exon_mean")})
    rs, truth = simulate_reads(ref, genes, depth, seed + 29, platform=platform, phased=phased,
                               **{k: v for k, v in kw.items() if k in ("mean_len", "variants", "err_mismatch", "err_ins", "err_del")})
    return ref, rs, truth


def random_weights(channels, seed=1234, ref_bias=0.0):
    """Seeded random network weights in the blob layout of include/c3r.h (Keras order).
    N(0,0.05) input kernels, N(0,0.02) recurrent kernels, unit forget-gate bias.
    ref_bias > 0 is added to the bias of the zygosity head's hom-ref class: random weights call every candidate a variant, a trained
    model calls a few percent (tools/e2e_full.py --ref_bias: the host stages behind the network at a realistic record rate)."""
    rng = np.random.RandomState(seed)
    parts = []
    for (cin, H) in ((channels, 128), (256, 160)):
        for _d in range(2):
            parts.append(rng.normal(0, 0.05, size=(cin, 4 * H)))
            parts.append(rng.normal(0, 0.02, size=(H, 4 * H)))
            b = np.zeros(4 * H)
            b[H:2 * H] = 1.0
            parts.append(b)
    parts.append(rng.normal(0, 0.02, size=(33 * 320, 128)))
    parts.append(rng.normal(0, 0.05, size=128))
    for _ in range(2):
        parts.append(rng.normal(0, 0.08, size=(128, 128)))
        parts.append(rng.normal(0, 0.05, size=128))
    parts.append(rng.normal(0, 0.3, size=(128, 21)))
    parts.append(rng.normal(0, 0.1, size=21))
    parts.append(rng.normal(0, 0.3, size=(128, 3)))
    zb = rng.normal(0, 0.1, size=3)
    zb[0] += ref_bias
    parts.append(zb)
    return np.concatenate([p.reshape(-1) for p in parts]).astype(np.float32)


# ----------------------------------------------------------------------------- chr20-scale generator (C++)
import ctypes as _C
import os as _os

_SYNTH_LIB = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "libc3r_synth.so")


class _SynthParams(_C.Structure):
    _fields_ = [("contig_len", _C.c_int64), ("seed", _C.c_uint64), ("depth", _C.c_double), ("expressed_frac", _C.c_double),
                ("platform", _C.c_int32), ("phased", _C.c_int32), ("intron_lo", _C.c_double), ("intron_hi", _C.c_double),
                ("region_start", _C.c_int64), ("region_end", _C.c_int64), ("expr_sigma", _C.c_double), ("max_level", _C.c_double)]


class _SynthResult(_C.Structure):
    _fields_ = [("ref", _C.c_void_p), ("ref_len", _C.c_int64), ("reads", _C.c_void_p), ("n_reads", _C.c_int64),
                ("cigar", _C.c_void_p), ("n_cigar", _C.c_int64), ("seq", _C.c_void_p), ("n_seq", _C.c_int64),
                ("n_exonic", _C.c_int64), ("n_genes", _C.c_int64), ("owner", _C.c_void_p)]


CHR20_LEN = 64444167          # GRCh38 chr20
SEED = 20240422


def generate_contig(contig_len=CHR20_LEN, seed=SEED, depth=20.0, expressed_frac=0.03, platform="ont", phased=False,
                    intron_lo=100.0, intron_hi=100000.0, region=None, expr_sigma=0.0, max_level=0.0):
    """Synthetic contig + alignments at BASELINE.json config scale.  Returns (ref_bytes, ReadSet, info).
    expr_sigma > 0: log-normal gene expression with that sigma and mean `depth` (2.3: four to five decades — a few loci in the thousands,
    a long tail of one-to-three-read islands); max_level caps a gene's level."""
    from .reads import READ_DTYPE
    if not _os.path.exists(_SYNTH_LIB):
        raise ImportError("libc3r_synth.so missing — run __graft_entry__.build()")
    L = _C.CDLL(_SYNTH_LIB)
    L.c3r_synth_generate.argtypes = [_C.POINTER(_SynthParams), _C.POINTER(_SynthResult)]
    L.c3r_synth_free.argtypes = [_C.POINTER(_SynthResult)]
    p = _SynthParams(contig_len, seed, depth, expressed_frac, 0 if platform == "ont" else 1, int(phased), intron_lo, intron_hi,
                     region[0] if region else 0, region[1] if region else 0, float(expr_sigma), float(max_level))
    r = _SynthResult()
    rc = L.c3r_synth_generate(_C.byref(p), _C.byref(r))
    if rc != 0:
        raise RuntimeError("c3r_synth_generate failed: %d" % rc)
    try:
        ref = _C.string_at(r.ref, r.ref_len)
        reads = np.frombuffer(_C.string_at(r.reads, r.n_reads * 32), dtype=READ_DTYPE).copy()
        cigar = np.frombuffer(_C.string_at(r.cigar, r.n_cigar * 4), dtype=np.uint32).copy()
        seq = np.frombuffer(_C.string_at(r.seq, r.n_seq), dtype=np.uint8).copy()
        info = dict(n_reads=int(r.n_reads), n_exonic=int(r.n_exonic), n_genes=int(r.n_genes))
    finally:
        L.c3r_synth_free(_C.byref(r))
    return ref, ReadSet(reads, cigar, seq), info
