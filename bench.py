#!/usr/bin/env python3
"""bench.py — candidate sites/sec (tensor build + inference) on synthetic ONT dRNA004 chr20 ~20x.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full pass of the hot path over one synthetic chr20 (BASELINE.json configs[1]), as SURVEY.md 8(d) defines
the metric: from the contig's aligned-read records RESIDENT IN HOST MEMORY (flat c3r_read_t records, BAM-encoded CIGARs, 4-bit
bases — what a BAM reader hands over, where the reference starts `samtools mpileup`, src/create_tensor_pileup.py:436-451) to the
[n, 24] probabilities on the host: upload + the position-binned pile table built on the device (c3r_load_reads), tensor
build over the 13 five-megabase chunks (shared/param_p.py:91: CIGAR walk -> counts -> candidates -> windows), network forward,
probabilities back.  Only the reference sequence and the weights are resident in HBM before the timed region (they do not change
from pass to pass).  The rate with the read tables already prepared on the device is reported beside it as `resident_inputs`.  Multi-GPU: every rank owns its own chr20-sized contig (weak scaling, the
reference shards by contig/chunk with no exchange step: run_clair3_rna:681-706); no data-path collective.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

CHUNK = 5000000            # shared/param_p.py:91 CHUNK_SIZE
# algorithmic work per emitted candidate
def flop_per_site(channels=18):
    return {"k_lstm1": 2.0 * (channels + 128) * 512 * 33 * 2, "k_lstm2": 2.0 * (256 + 160) * 640 * 33 * 2,
            "k_fc4": 2.0 * 10560 * 128, "k_heads": 2.0 * (128 * 256 + 128 * 24)}


FLOP_PER_SITE = flop_per_site(18)
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16/f16 MFMA peak (never the 2:1-sparse figure)
PEAK_HBM_GBPS = 8000.0
NET_KERNELS = ("k_lstm1", "k_lstm2", "k_fc4", "k_heads")


def k1_algorithmic_bytes(rs, centres, channels, min_mq=5, excl_flags=2316):
    """ALGORITHMIC bytes of the tensor build for these candidates, SURVEY.md 8(d)'s formula evaluated on the data (not its worked
    example): per candidate window [c-16, c+16], over the reads whose span overlaps it, 16 B of header + 4 B per CIGAR op that
    intersects the window (at least one) + half a byte per query base consumed inside it, + 33 B of reference; out: 33 * C * 4 B of
    int32 window + 48 B of site record.  Headers are counted for every read (the filters have to look at them), ops and bases for
    the reads that pass them.  (The per-read round-up of the half bytes is taken as its mean, a quarter byte per read.)
    centres: 1-based positions.  Returns total bytes."""
    r = rs.reads
    n = len(r)
    if n == 0 or len(centres) == 0:
        return 0
    cig = rs.cigar.astype(np.int64)
    op, ln = cig & 15, cig >> 4
    rid = np.repeat(np.arange(n), r["n_cigar"].astype(np.int64))
    # (the generator lays the CIGARs out back to back in read order)
    assert int(r["cigar_off"][0]) == 0 and np.all(np.diff(r["cigar_off"].astype(np.int64)) == r["n_cigar"][:-1].astype(np.int64))
    ref_len = np.where((op == 0) | (op == 2) | (op == 3) | (op == 7) | (op == 8), ln, 0)
    cum = np.cumsum(ref_len) - ref_len
    first = r["cigar_off"].astype(np.int64)
    start = r["pos"].astype(np.int64)[rid] + cum - cum[first][rid]
    end = start + ref_len
    rend = np.zeros(n, np.int64)
    np.maximum.at(rend, rid, end)
    rend = np.maximum(rend, r["pos"].astype(np.int64))
    flag = r["flag"].astype(np.int64)
    ok = ((flag & excl_flags) == 0) & ((flag & 4) == 0) & ~(((flag & 1) != 0) & ((flag & 2) == 0)) & (r["mapq"] >= min_mq) & (rend > r["pos"])
    c0 = np.asarray(centres, np.int64) - 1
    w0, w1 = c0 - 16, c0 + 17
    pos_sorted, end_sorted = np.sort(r["pos"].astype(np.int64)), np.sort(rend)
    n_hdr = np.searchsorted(pos_sorted, w1, "left") - np.searchsorted(end_sorted, w0, "right")
    keep = ok[rid] & (op != 5) & (op != 6) & (ln > 0)
    s_k, e_k = start[keep], end[keep]
    n_ops = np.searchsorted(np.sort(s_k), w1, "left") - np.searchsorted(np.sort(e_k), w0, "right")
    n_pass = np.searchsorted(np.sort(r["pos"].astype(np.int64)[ok]), w1, "left") - np.searchsorted(np.sort(rend[ok]), w0, "right")
    n_ops = np.maximum(n_ops, n_pass)

    def bases_below(x, s_, e_):          # query bases of M-like ops that sit on reference positions < x
        o1, o2 = np.argsort(s_, kind="stable"), np.argsort(e_, kind="stable")
        ss, es = s_[o1], e_[o2]
        a = np.searchsorted(ss, x, "left"); b = np.searchsorted(es, x, "right")
        csa = np.concatenate([[0], np.cumsum(ss)]); csb = np.concatenate([[0], np.cumsum(s_[o2])]); clb = np.concatenate([[0], np.cumsum((e_ - s_)[o2])])
        return clb[b] + x * (a - b) - (csa[a] - csb[b])
    m = ok[rid] & ((op == 0) | (op == 7) | (op == 8))
    bases = bases_below(w1, start[m], end[m]) - bases_below(w0, start[m], end[m])
    ins = ok[rid] & (op == 1)
    si = np.sort(start[ins]); ci = np.concatenate([[0], np.cumsum(ln[ins][np.argsort(start[ins], kind="stable")])])
    bases = bases + ci[np.searchsorted(si, w1, "left")] - ci[np.searchsorted(si, w0, "right")]
    total_in = 16 * n_hdr.sum() + 4 * n_ops.sum() + 0.5 * bases.sum() + 0.25 * n_pass.sum() + 33 * len(c0)
    total_out = (33 * channels * 4 + 48) * len(c0)
    return float(total_in + total_out)


def phase1_bytes(rs, channels, min_mq=5, excl_flags=2316):
    """SURVEY.md 8(d), phase 1 — the column scan the reference runs over every position of the pileup stream (src/create_tensor_pileup.py:520-560),
    whether or not a candidate comes of it: the input records (headers + CIGAR ops + packed bases, as handed over) + C * 4 bytes of counts per
    position.  Two position counts: `aligned` — positions where at least one passing read has an aligned base or a deletion (the only ones whose
    counts can be non-zero; what `achieved` is priced on: the smaller, stricter figure) — and `covered`, which adds the positions that reads only
    span with a ref-skip (mpileup prints a row of '>' / '<' there and the reference parses it).  Returns (bytes on aligned, aligned, covered)."""
    r = rs.reads
    n = len(r)
    in_bytes = int(rs.reads.nbytes + rs.cigar.nbytes + rs.seq.nbytes)
    if n == 0:
        return float(in_bytes), 0, 0
    cig = rs.cigar.astype(np.int64)
    op, ln = cig & 15, cig >> 4
    ref_len = np.where((op == 0) | (op == 2) | (op == 3) | (op == 7) | (op == 8), ln, 0)
    csum = np.concatenate([[0], np.cumsum(ref_len)])
    first = r["cigar_off"].astype(np.int64)
    ncig = r["n_cigar"].astype(np.int64)
    span = csum[first + ncig] - csum[first]
    pos = r["pos"].astype(np.int64)
    end = pos + span
    flag = r["flag"].astype(np.int64)
    ok = ((flag & excl_flags) == 0) & ((flag & 4) == 0) & ~(((flag & 1) != 0) & ((flag & 2) == 0)) & (r["mapq"] >= min_mq) & (end > pos)

    def union(p, e):
        if len(p) == 0:
            return 0
        o = np.argsort(p, kind="stable")
        p, e = p[o], e[o]
        reach = np.maximum.accumulate(e)
        prev = np.concatenate([[p[0]], reach[:-1]])
        return int(np.sum(np.maximum(e - np.maximum(p, prev), 0)))
    covered = union(pos[ok], end[ok])
    # aligned pieces: M / = / X / D ops of the passing reads (the generator lays the CIGARs out back to back in read order)
    rid = np.repeat(np.arange(n), ncig)
    start = pos[rid] + (csum[:-1] - csum[first][rid])
    keep = ok[rid] & ((op == 0) | (op == 2) | (op == 7) | (op == 8)) & (ln > 0)
    aligned = union(start[keep], start[keep] + ln[keep])
    return float(in_bytes + 4 * channels * aligned), aligned, covered


def chunk_list(contig_len, chunk=CHUNK):
    """(ctg_start, ctg_end) per chunk, the arithmetic of src/create_tensor_pileup.py:380-392."""
    n = (contig_len + chunk - 1) // chunk
    size = contig_len // n + 1 if contig_len % n else contig_len // n
    return [(size * i, size * i + size) for i in range(n)]


_CPU = {}


def _cpu_region(k):
    """Text stages of one region in one worker process, as in one `samtools mpileup | pypy create_tensor` chunk job of the
    reference (run_clair3_rna:678-708): text mpileup rows, parse, window driver -> the int32 batch, left in a scratch file."""
    orc, rs, ref, size, contig_len = _CPU["orc"], _CPU["rs"], _CPU["ref"], _CPU["size"], _CPU["contig_len"]
    beg, end = k * size, min((k + 1) * size, contig_len)
    rows = orc.mpileup(rs.reads, rs.cigar, rs.seq, "chr20", max(1, beg - 33), end + 33)
    rstart = max(1, beg - 1000)
    lines = orc.create_tensor(rows, "chr20", ref[rstart - 1:end + 1000].decode(), rstart, orc.make_params())
    X, _ = orc.batch_from_lines(lines, 18)
    if len(X):
        np.save(os.path.join(_CPU["dir"], "%06d.npy" % k), X)
    return len(X)


def cpu_baseline(rs, ref, weights, contig_len):
    """The oracle (a port of the reference pipeline) on the host's cores, bounded: the text stages (mpileup rows -> parse -> window
    driver) one worker process per 250-kb region on every core, as the reference fans its chunk jobs out with GNU parallel; then
    the fp32 network over all their candidates on every core (OpenMP over sites — what many concurrent single-threaded
    call_variants processes amount to, without their load imbalance)."""
    import multiprocessing as mp
    import shutil
    import tempfile
    from oracle import oracle as orc
    cores = os.cpu_count() or 1
    size = 125000
    n_regions = min((contig_len + size - 1) // size, 2 * cores)        # bounded: two short regions per core (~4 s on the 256-thread box)
    workers = min(cores, n_regions)
    scratch = tempfile.mkdtemp(prefix="c3r_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    _CPU.update(orc=orc, rs=rs, ref=ref, size=size, contig_len=contig_len, dir=scratch)
    try:
        t0 = time.perf_counter()
        with mp.get_context("fork").Pool(workers) as pool:              # fork: the data is inherited, nothing is exec'd
            counts = pool.map(_cpu_region, range(n_regions), chunksize=1)
        t1 = time.perf_counter()
        X = [np.load(os.path.join(scratch, f)) for f in sorted(os.listdir(scratch))]
        n_cpu = int(sum(counts))
        n_net = min(n_cpu, 120 * cores)                                  # bounded: ~4 s of network on this box
        if n_net:
            orc.forward(weights, np.concatenate(X)[:n_net])
        t2 = time.perf_counter()
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    if not n_cpu:
        return None
    per_site = (t1 - t0) / n_cpu + (t2 - t1) / n_net                     # seconds per site through both stages
    return dict(value=round(1.0 / per_site, 1), unit="sites/s", cores=cores, kind="port",
                note="(the sample is bounded to ~10 s of text stages + ~5 s of network since round 5 — 256 short regions instead of whole chunks —, so the "
                     "figure is not comparable with rounds 1-4's 5.5 k) "
                     "a port of the reference pipeline, not a tuned CPU code: the network leg is a scalar fp32 triple loop "
                     "(%.1f GFLOP/s per core here; TensorFlow/Eigen reach 10-100x that), so this is a baseline to read beside the "
                     "number, never a speed-up denominator" % (47.8e6 * n_net / max(t2 - t1, 1e-9) / cores / 1e9),
                sample="chr20:1-%d of the same synthetic contig: text mpileup + parse + window driver in %d worker processes, one 125-kb "
                       "region each (%d candidates, %.1f s), then the fp32 network on %d cores with OpenMP over the first %d of them (%.1f s); "
                       "value = 1 / (s per site of stage 1 + s per site of stage 2)"
                       % (min(n_regions * size, contig_len), workers, n_cpu, t1 - t0, cores, n_net, t2 - t1))


def rooflines(kernels, n_prof, rs, site_pos, channels, precision):
    """One profiled pass (HIP-event kernel statistics of c3r_set_profiling) -> (roofline of the dominant kernel, the tensor build against
    ITS roofline, the two halves' rates).  `kernels` loses its "h2d_reads" entry (PCIe time, reported beside the kernels)."""
    fps = flop_per_site(channels)
    h2d_ms = kernels.pop("h2d_reads", {"total_ms": 0.0})["total_ms"]          # (the upload of the pass's records: PCIe time, reported beside the kernels)
    dom = max(kernels, key=lambda k: kernels[k]["total_ms"])
    st = kernels[dom]
    avg_ms = st["total_ms"] / st["launches"]
    if dom in fps:
        per_site = fps[dom]
        if dom == "k_lstm2" and precision != "f32":
            per_site += fps["k_fc4"]          # the L4 dense layer is fused into the layer-2 kernel
        flops_per_launch = per_site * n_prof / st["launches"]
        ach = flops_per_launch / (avg_ms * 1e-3) / 1e12
        if precision in ("f16+f8", "auto"):
            # algorithmic flops against the dense f16 peak; the kernel executes one f16 product plus two fp8 products (on the
            # block-scaled pipe at twice the f16 rate) per algorithmic product = 2 f16-equivalents of matrix-pipe time
            roofline = dict(kernel=dom, bound="mfma", achieved=round(ach, 2), peak=PEAK_F16_MFMA_TFLOPS, unit="TFLOP/s",
                            frac=round(ach / PEAK_F16_MFMA_TFLOPS, 4), traffic=None, avg_launch_ms=round(avg_ms, 4),
                            launches=st["launches"], executed_f16_equiv_tflops=round(2 * ach, 1),
                            executed_frac=round(2 * ach / PEAK_F16_MFMA_TFLOPS, 4),
                            note="f16 main term + both correction terms as one block-scaled fp8 MFMA (K = 64), fp32 accumulation")
        elif precision == "f16x3":
            # ALGORITHMIC flops against the dense f16 MFMA peak.  The kernel executes 3 f16 products per algorithmic
            # product (hi*hi + hi*lo + lo*hi), so matrix-pipe utilisation is 3x `frac`.
            roofline = dict(kernel=dom, bound="mfma", achieved=round(ach, 2), peak=PEAK_F16_MFMA_TFLOPS, unit="TFLOP/s",
                            frac=round(ach / PEAK_F16_MFMA_TFLOPS, 4), traffic=None, avg_launch_ms=round(avg_ms, 4),
                            launches=st["launches"], executed_tflops=round(3 * ach, 1),
                            executed_frac=round(3 * ach / PEAK_F16_MFMA_TFLOPS, 4),
                            vs_f32_mfma_peak=round(ach / PEAK_F32_MFMA_TFLOPS, 3),
                            note="split-f16: fp32-equivalent GEMM as 3 f16 MFMAs with fp32 accumulation")
        else:
            roofline = dict(kernel=dom, bound="mfma", achieved=round(ach, 2), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                            frac=round(ach / PEAK_F32_MFMA_TFLOPS, 4), traffic=None, avg_launch_ms=round(avg_ms, 4),
                            launches=st["launches"])
    else:
        roofline = dict(kernel=dom, bound="hbm", achieved=None, peak=PEAK_HBM_GBPS, unit="GB/s", frac=None, traffic=None,
                        avg_launch_ms=round(avg_ms, 4), launches=st["launches"])
    # the two halves on their own (device time of each half's kernels in the profiled pass).  The tensor-build half
    # is everything that is not the network: read preparation (upload excluded: copies are not kernels; reported as h2d), scan, windows, tokens.
    net_ms = sum(v["total_ms"] for k, v in kernels.items() if k in NET_KERNELS)
    k1_ms = sum(v["total_ms"] for k, v in kernels.items()) - net_ms
    prep = ("k_prep_count", "k_prefmax_bins", "k_bin_scan", "k_prep_write", "k_legacy_tables")
    prep_ms = sum(v["total_ms"] for k, v in kernels.items() if k in prep)
    k1_bytes = k1_algorithmic_bytes(rs, site_pos, channels) if n_prof else 0.0
    tb_gbps = k1_bytes / (k1_ms * 1e-3) / 1e9 if k1_ms else 0.0
    stage_rates = dict(tensor_build_sites_per_s=round(n_prof / (k1_ms * 1e-3), 1) if k1_ms else None,
                       inference_sites_per_s=round(n_prof / (net_ms * 1e-3), 1) if net_ms else None,
                       tensor_build_algorithmic_GBps=round(tb_gbps, 1) if k1_ms else None,
                       tensor_build_ms=round(k1_ms, 3), read_preparation_ms=round(prep_ms, 3), inference_ms=round(net_ms, 3))
    p1_bytes, aligned, covered = phase1_bytes(rs, channels)
    p1_gbps = p1_bytes / (k1_ms * 1e-3) / 1e9 if k1_ms else 0.0
    if dom not in fps and roofline["bound"] == "hbm" and k1_ms:
        # the dominant kernel is a tensor-build kernel (deep coverage, few candidates): it IS the column scan, and the scan's bytes — every input
        # record once + C * 4 bytes of counts per covered position (phase 1) — over ITS average launch are its rate
        ach = p1_bytes / (avg_ms * st["launches"] * 1e-3) / 1e9
        roofline.update(achieved=round(ach, 1), frac=round(ach / PEAK_HBM_GBPS, 4),
                        note="a tensor-build kernel dominates this workload: phase-1 bytes (input records + C*4 B per covered position, "
                             "roofline_tensor_build.phase1) over this kernel's time; the build as a whole: roofline_tensor_build")
    # the tensor-build half against ITS roofline (HBM): SURVEY 8(d)'s algorithmic bytes, evaluated on this pass's candidates,
    # over the summed device time of ALL tensor-build kernels of the profiled pass (read preparation included)
    roofline_tb = dict(kernels=sorted(k for k in kernels if k not in NET_KERNELS), bound="hbm",
                       achieved=round(tb_gbps, 1), peak=PEAK_HBM_GBPS, unit="GB/s", frac=round(tb_gbps / PEAK_HBM_GBPS, 4),
                       bytes_per_site=round(k1_bytes / n_prof, 1) if n_prof else None, bytes_per_pass=int(k1_bytes), ms=round(k1_ms, 3),
                       note="bytes: SURVEY 8(d) formula evaluated on the pass's own candidates and reads (bench.k1_algorithmic_bytes)")
    in_bytes = int(rs.reads.nbytes + rs.cigar.nbytes + rs.seq.nbytes)
    roofline_tb["phase1"] = dict(bytes=int(p1_bytes), input_record_bytes=in_bytes, aligned_positions=aligned, covered_positions_incl_ref_skips=covered,
                                 achieved=round(p1_gbps, 1), unit="GB/s", frac=round(p1_gbps / PEAK_HBM_GBPS, 4), ms=round(k1_ms, 3),
                                 bytes_per_aligned_position=round(p1_bytes / aligned, 1) if aligned else None,
                                 frac_on_covered_positions=round((in_bytes + 4 * channels * covered) / (k1_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4) if k1_ms else None,
                                 note="SURVEY 8(d) phase 1, the column scan: input records (headers + CIGAR ops + packed bases) + C*4 B of counts per position that "
                                      "holds an aligned base or a deletion of a passing read, over the summed device time of all tensor-build kernels "
                                      "(bench.phase1_bytes); frac_on_covered_positions prices the ref-skip-only positions too (rows the reference parses, "
                                      "columns this build never forms)")
    h2d_bytes = int(rs.reads.nbytes + rs.cigar.nbytes + rs.seq.nbytes)
    roofline_tb["h2d"] = dict(bytes=h2d_bytes, ms=round(h2d_ms, 3), GBps=round(h2d_bytes / (h2d_ms * 1e-3) / 1e9, 1) if h2d_ms else None, included_in_ms=False,
                              frac_with_h2d=round(k1_bytes / ((k1_ms + h2d_ms) * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4) if k1_ms else None,
                              note="the pass's records go up from page-locked host memory inside c3r_load_reads (PCIe, not HBM): inside every timed step, "
                                   "outside roofline_tensor_build.ms; frac_with_h2d counts it as if it were kernel time")
    return roofline, roofline_tb, stage_rates


def extra_config(name, workload, gen, channels, precision, local_rank, steps=4, params=None):
    """One more BASELINE.json configuration on one GPU, as an ADDITIONAL object of the bench line (never `value`): `steps` passes of the
    whole hot path (host-resident records -> device tables -> tensor build -> network -> probabilities on the host) on one context, then
    one profiled pass for the kernel table and the two rooflines."""
    import torch
    from clair3_rna_amd import capi, synth
    ref, rs, info = synth.generate_contig(**gen)
    contig_len = gen["contig_len"]
    chunks = chunk_list(contig_len)
    w = synth.random_weights(channels)
    rsh = capi.pinned_readset(rs)
    e = capi.Engine(local_rank)
    try:
        e.set_params(channels=channels, **(params or {}))
        e.set_reference(1, ref); e.load_weights(w, channels); e.set_precision(precision)

        def step():
            e.load_reads(rsh)
            e.begin_batch(); n = e.scan_regions(chunks); e.end_batch()
            if n:
                e.infer()
            return n
        step(); step()                                   # buffers sized, clocks up
        e.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        sites = sum(step() for _ in range(steps))
        e.synchronize(); torch.cuda.synchronize()
        el = time.perf_counter() - t0
        e.set_profiling(True)
        step()                         # (the profiler's serial launch order once, unrecorded)
        e.reset_kernel_stats()
        n_prof = step()
        e.set_profiling(False)
        kernels = e.kernel_stats()
        roof, roof_tb, rates = rooflines(kernels, n_prof, rs, e.sites()["pos"] if n_prof else [], channels, precision)
        mode = e.precision()[0]
        # the same passes on TWO contexts, pipelined the way the headline is (one context's uploads and tensor build beside the other's network)
        two = None
        e2 = capi.Engine(local_rank)
        try:
            e2.set_params(channels=channels, **(params or {}))
            e2.set_reference(1, ref); e2.load_weights(w, channels); e2.set_precision(precision)
            pair = (e, e2)

            def run2(k):
                # one host thread per context (the library calls release the GIL): c3r_load_reads and the scan block their caller on read-backs,
                # so a single thread would serialise one context's upload behind the other's scan
                import threading
                tots, errs = [0, 0], []

                def drive(j):
                    try:
                        x = pair[j]
                        for _ in range(k // 2):
                            x.load_reads(rsh)
                            x.begin_batch(); n = x.scan_regions(chunks); x.end_batch()
                            if n:
                                x.infer()
                            tots[j] += n
                    except Exception as ex:          # noqa: BLE001
                        errs.append(ex)
                th = [threading.Thread(target=drive, args=(j,)) for j in range(2)]
                for t_ in th:
                    t_.start()
                for t_ in th:
                    t_.join()
                if errs:
                    raise errs[0]
                return sum(tots)
            run2(2)
            e.synchronize(); e2.synchronize(); torch.cuda.synchronize()
            # (at least 32 steps: the pipeline of two contexts fills with one upload that nothing hides and drains with one context's kernels alone —
            # 2.7 ms of a 500x step's 4.4; over the 8 steps of rounds 5-6 that read as 4.8 ms per step)
            k2 = max(2 * steps, 32)
            t0 = time.perf_counter()
            s2 = run2(k2)
            e.synchronize(); e2.synchronize(); torch.cuda.synchronize()
            el2 = time.perf_counter() - t0
            two = dict(value=round(s2 / el2, 1), ms_per_step=round(1e3 * el2 / k2, 3), steps=k2,
                       reads_per_s=round(info["n_reads"] * k2 / el2, 1), vs_one_context=round((s2 / el2) / (sites / el), 3) if sites else None,
                       host_threads=2)
        except Exception as ex:                       # (an additional figure: never loses the one above)
            two = dict(error=repr(ex)[:200])
        finally:
            e2.close()
    finally:
        e.close()
    return dict(name=name, workload=workload, value=round(sites / el, 1), unit="sites/s", steps=steps, ms_per_step=round(1e3 * el / steps, 3),
                sites_per_step=round(sites / steps, 1), reads=info["n_reads"], reads_per_s=round(info["n_reads"] * steps / el, 1),
                exonic_bp=info["n_exonic"], channels=channels, precision=mode, contig_len=contig_len, streams=1,
                host_and_copies_ms_per_step=round(1e3 * el / steps - sum(v["total_ms"] for v in kernels.values()), 3), two_contexts=two,
                roofline=roof, roofline_tensor_build=roof_tb, stage_rates=rates,
                kernels_ms_per_step={k: round(v["total_ms"], 3) for k, v in sorted(kernels.items())})


def run_strong(args, rank, local_rank, world, one_gpu, emit=True):
    """--scaling strong: BASELINE.json configs[2] — the 24 GRCh38 contigs, sharded by contig over the ranks (LPT by length, the
    reference's fan-out at run_clair3_rna:441-449,681-706), inputs host-resident.  One step = every rank takes ITS contigs from
    host records (reads AND reference: both change from contig to contig) to probabilities; total work is fixed, so the N-rank
    value measures load balance and the host feed, not N private copies.  No data-path collective."""
    import torch
    from clair3_rna_amd import capi, shard, synth
    names = [n for n, _l in shard.GRCH38]
    lens = [max(200000, int(l * args.genome_scale)) for _n, l in shard.GRCH38]
    plan = shard.lpt_assign(lens, world)
    mine = sorted(plan[rank], key=lambda i: -lens[i])         # largest first: the pass ends on the shortest network launch (the one tail nothing hides)
    depth = args.depth if args.depth != 20.0 else 30.0                       # configs[2]: ~30x
    weights = synth.random_weights(18)
    torch.cuda.set_device(local_rank)
    dist, red_dev = None, "cuda"
    if world > 1:
        import torch.distributed as dist
        if one_gpu:
            dist.init_process_group("gloo"); red_dev = "cpu"
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    data = []
    for ci in mine:
        ref, rs, info = synth.generate_contig(contig_len=lens[ci], seed=synth.SEED + 1000 + ci, depth=depth)
        # the reference as the fetcher of call_sample hands it over: an upper-cased uint8 array, used in place (c3r_set_reference_view) — the
        # copying form spends 5-11 ms of the host thread per contig on upper-casing (profiles/r5/strong_1gpu_attribution.txt)
        data.append((ci, capi.pinned_copy(np.frombuffer(ref, dtype=np.uint8)), capi.pinned_readset(rs), chunk_list(lens[ci]), info))
    engs = []
    for _ in range(1 if args.no_overlap else max(1, args.contexts)):
        e = capi.Engine(local_rank)
        e.set_params(); e.load_weights(weights, 18); e.set_precision(args.precision)
        engs.append(e)

    def run_steps(k):
        total, pending, ne = 0, [None] * len(engs), len(engs)
        i = 0
        for _ in range(k):
            for (_ci, ref, rs, chunks, _info) in data:
                j = i % ne; i += 1
                e = engs[j]
                if pending[j]:
                    e.fetch_probs(pending[j])
                e.set_reference(1, ref, upper_view=True)
                e.load_reads(rs)
                e.begin_batch(); n = e.scan_regions(chunks); e.end_batch()
                if n:
                    e.infer(fetch=False)
                pending[j] = n
                total += n
        for j in range(ne):
            if pending[j]:
                engs[j].fetch_probs(pending[j])
        return total

    def barrier():
        for e in engs:
            e.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
    run_steps(1)                                   # buffers sized (largest contig first in every rank's list? no: any order — one untimed pass)
    barrier()
    run_steps(max(args.warmup, 0))
    barrier()
    t0 = time.perf_counter()
    sites = run_steps(args.steps)
    for e in engs:
        e.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    loads = [sum(lens[i] for i in p) for p in plan]
    my_reads = float(sum(info["n_reads"] for (_ci, _r, _rs, _ch, info) in data))
    reads_max, reads_sum = my_reads, my_reads
    # what this rank keeps resident: device memory in use on ITS device while its contexts are alive (with the one-GPU test hook: of all ranks
    # together), and the page-locked host bytes of its inputs
    free_b, total_b = torch.cuda.mem_get_info()
    hbm_used = float(total_b - free_b)
    pinned = float(sum(r.nbytes + rs.reads.nbytes + rs.cigar.nbytes + rs.seq.nbytes for (_ci, r, rs, _ch, _info) in data))
    pinned_max = pinned
    if dist is not None:
        dist.barrier()
        elapsed = shard.reduce_max(dist, elapsed, device=red_dev)
        sites = int(shard.reduce_sum(dist, sites, device=red_dev))
        reads_max, reads_sum = shard.reduce_max(dist, my_reads, device=red_dev), shard.reduce_sum(dist, my_reads, device=red_dev)
        hbm_used = shard.reduce_max(dist, hbm_used, device=red_dev)
        pinned_max = shard.reduce_max(dist, pinned, device=red_dev)
    for e in engs:
        e.close()
    if rank == 0:
        out = {"metric": "candidate sites/sec (tensor build + inference)", "value": round(sites / elapsed, 1), "unit": "sites/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / max(1, args.steps), 3), "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "f16 hi/lo split x3, f32 accumulate (fp32-equivalent)" if args.precision == "f16x3" else args.precision,
               "data": "synthetic",
               "config": {"workload": "synthetic ONT dRNA004 whole genome ~%dx, 24 GRCh38 contigs x %.3g (BASELINE.json configs[2])" % (int(depth), args.genome_scale),
                          "inputs": "host-resident flat records and reference per contig", "contigs": len(names), "genome_bp": int(sum(lens)),
                          "parallelism": "contigs dealt largest-first to %d rank(s), no collective" % world, "lpt_imbalance": round(shard.imbalance(lens, plan), 4),
                          "read_imbalance": round(reads_max / (reads_sum / world), 4) if reads_sum else None,      # max / mean READS per rank (the work), not lengths
                          "contigs_per_rank": [len(p) for p in plan], "bp_per_rank": loads, "sites_per_step": round(sites / max(1, args.steps), 1),
                          "precision": args.precision, "streams": len(engs),
                          "hbm_in_use_bytes_max_rank": int(hbm_used), "pinned_input_bytes_max_rank": int(pinned_max),
                          "host_threads_per_rank": int(os.environ.get("C3R_THREADS", "0")) or None},
               "roofline": None, "cpu_baseline": None}
        if emit:
            print(json.dumps(out), flush=True)
    if dist is not None and emit:
        dist.destroy_process_group()
    return out if rank == 0 else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--depth", type=float, default=20.0)
    ap.add_argument("--contig_len", type=int, default=0, help="default: chr20 (64,444,167)")
    ap.add_argument("--contexts", type=int, default=2, help="engine contexts (HIP streams) whose passes are pipelined on the GPU")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_profile", action="store_true")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default, what the driver runs): every rank owns one chr20-sized contig; strong: the 24 GRCh38 contigs of "
                         "BASELINE.json configs[2] dealt to the ranks largest-first (LPT), total work fixed")
    ap.add_argument("--genome_scale", type=float, default=1.0, help="--scaling strong: shrink every contig by this factor (quick runs)")
    ap.add_argument("--no_resident", action="store_true", help="skip the additional measurement with the read tables already on the device")
    ap.add_argument("--no_f32", action="store_true", help="skip the three additional steps on the fp32 MFMA path (reported as f32_mfma)")
    ap.add_argument("--no_fast", action="store_true", help="skip the additional measurement in precision 'auto' (reported as fast_precision)")
    ap.add_argument("--no_strong", action="store_true", help="skip the additional N = 1 run of the strong-scaling configuration (BASELINE.json configs[2] at "
                                                             "a quarter of every contig's length), reported as strong_1gpu")
    ap.add_argument("--no_extra", action="store_true", help="skip the additional N = 1 runs of BASELINE.json configs[3] (MAS-Seq, 30 channels: phased_1gpu), configs[4] "
                                                            "(500x: stress_500x) and of the 20,000x locus that trips mpileup's depth cap (depth_cap_20000x)")
    ap.add_argument("--no_overlap", action="store_true",
                    help="one context / one stream: tensor build and network strictly back to back")
    ap.add_argument("--precision", choices=["f16x3", "f32", "f16+f8", "auto"], default="f16x3",
                    help="network GEMM arithmetic: split-f16 (fp32-equivalent, default) or fp32 MFMA")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    if torch.cuda.device_count() == 0:        # (counting devices does not initialise the GPU; the CPU baseline below forks first)
        print("bench.py needs an MI355X (no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    # C3R_BENCH_ONE_GPU=1 (test hook for 1-GPU boxes): all ranks share device 0 and rendezvous over gloo, so that the N > 1
    # control flow (sharding by rank, barrier, max-over-ranks timing, summed sites) can be exercised; never set by the driver
    one_gpu = os.environ.get("C3R_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    from clair3_rna_amd import capi, shard, synth
    if world > 1 or os.environ.get("C3R_HOST_SLICE"):
        shard.host_budget(apply=True)         # this rank's share of the node's cores (its GPU's NUMA node), host thread counts to match
    if args.scaling == "strong":
        return run_strong(args, rank, local_rank, world, one_gpu)
    contig_len = args.contig_len or synth.CHR20_LEN
    ref, rs, info = synth.generate_contig(contig_len=contig_len, seed=synth.SEED + rank, depth=args.depth)
    weights = synth.random_weights(18)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(rs, ref, weights, contig_len)      # forks workers: must come before anything touches the GPU
    torch.cuda.set_device(local_rank)
    dist, red_dev = None, "cuda"
    if world > 1:
        import torch.distributed as dist
        if one_gpu:
            dist.init_process_group("gloo")
            red_dev = "cpu"
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    chunks = chunk_list(contig_len)
    # the host-resident flat records of the contig, in page-locked memory (c3r_host_alloc) so that their upload is a DMA transfer
    rs_host = capi.pinned_readset(rs)
    eng = capi.Engine(local_rank)
    eng.set_params()
    eng.load_reads(rs_host)
    eng.set_reference(1, ref)
    eng.load_weights(weights, 18)
    eng.set_precision(args.precision)

    def tensor_build(e):
        # tensor build chunk by chunk (the reference's work items); the candidates of all chunks stay resident
        # the reference's work items are the 13 chunks (each with its own +-33 bp halo); they go through ONE set of kernel
        # launches (c3r_pileup_scan_regions) and their candidates stay resident for the network
        e.begin_batch()
        total = e.scan_regions(chunks)
        e.end_batch()
        return total

    host_inputs = [True]     # False: the `resident_inputs` measurement (read tables prepared once, outside the timed region)

    def one_step():
        # ... and go through the network in ONE launch per layer (batch mode); probabilities come back to the host
        if host_inputs[0]:
            eng.load_reads(rs_host)
        total = tensor_build(eng)
        if total:
            eng.infer()
        return total

    # Two contexts = two HIP streams on the same GPU: while context A runs the network of pass i, context B builds the
    # tensors of pass i+1 (the scan workgroups fit beside the persistent LSTM workgroups: 29 KB vs 128 KB of LDS).
    # Every pass still does all of its work; only the order of independent passes is pipelined.
    engs = [eng]
    for _ in range(0 if args.no_overlap else max(1, args.contexts) - 1):
        e2 = capi.Engine(local_rank)
        e2.set_params(); e2.load_reads(rs_host); e2.set_reference(1, ref); e2.load_weights(weights, 18); e2.set_precision(args.precision)
        engs.append(e2)

    def run_steps(k):
        if len(engs) == 1:
            return sum(one_step() for _ in range(k))
        ne = len(engs)
        total, pending = 0, [None] * ne
        for i in range(k):
            j = i % ne
            e = engs[j]
            if pending[j] is not None:                            # results of pass i-ne (already long finished)
                e.fetch_probs(pending[j])
            if host_inputs[0]:
                e.load_reads(rs_host)                             # host records -> device tables: part of the pass
            n = tensor_build(e)                                   # overlaps with the other contexts' network launches
            if n:
                e.infer(fetch=False)
            pending[j] = n
            total += n
        for j in range(ne):
            if pending[j]:
                engs[j].fetch_probs(pending[j])
        return total

    def barrier():
        for e in engs:
            e.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # setup, untimed: each context sizes its device buffers on first use (hipMalloc of several GB, 0.1-0.2 s on a fresh
    # box).  With W < 2 the second context would otherwise meet its first pass inside the timed region.
    for e in engs:
        if tensor_build(e):
            e.infer()
    barrier()
    run_steps(max(args.warmup, 0))
    barrier()
    t0 = time.perf_counter()
    sites = run_steps(args.steps)
    for e in engs:
        e.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        from clair3_rna_amd import shard
        elapsed = shard.reduce_max(dist, elapsed, device=red_dev)    # max over ranks
        sites = int(shard.reduce_sum(dist, sites, device=red_dev))   # whole-job aggregate
    sites_per_step_rank = sites / max(1, args.steps) / world

    # ---- the same K steps with the read tables already prepared on the device (what rounds 1-2 reported as `value`): an ADDITIONAL
    # figure that shows what handing the records over costs
    resident = None
    if args.steps > 0 and not args.no_resident:
        host_inputs[0] = False
        barrier()
        run_steps(min(2, args.steps))
        barrier()
        rsteps = max(args.steps, 100)          # seconds, not tenths, of back-to-back device work: long enough for an outside sampler to see
        t1 = time.perf_counter()
        rsites = run_steps(rsteps)
        for e in engs:
            e.synchronize()
        torch.cuda.synchronize()
        rel = time.perf_counter() - t1
        if dist is not None:
            dist.barrier()
            from clair3_rna_amd import shard
            rel = shard.reduce_max(dist, rel, device=red_dev)
            rsites = int(shard.reduce_sum(dist, rsites, device=red_dev))
        resident = dict(value=round(rsites / rel, 1), unit="sites/s", ms_per_step=round(1e3 * rel / rsteps, 3), steps=rsteps,
                        note="read tables (headers, position-binned pile table) prepared once outside the timed region; not the headline")
        host_inputs[0] = True

    # ---- the same K steps once more in precision "auto" (f16 main term + fp8 correction terms where the library's calibration through
    # the loaded weights allows it): an ADDITIONAL figure, never `value` — the headline stays on the fp32-equivalent arithmetic
    fast = None
    if args.precision == "f16x3" and not args.no_fast and args.steps > 0:
        for e in engs:
            e.set_precision("auto")
        mode, cal = eng.precision()
        if mode == "f16+f8":
            barrier()
            run_steps(2)
            barrier()
            t1 = time.perf_counter()
            fsites = run_steps(args.steps)
            for e in engs:
                e.synchronize()
            torch.cuda.synchronize()
            fel = time.perf_counter() - t1
            if dist is not None:
                dist.barrier()
                from clair3_rna_amd import shard
                fel = shard.reduce_max(dist, fel, device=red_dev)
                fsites = int(shard.reduce_sum(dist, fsites, device=red_dev))
            fast = dict(value=round(fsites / fel, 1), unit="sites/s", ms_per_step=round(1e3 * fel / args.steps, 3), precision="auto -> f16+f8",
                        dtype="f16 main term + fp8 (block-scaled) correction terms in layer 2 and L4, f32 accumulate",
                        calibration_max_dP=cal, guard=4e-5,
                        note="opt-in arithmetic, ~10x the error of f16x3 (max |dP| 2-3e-5 on these weights, tolerance 1e-4); not the headline")
        else:
            fast = dict(value=None, precision="auto -> " + mode, calibration_max_dP=cal, guard=4e-5)
        for e in engs:
            e.set_precision(args.precision)

    # ---- three steps in the reference's own arithmetic (clair3_rna/model.py:175-216 runs fp32): the exact-fp32 MFMA path, which is also where the
    # split-f16 guard sends weights it cannot hold — an ADDITIONAL figure in every line
    f32 = None
    if args.precision != "f32" and args.steps > 0 and not args.no_f32:
        try:
            for e in engs:
                e.set_precision("f32")
            barrier()
            run_steps(len(engs))                      # (every context once on this path, untimed)
            barrier()
            t1 = time.perf_counter()
            fsites = run_steps(3)
            for e in engs:
                e.synchronize()
            torch.cuda.synchronize()
            fel = time.perf_counter() - t1
            if dist is not None:
                dist.barrier()
                from clair3_rna_amd import shard
                fel = shard.reduce_max(dist, fel, device=red_dev)
                fsites = int(shard.reduce_sum(dist, fsites, device=red_dev))
            eng.set_profiling(True); eng.reset_kernel_stats()
            nf = one_step()
            eng.set_profiling(False)
            kf = eng.kernel_stats()
            net_ms = sum(v["total_ms"] for k, v in kf.items() if k in NET_KERNELS)
            fl = sum(flop_per_site(18).values()) * nf
            f32 = dict(value=round(fsites / fel, 1), unit="sites/s", ms_per_step=round(1e3 * fel / 3, 3), steps=3, precision="f32",
                       dtype="f32 (v_mfma_f32_32x32x2_f32), f32 accumulate", network_ms=round(net_ms, 3),
                       network_tflops=round(fl / (net_ms * 1e-3) / 1e12, 1) if net_ms else None,
                       network_frac_of_f32_mfma_peak=round(fl / (net_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4) if net_ms else None,
                       kernels_ms_per_step={k: round(v["total_ms"], 3) for k, v in sorted(kf.items()) if k != "h2d_reads"},
                       note="the same passes with c3r_set_precision(0): the reference's own arithmetic; not the headline")
        except Exception as ex:
            f32 = dict(error="%s: %s" % (type(ex).__name__, ex))
        finally:
            for e in engs:
                e.set_precision(args.precision)

    # ---- per-kernel durations, live, with HIP events on the engine's stream (one extra untimed step)
    roofline, kernels, stage_rates, roofline_tb = None, {}, None, None
    if not args.no_profile:
        eng.set_profiling(True)
        one_step()                     # (under the profiler every kernel runs on the context's main stream, k_fused_deep for the first time: one pass unrecorded)
        eng.reset_kernel_stats()
        n_prof = one_step()
        eng.set_profiling(False)
        kernels = eng.kernel_stats()
        dom = max(kernels, key=lambda k: kernels[k]["total_ms"])
        roofline, roofline_tb, stage_rates = rooflines(kernels, n_prof, rs, eng.sites()["pos"] if n_prof else [], 18, args.precision)
        traffic_fn = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # per-launch HBM bytes from rocprofv3 --pmc passes
        if roofline and os.path.exists(traffic_fn):
            try:
                per_kernel = json.load(open(traffic_fn)).get(args.precision, {})
                roofline["traffic"] = per_kernel.get(dom)
                roofline["traffic_source"] = "static: profiles/pmc_traffic.json, per-launch FETCH_SIZE / WRITE_SIZE of separate rocprofv3 --pmc passes over this command (tools/profile_bench.sh), not collected in this run"
                # counter bytes of one pass through every tensor-build kernel (launches per pass: one each, the three-launch scans and the
                # library sort a handful — their per-launch averages are small): raw FETCH_SIZE + WRITE_SIZE, see profiles/*_pmc_*.csv
                tb = {k: v for k, v in per_kernel.items() if k.startswith("k_") and not k.startswith(("k_lstm", "k_heads", "k_fc4"))}
                roofline_tb["traffic"] = int(sum(tb.values()))
                roofline_tb["traffic_note"] = "sum of per-launch HBM counter bytes of the tensor-build kernels (profiles/pmc_traffic.json); algorithmic bytes are bytes_per_pass"
            except Exception:
                pass

    # ---- BASELINE.json configs[2] on ONE GPU: the 24 GRCh38 contigs (a quarter of each) from host-resident reads AND reference, so that the
    # driver's record carries the configuration `--scaling strong` shards over N ranks — an ADDITIONAL figure, never `value`
    n_streams = len(engs)
    strong, phased, stress, capped, real = None, None, None, None, None
    if world == 1 and args.steps > 0 and not (args.no_strong and args.no_extra):
        for e in engs:
            e.close()
        engs = []
    if world == 1 and not args.no_strong and args.steps > 0:
        try:
            sa = argparse.Namespace(**vars(args))
            sa.genome_scale, sa.steps, sa.warmup = 0.25, 1, 0
            so = run_strong(sa, rank, local_rank, world, one_gpu, emit=False)
            strong = dict(value=so["value"], unit="sites/s", ms_per_step=so["ms_per_step"], workload=so["config"]["workload"], inputs=so["config"]["inputs"],
                          genome_bp=so["config"]["genome_bp"], sites_per_step=so["config"]["sites_per_step"], streams=so["config"]["streams"],
                          note="python bench.py --scaling strong --genome_scale 0.25 --steps 1 at N = 1; with --gpus N the same contigs are dealt to N ranks")
        except Exception as ex:                       # an additional measurement must never cost the headline
            strong = dict(error="%s: %s" % (type(ex).__name__, ex))
    # ---- BASELINE.json configs[3] and configs[4] on ONE GPU, and the depth-cap cliff: ADDITIONAL figures, never `value`
    if world == 1 and not args.no_extra and args.steps > 0:
        def guarded(fn):
            try:
                return fn()
            except Exception as ex:
                return dict(error="%s: %s" % (type(ex).__name__, ex))
        phased = guarded(lambda: extra_config(
            "phased_1gpu", "synthetic PacBio MAS-Seq chr20 ~30x with HP tags, 30 channels, phased-shape weights (BASELINE.json configs[3], one contig)",
            dict(contig_len=synth.CHR20_LEN, seed=synth.SEED + 3, depth=30.0, platform="hifi", phased=True), 30, args.precision, local_rank))
        stress = guarded(lambda: extra_config(
            "stress_500x", "synthetic ONT dRNA004 windows at ~500x: 16 Mb contig, expressed loci at mean depth 500 (BASELINE.json configs[4])",
            dict(contig_len=16000000, seed=synth.SEED + 4, depth=500.0), 18, args.precision, local_rank))
        # the shape real RNA-seq has: log-normal gene expression over four to five decades on one chr20-sized contig — a few loci in the thousands,
        # a long tail of one-to-three-read islands that emit nothing (src/create_tensor_pileup.py:512-516, :551-560), mean ~20x
        real = guarded(lambda: extra_config(
            "realistic_expr", "synthetic ONT dRNA004 chr20, log-normal gene expression (sigma 2.3, mean ~20x, capped at 12,000x): a few loci at 1,000-10,000x, "
            "a long tail of one-to-three-read islands", dict(contig_len=synth.CHR20_LEN, seed=synth.SEED + 6, depth=20.0, expr_sigma=2.3, max_level=12000.0),
            18, args.precision, local_rank))
        # one locus far beyond samtools mpileup's -d 8000: the depth-cap rule runs (csrc/c3r_lib.hip, depth_cap_mask), with the cap and without it
        deep = dict(contig_len=400000, seed=synth.SEED + 5, depth=20000.0, expressed_frac=0.01, intron_lo=100.0, intron_hi=800.0)
        capped = guarded(lambda: extra_config("depth_cap_20000x", "one 400-kb contig with loci at ~20,000x: mpileup's depth cap -d 8000 in force", deep, 18, args.precision, local_rank, steps=3))
        nocap = guarded(lambda: extra_config("depth_cap_20000x_off", "the same reads with max_depth = 0 (no cap)", deep, 18, args.precision, local_rank, steps=3, params=dict(max_depth=0)))
        if isinstance(capped, dict) and "error" not in capped:
            if isinstance(nocap, dict) and "error" not in nocap:
                capped["without_cap"] = dict(ms_per_step=nocap["ms_per_step"], sites_per_step=nocap["sites_per_step"], value=nocap["value"],
                                             kernels_ms_per_step=nocap["kernels_ms_per_step"])
                capped["cap_cost_ms_per_step"] = round(capped["ms_per_step"] - nocap["ms_per_step"], 3)
            else:                                     # (a failed control is part of the record, not a missing key)
                capped["without_cap"] = nocap if isinstance(nocap, dict) else dict(error="no result")

    if rank == 0:
        out = {
            "metric": "candidate sites/sec (tensor build + inference)",
            "value": round(sites / elapsed, 1), "unit": "sites/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / max(1, args.steps), 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"f16x3": "f16 hi/lo split x3, f32 accumulate (fp32-equivalent)", "f32": "f32",
                                         "f16+f8": "f16 main term + fp8 (MX) correction terms, f32 accumulate (max |dP| 2-3e-5, NOT fp32-equivalent)",
                                         "auto": "auto: f16+f8 where the calibration allows, else f16x3"}[args.precision],
            "data": "synthetic",
            "config": {"workload": "synthetic ONT dRNA004 chr20 ~%dx (BASELINE.json configs[1])" % int(args.depth),
                       "inputs": "host-resident flat records (c3r_read_t + BAM CIGARs + 4-bit bases, page-locked); reference and weights resident in HBM",
                       "contig_len": contig_len, "chunks": len(chunks), "channels": 18, "precision": args.precision, "reads_per_rank": info["n_reads"],
                       "exonic_bp_per_rank": info["n_exonic"], "sites_per_step_per_rank": round(sites_per_step_rank, 1),
                       "parallelism": "chunks sharded by contig, %d rank(s), no collective" % world,
                       "streams": n_streams},
            "roofline": roofline, "roofline_tensor_build": roofline_tb, "cpu_baseline": cpu, "stage_rates": stage_rates,
            "resident_inputs": resident, "fast_precision": fast, "f32_mfma": f32, "strong_1gpu": strong, "phased_1gpu": phased, "stress_500x": stress,
            "depth_cap_20000x": capped, "realistic_expr": real,
            "kernels_ms_per_step": {k: round(v["total_ms"], 3) for k, v in sorted(kernels.items())},
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
