#!/usr/bin/env python3
"""What mpileup's depth cap costs a big contig that holds ONE deep locus:  python tools/cap_zone_probe.py  (one MI355X)
A 150-Mb contig at 30x (BASELINE configs[2] shape) with the reads of a 20,000x locus spliced in; the tensor build of the whole contig
(c3r_load_reads + c3r_pileup_scan_regions) with the cap off (max_depth 0), with the rule run zone by zone (default) and with the rule run
over every read of every region (C3R_CAP_ALL=1, what rounds 1-4 did)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from clair3_rna_amd import capi, synth  # noqa: E402
from clair3_rna_amd.reads import ReadSet  # noqa: E402


def merge(a, b):
    rb = b.reads.copy()
    rb["cigar_off"] += len(a.cigar)
    rb["seq_off"] += len(a.seq)
    reads = np.concatenate([a.reads, rb])
    order = np.argsort(reads["pos"], kind="stable")
    return ReadSet(reads[order], np.concatenate([a.cigar, b.cigar]), np.concatenate([a.seq, b.seq]))


def main():
    import torch
    L = 150_000_000
    ref, rs, _ = synth.generate_contig(contig_len=L, seed=synth.SEED + 11, depth=30.0)
    dref, drs, _ = synth.generate_contig(contig_len=400000, seed=synth.SEED + 5, depth=20000.0, expressed_frac=0.01, intron_lo=100.0, intron_hi=800.0)
    off = 70_000_000
    drs.reads["pos"] += off
    ref = bytearray(ref)
    ref[off:off + len(dref)] = dref                      # the locus' reads agree with the reference under them
    ref = bytes(ref)
    # (the 30x reads over the replaced stretch no longer agree with the reference: out)
    keep = (rs.reads["pos"] > off + len(dref) + 200000) | (rs.reads["pos"] < off - 400000)
    rs = ReadSet(rs.reads[keep], rs.cigar, rs.seq)
    both = merge(rs, drs)
    print("contig %d Mb: %d reads at 30x + %d reads of one 20,000x locus" % (L // 1000000, len(rs.reads), len(drs.reads)))
    chunks = bench.chunk_list(L)
    eng = capi.Engine(0)
    eng.set_reference(1, ref)
    host = capi.pinned_readset(both)

    def run(label, reps=5, **params):
        nonlocal host
        eng.params = capi.default_params()
        eng.set_params(**params)
        ts, n = [], 0
        for r in range(reps + 1):
            eng.synchronize(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.load_reads(host)
            eng.begin_batch(); n = eng.scan_regions(chunks); eng.end_batch()
            eng.synchronize(); torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        print("%-34s %8d sites   load + scan %7.2f ms (best of %d), %d regions" % (label, n, 1e3 * min(ts[1:]), reps, len(chunks)))
        if os.environ.get("C3R_PROBE_KERNELS"):
            eng.set_profiling(True); eng.reset_kernel_stats()
            t0 = time.perf_counter()
            eng.load_reads(host)
            t1 = time.perf_counter()
            eng.begin_batch(); n = eng.scan_regions(chunks); eng.end_batch()
            eng.synchronize()
            t2 = time.perf_counter()
            eng.set_profiling(False)
            ks = eng.kernel_stats()
            print("    load_reads %.2f ms, scan %.2f ms; kernels: %s" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), ", ".join("%s %.3f (x%d)" % (k, v["total_ms"], v["launches"]) for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["total_ms"])[:8])))
    if os.environ.get("C3R_PROBE_PARTS"):
        for label, part in (("30x contig alone", rs), ("deep locus alone, at 70 Mb", drs)):
            host = capi.pinned_readset(part)
            run(label + ", cap off", reps=3, max_depth=0)
            run(label + ", cap 8000", reps=3)
        host = capi.pinned_readset(both)
    run("cap off (max_depth 0)", max_depth=0)
    run("cap 8000, zone by zone")
    os.environ["C3R_CAP_ALL"] = "1"
    run("cap 8000, every read (C3R_CAP_ALL=1)")
    del os.environ["C3R_CAP_ALL"]
    run("cap 8000, zone by zone (again)")


if __name__ == "__main__":
    main()
