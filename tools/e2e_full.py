"""22 full-length GRCh38 contigs (2.9 Gb, 2.4 M reads, ~9.15 M candidate sites) through call_sample at several fetch-thread counts;
the BAM is generated once (~90 s).   python tools/e2e_full.py [--ft 4,6,8] [--reps 2] [--scale 1.0]     (C3R_TIMING=1: per-contig timeline)"""
import argparse
import os
import sys
import tempfile
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

from clair3_rna_amd import bam, bamio, call_sample, io, synth

GR = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422, 135086622, 133275309,
      114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ft", default="4,6,8,12")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--two_streams", default="", help="e.g. 0,1: repeat every setting with C3R_TWO_STREAMS set that way")
    ap.add_argument("--env", default="", help="NAME=v1,v2: repeat every setting with that environment variable set to each value")
    ap.add_argument("--ref_bias", type=float, default=0.0, help="synth.random_weights(ref_bias=...): > 0 makes most candidates reference calls (no VCF record), as a trained model does")
    ap.add_argument("--extra", default="", help="further call_sample arguments, space-separated")
    a = ap.parse_args()
    tmp = tempfile.mkdtemp(dir="/tmp")
    contigs, reads = [], {}
    for i, L in enumerate(GR):
        ref, rs, _ = synth.generate_contig(contig_len=int(L * a.scale), seed=synth.SEED + i, depth=20.0)
        contigs.append(("chr%d" % (i + 1), ref.decode())); reads["chr%d" % (i + 1)] = rs
    fa, bm, wfn = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "in.bam"), os.path.join(tmp, "model")
    io.write_fasta(fa, contigs); bam.write_bam(bm, [(n, len(r)) for n, r in contigs], reads); bamio.index_build(bm)
    np.save(wfn + ".c3rw.npy", synth.random_weights(18, ref_bias=a.ref_bias))
    del contigs, reads
    ename, evals = ("C3R_TWO_STREAMS", a.two_streams.split(",")) if a.two_streams else (None, [None])
    if a.env:
        ename, evals = a.env.split("=")[0], a.env.split("=")[1].split(",")
    for ft, ts in [(int(x), t) for x in a.ft.split(",") for t in evals]:
        if ts is not None:
            os.environ[ename] = ts
            print("%s=%s" % (ename, ts))
        for rep in range(a.reps):
            out = os.path.join(tmp, "out_%d_%d" % (ft, rep))
            msgs = []
            t0 = time.time()
            call_sample.Run(call_sample.build_parser().parse_args(["--bam_fn", bm, "--ref_fn", fa, "--output_dir", out, "--pileup_model_path", wfn,
                                                                    "--fetch_threads", str(ft)] + a.extra.split()), log=msgs.append)
            print("fetch_threads %2d rep %d: %.2f s   %s" % (ft, rep, time.time() - t0, msgs[-1].strip()), flush=True)
            if rep == 0:
                print("   " + " ".join(m.strip() for m in msgs if "records written" in m), flush=True)
            if os.environ.get("C3R_TIMING"):
                print("\n".join(m for m in msgs if "[timeline" in m or "[device_stage" in m), flush=True)


if __name__ == "__main__":
    main()
