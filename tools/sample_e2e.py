# End-to-end timing of the whole-sample driver on the GPU box: synthetic multi-contig BAM + FASTA (generator of SURVEY.md §8d,
# contig lengths scaled by --scale so that the BAM is written in seconds) -> clair3_rna_amd.call_sample -> merged VCF.
# Reports BAM->VCF wall time and candidate sites/s INCLUDING the host stages (fetch, decode, merge, bgzip+tabix).
#   python tools/sample_e2e.py [--contigs 6] [--scale 0.25] [--depth 20]
import argparse, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from clair3_rna_amd import bam, bamio, call_sample, io, synth

GRCH38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422, 135086622, 133275309,
          114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468]
ap = argparse.ArgumentParser()
ap.add_argument("--contigs", type=int, default=6); ap.add_argument("--scale", type=float, default=0.25)
ap.add_argument("--depth", type=float, default=20.0); ap.add_argument("--repeat", type=int, default=2)
ap.add_argument("--contexts", type=int, default=None); ap.add_argument("--fetch_threads", type=int, default=None)
ap.add_argument("--chr20", action="store_true", help="one contig: the full synthetic chr20 of BASELINE.json configs[1]")
ap.add_argument("--check", action="store_true", help="also run call_var_bam per CHUNK_LIST row + sort_vcf and compare the files byte for byte")
a = ap.parse_args()
tmp = tempfile.mkdtemp(dir=os.environ.get("TMPDIR", "/tmp"))
t0 = time.time()
contigs, reads = [], {}
plan = [("chr20", synth.CHR20_LEN, synth.SEED)] if a.chr20 else [("chr%d" % (i + 1), int(GRCH38[i] * a.scale), synth.SEED + i) for i in range(a.contigs)]
for name, L, seed in plan:
    ref, rs, _ = synth.generate_contig(contig_len=L, seed=seed, depth=a.depth)
    contigs.append((name, ref.decode()))
    reads[name] = rs
fa, bm, wfn = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "in.bam"), os.path.join(tmp, "model")
io.write_fasta(fa, contigs)
bam.write_bam(bm, [(n, len(r)) for n, r in contigs], reads)
bamio.index_build(bm)
np.save(wfn + ".c3rw.npy", synth.random_weights(18))
n_reads = sum(len(r.reads) for r in reads.values())
a.contigs = len(plan)
print("inputs: %d contigs, %.0f Mb, %d reads, BAM %.1f MB (generated in %.0f s)" % (a.contigs, sum(len(r) for _, r in contigs) / 1e6, n_reads,
      os.path.getsize(bm) / 1e6, time.time() - t0), flush=True)
del contigs, reads
for rep in range(a.repeat):
    out = os.path.join(tmp, "out%d" % rep)
    argv = ["--bam_fn", bm, "--ref_fn", fa, "--output_dir", out, "--pileup_model_path", wfn]
    if a.contexts: argv += ["--contexts", str(a.contexts)]
    if a.fetch_threads: argv += ["--fetch_threads", str(a.fetch_threads)]
    msgs = []
    t1 = time.time()
    call_sample.Run(call_sample.build_parser().parse_args(argv), log=msgs.append)
    dt = time.time() - t1
    n_sites = int([m for m in msgs if "candidate sites" in m][0].split(" contigs, ")[1].split(" candidate")[0])
    print("run %d: BAM -> %s in %.2f s : %.0f candidate sites, %.2f M sites/s host-inclusive" % (rep, os.path.basename(out) + "/output.vcf.gz", dt, n_sites, n_sites / dt / 1e6))
    for m in msgs:
        if "device_stage" in m or "timeline" in m or m is msgs[-1]: print("   ", m)

if a.check:
    from clair3_rna_amd import call_var_bam, capi, sort_vcf
    out = os.path.join(tmp, "chk"); argv = ["--bam_fn", bm, "--ref_fn", fa, "--output_dir", out, "--pileup_model_path", wfn, "--no_compress"]
    call_sample.Run(call_sample.build_parser().parse_args(argv), log=lambda m: None)
    pdir = os.path.join(tmp, "pileup_output"); os.makedirs(pdir)
    eng = capi.Engine(0)
    t1 = time.time()
    rows = [r.split() for r in open(os.path.join(out, "tmp", "CHUNK_LIST"))]
    for ctg, cid, cnum in rows:
        av = ["--chkpnt_fn", wfn, "--bam_fn", bm, "--call_fn", os.path.join(pdir, "pileup_%s_%s.vcf" % (ctg, cid)), "--ref_fn", fa, "--ctgName", ctg,
              "--chunk_id", cid, "--chunk_num", cnum, "--snp_min_af", "0.08", "--indel_min_af", "0.15", "--minMQ", "5", "--minCoverage", "4", "--pileup",
              "--cmd_fn", os.path.join(out, "tmp", "CMD")]
        sys.stderr = open(os.devnull, "w")
        try:
            assert call_var_bam.Run(call_var_bam.build_parser().parse_args(av), engine=eng) == 0
        finally:
            sys.stderr = sys.__stderr__
    names = sorted(os.listdir(pdir), key=lambda n: (n.rsplit("_", 1)[0], int(n.rsplit("_", 1)[1].split(".")[0])))
    exp = os.path.join(tmp, "expected.vcf")
    sort_vcf.main(["--input_dir", pdir, "--vcf_fn_prefix", "pileup", "--output_fn", exp, "--ref_fn", fa, "--contigs_fn", os.path.join(out, "tmp", "CONTIGS"),
                   "--cmd_fn", os.path.join(out, "tmp", "CMD")], listing=names)
    x, y = open(os.path.join(out, "output.vcf")).read(), open(exp).read()
    assert x == y, "call_sample output differs from the per-chunk flow"
    print("CHECK OK: %d chunks through call_var_bam + sort_vcf in %.1f s -> %d bytes, %d records, identical to call_sample's output"
          % (len(rows), time.time() - t1, len(y), sum(1 for r in y.split("\n") if r and r[0] != "#")))
