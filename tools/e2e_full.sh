# full-length 22-contig sample through call_sample at several fetch-thread counts (the BAM is generated once):  bash tools/e2e_full.sh
cd /tmp && export TMPDIR=/tmp
python - <<'PY'
import os, sys, time, tempfile
R = os.environ["GRAFT_REPO_ROOT"]; sys.path.insert(0, R)
import numpy as np
from clair3_rna_amd import bam, bamio, call_sample, io, synth
sys.argv = ["x"]
GR = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422, 135086622, 133275309,
      114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468]
tmp = tempfile.mkdtemp(dir="/tmp")
contigs, reads = [], {}
for i, L in enumerate(GR):
    ref, rs, _ = synth.generate_contig(contig_len=L, seed=synth.SEED + i, depth=20.0)
    contigs.append(("chr%d" % (i + 1), ref.decode())); reads["chr%d" % (i + 1)] = rs
fa, bm, wfn = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "in.bam"), os.path.join(tmp, "model")
io.write_fasta(fa, contigs); bam.write_bam(bm, [(n, len(r)) for n, r in contigs], reads); bamio.index_build(bm)
np.save(wfn + ".c3rw.npy", synth.random_weights(18))
del contigs, reads
for ft in (4, 6, 8, 12):
    for rep in range(2):
        out = os.path.join(tmp, "out_%d_%d" % (ft, rep))
        msgs = []
        t0 = time.time()
        call_sample.Run(call_sample.build_parser().parse_args(["--bam_fn", bm, "--ref_fn", fa, "--output_dir", out, "--pileup_model_path", wfn, "--fetch_threads", str(ft)]), log=msgs.append)
        print("fetch_threads %2d rep %d: %.2f s   %s" % (ft, rep, time.time() - t0, msgs[-1].strip()), flush=True)
PY
