"""Whole-sample driver (clair3_rna_amd/call_sample.py) against the reference's own flow on the same inputs: call_var_bam once per
CHUNK_LIST row (run_clair3_rna:678-708) + sort_vcf over the per-chunk files (run_clair3_rna:710-726).  Files must be byte-identical."""
import gzip
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sample(tmp, phased=False):
    from clair3_rna_amd import bam, bamio, io, synth
    spec = [("chr1", 50000, 3), ("chr2", 30000, 5), ("chrX", 26000, 7), ("scaffold_7", 20000, 9), ("chr5", 15000, 11)]
    contigs, reads = [], {}
    for name, L, seed in spec:
        ref, rs, _ = synth.small_case(seed=seed, ref_len=L, n_genes=max(3, L // 5000), depth=18, phased=phased)
        if name == "chr2":
            ref = ref[:5000] + ref[5000:9000].lower() + ref[9000:]       # soft-masked stretch: both flows must upper-case it
        if name == "chr1":                       # reads on which the two samtools printers differ (an insertion with a deletion / a pad behind it)
            from tests import helpers as H
            rs = H.merge_readsets(rs, H.indel_next_to_indel_reads(ref, int(rs.reads["pos"][len(rs.reads) // 2])))
        contigs.append((name, ref))
        if name != "chr5":                       # a contig of the reference without any read in the BAM
            reads[name] = rs
    fa, bm, wfn = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "in.bam"), os.path.join(tmp, "model")
    io.write_fasta(fa, contigs)
    bam.write_bam(bm, [(n, len(r)) for n, r in contigs], reads)
    bamio.index_build(bm)
    np.save(wfn + ".c3rw.npy", synth.random_weights(30 if phased else 18, seed=5))
    return fa, bm, wfn, dict(contigs)


def _reference_flow(tmp, fa, bm, wfn, out_dir, extra_chunk=(), extra_merge=None, phased=False):
    """call_var_bam per CHUNK_LIST row + sort_vcf, with the files call_sample left in out_dir/tmp (CHUNK_LIST, CONTIGS, CMD,
    split_beds) — the same files run_clair3_rna writes before STEP 1."""
    from clair3_rna_amd import call_var_bam, capi, sort_vcf
    pdir = os.path.join(tmp, "pileup_output")
    os.makedirs(pdir)
    eng = capi.Engine(0)
    for row in open(os.path.join(out_dir, "tmp", "CHUNK_LIST")):
        ctg, cid, cnum = row.split()
        argv = ["--chkpnt_fn", wfn, "--bam_fn", bm, "--call_fn", os.path.join(pdir, "pileup_%s_%s.vcf" % (ctg, cid)), "--sampleName", "SAMPLE",
                "--ref_fn", fa, "--extend_bed", os.path.join(out_dir, "tmp", "split_beds", ctg), "--ctgName", ctg, "--chunk_id", cid,
                "--chunk_num", cnum, "--platform", "ont", "--snp_min_af", "0.08", "--indel_min_af", "0.15", "--minMQ", "5",
                "--minCoverage", "4", "--pileup", "--cmd_fn", os.path.join(out_dir, "tmp", "CMD")] + list(extra_chunk)
        if phased:
            argv += ["--enable_phasing_model", "True"]
        assert call_var_bam.Run(call_var_bam.build_parser().parse_args(argv), engine=eng) == 0
    eng.close()
    exp = os.path.join(tmp, "expected.vcf")
    argv = ["--input_dir", pdir, "--vcf_fn_prefix", "pileup", "--output_fn", exp, "--ref_fn", fa, "--contigs_fn",
            os.path.join(out_dir, "tmp", "CONTIGS"), "--cmd_fn", os.path.join(out_dir, "tmp", "CMD")] + list(extra_merge or [])
    # Where chunks overlap (the +-33 bp halo; with head/tail calling both neighbours emit the halo candidates, from different
    # windows) the reference keeps the row of whichever file os.listdir returns last (src/sort_vcf.py:204-236) — arbitrary.
    # call_sample keeps the later chunk's row, so the comparison pins that order.
    names = sorted(os.listdir(pdir), key=lambda n: (n.rsplit("_", 1)[0], int(n.rsplit("_", 1)[1].split(".")[0])))
    assert sort_vcf.main(argv, listing=names) == 0
    return exp


def _run_sample(out_dir, fa, bm, wfn, extra=()):
    from clair3_rna_amd import call_sample
    argv = ["--bam_fn", bm, "--ref_fn", fa, "--output_dir", out_dir, "--pileup_model_path", wfn, "--chunk_size", "12000", "--no_compress"] + list(extra)
    assert call_sample.Run(call_sample.build_parser().parse_args(argv)) == 0
    return os.path.join(out_dir, "output.vcf")


def test_sample_equals_per_chunk_flow(tmp_path):
    tmp = str(tmp_path)
    fa, bm, wfn, _ = _sample(tmp)
    got = _run_sample(os.path.join(tmp, "out"), fa, bm, wfn)
    assert open(os.path.join(tmp, "out", "tmp", "CONTIGS")).read().split("\n") == ["chr1", "chr2", "chrX"]     # scaffold_7: not a major contig; chr5: no reads
    rows = [r.split() for r in open(os.path.join(tmp, "out", "tmp", "CHUNK_LIST"))]
    assert [r for r in rows if r[0] == "chr1"] == [["chr1", str(k), "5"] for k in range(1, 6)]
    exp = _reference_flow(tmp, fa, bm, wfn, os.path.join(tmp, "out"))
    a, b = open(got).read(), open(exp).read()
    assert a == b
    recs = [r for r in a.split("\n") if r and r[0] != "#"]
    assert len(recs) > 50 and {r.split("\t")[0] for r in recs} == {"chr1", "chr2", "chrX"}
    assert all(r.split("\t")[4] != "." for r in recs)                  # RefCall rows are dropped without --print_ref_calls
    # a BAM without .bai: the driver indexes a link under tmp/ instead of scanning the whole file once per contig
    os.remove(bm + ".bai")
    again = _run_sample(os.path.join(tmp, "out_noindex"), fa, bm, wfn, ["--contexts", "3", "--fetch_threads", "2"])   # and three contexts side by side
    assert open(again).read() == a and os.path.exists(os.path.join(tmp, "out_noindex", "tmp", "input.bam.bai"))
    assert not os.path.exists(bm + ".bai")


def test_sample_follows_the_samtools_it_is_given(tmp_path):
    """--mpileup_compat: the whole-sample driver restates the printer of the samtools the user names (auto: `--samtools --version`), and
    its records equal the per-chunk flow's under either printer; the two printers give different files on this sample (chr1 holds reads
    with an insertion that has a deletion / a pad right behind it), and the >= 1.11 one is what a missing samtools resolves to."""
    from tests import helpers as H
    tmp = str(tmp_path)
    fa, bm, wfn, _ = _sample(tmp)
    old = H.fake_samtools(os.path.join(tmp, "samtools-1.10"), "1.10")
    new = H.fake_samtools(os.path.join(tmp, "samtools-1.19"), "1.19.2")
    a0 = open(_run_sample(os.path.join(tmp, "o0"), fa, bm, wfn, ["--samtools", old, "--print_ref_calls"])).read()
    a1 = open(_run_sample(os.path.join(tmp, "o1"), fa, bm, wfn, ["--samtools", new, "--print_ref_calls"])).read()
    assert a0 != a1
    assert open(_run_sample(os.path.join(tmp, "f0"), fa, bm, wfn, ["--mpileup_compat", "0", "--print_ref_calls"])).read() == a0
    assert open(_run_sample(os.path.join(tmp, "f1"), fa, bm, wfn, ["--samtools", os.path.join(tmp, "none"), "--print_ref_calls"])).read() == a1
    for k, (text, st) in enumerate(((a0, old), (a1, new))):
        os.makedirs(os.path.join(tmp, "flow%d" % k))
        exp = _reference_flow(os.path.join(tmp, "flow%d" % k), fa, bm, wfn, os.path.join(tmp, "o%d" % k), extra_chunk=["--samtools", st], extra_merge=["--show_ref", "True"])
        assert open(exp).read() == text
    # against the oracle: the candidate lines of chr1 under the >= 1.11 text, position by position (RefCall rows are printed too)
    from clair3_rna_amd import bamio
    bf = bamio.BamFile(bm)
    rs1 = bf.fetch("chr1")
    bf.close()
    ref1 = open(fa).read().split(">")[1].split("\n", 1)[1].replace("\n", "")
    pos1 = sorted({int(r.split("\t")[1]) for r in a1.split("\n") if r.startswith("chr1\t")})
    chunks = [r.split() for r in open(os.path.join(tmp, "o1", "tmp", "CHUNK_LIST")) if r.split()[0] == "chr1"]
    from clair3_rna_amd import call_var_bam
    exp_pos = set()
    for _c, cid, cnum in chunks:
        a, b = call_var_bam.chunk_region(len(ref1), int(cid), int(cnum))
        exp_pos.update(int(l.split("\t")[1]) for l in H.oracle_chunk(rs1, ref1, 1, a, b, mpileup_compat=1)["lines"])
    assert pos1 == sorted(exp_pos) and len(pos1) > 30


def test_sample_options_bed_refcalls_all_contigs_tagging(tmp_path):
    """--bed_fn (chunks follow the BED extent, split_beds), --print_ref_calls, --include_all_ctgs is implied by the BED contigs,
    head/tail + splice padding, --qual and REDIportal tagging incl. the _no_tagging twin."""
    tmp = str(tmp_path)
    fa, bm, wfn, refs = _sample(tmp)
    bed = os.path.join(tmp, "conf.bed")
    with open(bed, "w") as f:
        f.write("#track\nchr1\t2000\t31000\nchr1\t33000\t47000\nscaffold_7\t100\t19000\nchrX\t5000\t5000\nchrX\t9000\t25000\n")
    first = _run_sample(os.path.join(tmp, "probe"), fa, bm, wfn, ["--bed_fn", bed])
    var = [r.split("\t") for r in open(first) if r[0] != "#"]
    snps = [v for v in var if len(v[3]) == 1 and len(v[4]) == 1][:12]
    assert len(snps) >= 6
    redi = os.path.join(tmp, "redi.tsv.gz")
    with gzip.open(redi, "wt") as f:
        f.write("Region\tPosition\tRef\tEd\tStrand\tdb\n")
        for v in snps:
            f.write("%s\t%s\t%s\t%s\t+\tA,R\n" % (v[0], v[1], v[3], v[4]))
    opts = ["--bed_fn", bed, "--print_ref_calls", "--qual", "8", "--enable_variant_calling_at_sequence_head_and_tail",
            "--enable_padding_in_splice_junction_regions", "--tag_variant_using_readiportal", "--readiportal_source_fn", redi]
    got = _run_sample(os.path.join(tmp, "out"), fa, bm, wfn, opts)
    assert open(os.path.join(tmp, "out", "tmp", "CONTIGS")).read().split("\n") == ["chr1", "chrX", "scaffold_7"]
    exp = _reference_flow(tmp, fa, bm, wfn, os.path.join(tmp, "out"),
                          extra_chunk=["--bed_fn", bed, "--enable_variant_calling_at_sequence_head_and_tail", "True",
                                       "--enable_padding_in_splice_junction_regions", "True"],
                          extra_merge=["--show_ref", "True", "--qual", "8", "--tag_variant_using_readiportal", "True",
                                       "--readiportal_source_fn", redi, "--output_no_tagging_fn", os.path.join(tmp, "expected_nt.vcf")])
    a = open(got).read()
    assert a == open(exp).read()
    assert open(os.path.join(tmp, "out", "output_no_tagging.vcf")).read() == open(os.path.join(tmp, "expected_nt.vcf")).read()
    assert a.count("\tRNAEditing\t") >= 4 and "\tRefCall\t" in a and "\tLowQual\t" in a


def test_sample_phased_genotyping_and_compressed_output(tmp_path):
    """30-channel pass on a haplotagged BAM, --genotyping_mode_vcf_fn (one scan per chunk with its own site list), --ctg_name,
    and the default bgzip + tabix output."""
    from clair3_rna_amd import call_sample
    tmp = str(tmp_path)
    fa, bm, wfn, _ = _sample(tmp, phased=True)
    out = os.path.join(tmp, "probe")
    argv = ["--bam_fn", bm, "--ref_fn", fa, "--output_dir", out, "--pileup_model_path", "unused", "--phased_pileup_model_path", wfn,
            "--enable_phasing_model", "--chunk_size", "12000", "--no_compress", "--ctg_name", "chr2,chrX"]
    assert call_sample.Run(call_sample.build_parser().parse_args(argv)) == 0
    got = os.path.join(out, "output_enable_phasing.vcf")
    exp = _reference_flow(tmp, fa, bm, wfn, out, phased=True)
    a = open(got).read()
    assert a == open(exp).read() and a.count("\n") > 60
    # genotyping mode on the sites just called (+ a few uncovered positions)
    sites = [r.split("\t")[:2] for r in a.split("\n") if r and r[0] != "#"][::2]
    known = os.path.join(tmp, "known.vcf")
    with open(known, "w") as f:
        f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\n")
        for c, p in sites:
            f.write("%s\t%s\t.\tA\tC\n" % (c, p))
        f.write("chr2\t29990\t.\tA\tC\n")
    out2 = os.path.join(tmp, "geno")
    argv2 = ["--bam_fn", bm, "--ref_fn", fa, "--output_dir", out2, "--pileup_model_path", "unused", "--phased_pileup_model_path", wfn,
             "--enable_phasing_model", "--chunk_size", "12000", "--genotyping_mode_vcf_fn", known, "--print_ref_calls"]
    assert call_sample.Run(call_sample.build_parser().parse_args(argv2)) == 0
    os.rename(os.path.join(tmp, "pileup_output"), os.path.join(tmp, "pileup_output_1"))
    exp2 = _reference_flow(tmp, fa, bm, wfn, out2, extra_chunk=["--vcf_fn", known], extra_merge=["--show_ref", "True"], phased=True)
    gz = os.path.join(out2, "output_enable_phasing.vcf.gz")
    assert os.path.exists(gz + ".tbi") and not os.path.exists(gz[:-3])
    assert gzip.open(gz, "rt").read() == open(exp2).read()


def test_two_ranks_share_the_contigs_and_rank0_assembles_the_same_file(tmp_path):
    """One process per GPU (here: two ranks on the one GPU of the box): contigs dealt largest-first, per-contig parts under
    tmp/parts, rank 0 concatenates — the result must be the single-process file, incl. tagging and the _no_tagging twin."""
    import socket
    import subprocess
    import sys
    tmp = str(tmp_path)
    fa, bm, wfn, _ = _sample(tmp)
    one = _run_sample(os.path.join(tmp, "one"), fa, bm, wfn, ["--include_all_ctgs", "--print_ref_calls"])
    var = [r.split("\t") for r in open(one) if r[0] != "#" and r.split("\t")[4] != "."]
    redi = os.path.join(tmp, "redi.tsv")
    with open(redi, "w") as f:
        f.write("Region\tPosition\tRef\tEd\tStrand\tdb\n")
        for v in var[::40]:
            f.write("%s\t%s\t%s\t%s\t+\tA\n" % (v[0], v[1], v[3], v[4]))
    opts = ["--include_all_ctgs", "--print_ref_calls", "--tag_variant_using_readiportal", "--readiportal_source_fn", redi]
    one = _run_sample(os.path.join(tmp, "one_t"), fa, bm, wfn, opts)
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    out2 = os.path.join(tmp, "two")
    argv = [sys.executable, "-m", "clair3_rna_amd.call_sample", "--bam_fn", bm, "--ref_fn", fa, "--output_dir", out2, "--pileup_model_path", wfn,
            "--chunk_size", "12000", "--no_compress", "--gpu_id", "0"] + opts
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen(argv, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    def body(fn):                                                   # (the ##cmdline header line names the launching command)
        return [r for r in open(fn).read().split("\n") if not r.startswith("##cmdline=")]
    assert body(os.path.join(out2, "output.vcf")) == body(one)
    assert body(os.path.join(out2, "output_no_tagging.vcf")) == body(os.path.join(tmp, "one_t", "output_no_tagging.vcf"))
    assert open(os.path.join(out2, "tmp", "CONTIGS")).read() == open(os.path.join(tmp, "one_t", "tmp", "CONTIGS")).read()
    parts = [n for n in os.listdir(os.path.join(out2, "tmp", "parts")) if n.endswith(".json")]
    assert sorted(parts) == ["rank0.json", "rank1.json"]
    import json
    c0, c1 = (json.load(open(os.path.join(out2, "tmp", "parts", n)))["called"] for n in sorted(parts))
    assert c0 and c1 and not set(c0) & set(c1)                      # both ranks worked, on different contigs


def test_one_rank_on_its_eighth_of_the_host(tmp_path):
    """What rank 0 of an 8-GPU run gets of the host, tried on one GPU: C3R_HOST_SLICE=0/8 pins the process to the slice shard.host_budget
    computes, cuts the decode / fetch / inflate / compression threads to match, and the sample comes out byte-identical to the unconfined
    run.  (Rates of the confined against the unconfined driver: tools/host_slice.py -> profiles/r4/host_slice.txt.)"""
    import subprocess
    import sys
    tmp = str(tmp_path)
    fa, bm, wfn, _ = _sample(tmp)
    got = _run_sample(os.path.join(tmp, "out"), fa, bm, wfn)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys; sys.path.insert(0, %r)\n"
            "from clair3_rna_amd import call_sample\n"
            "n0 = len(os.sched_getaffinity(0))\n"
            "argv = ['--bam_fn', %r, '--ref_fn', %r, '--output_dir', %r, '--pileup_model_path', %r, '--chunk_size', '12000', '--no_compress']\n"
            "for k in range(2):\n"
            "    assert call_sample.Run(call_sample.build_parser().parse_args(argv)) == 0\n"
            "    print('SLICE', n0, len(os.sched_getaffinity(0)), os.environ.get('C3R_THREADS'), os.environ.get('C3R_FETCH_INFLATE'))\n"
            % (root, bm, fa, os.path.join(tmp, "out8"), wfn))
    env = dict(os.environ, C3R_HOST_SLICE="0/8")
    for k in ("C3R_THREADS", "OMP_NUM_THREADS", "C3R_FETCH_INFLATE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l.split() for l in p.stdout.split("\n") if l.startswith("SLICE")]
    assert len(lines) == 2
    for _tag, n0, n1, thr, infl in lines:                     # the second run of the process keeps the same slice (it used to shrink: 32 -> 4 -> 1)
        n0, n1 = int(n0), int(n1)
        assert n1 == max(1, n0 // 8) and 1 <= int(thr) <= n1 and int(infl) >= 1
    strip = lambda t: [l for l in t.split("\n") if not l.startswith("##cmdline=")]       # (the child's argv differs)
    assert strip(open(os.path.join(tmp, "out8", "output.vcf")).read()) == strip(open(got).read())


@pytest.mark.parametrize("ref_bias", [0.0, 4.0])
def test_snapshot_without_ref_calls_equals_the_full_snapshot(ref_bias):
    """c3r_rows_begin_ex: with drop_ref_calls only the sites that survive the decoder's early RefCall exit (clair3_rna/call_variants.py:540-542,
    applied on the device) are copied out, and the decoder reads inserted bases from the caller's own arrays.  Decoded without show_ref the
    rows must be those of the full snapshot, byte for byte — with weights that call almost everything a variant and with a zygosity head
    biased to 0/0 the way a trained model is (most sites dropped on the device)."""
    from clair3_rna_amd import capi, synth
    ref, rs, _ = synth.small_case(seed=77, ref_len=60000, n_genes=12, depth=25)
    e = capi.Engine(0)
    try:
        e.set_params(min_coverage=2)
        e.load_reads(rs); e.set_reference(1, ref)
        e.load_weights(synth.random_weights(18, seed=9, ref_bias=ref_bias), 18)
        n = e.scan(1, len(ref))
        assert n > 500
        e.infer(fetch=False)
        full_all, n_all = e.rows_begin(host_reads=False).decode("chr20", qual=2, show_ref=True)
        full, n_full = e.rows_begin(host_reads=False).decode("chr20", qual=2, show_ref=False)
        inplace, n_inplace = e.rows_begin(host_reads=True).decode("chr20", qual=2, show_ref=False)
        snap = e.rows_begin(drop_ref_calls=True)
        with pytest.raises(ValueError):
            snap.decode("chr20", qual=2, show_ref=True)
        snap = e.rows_begin(drop_ref_calls=True)
        filt, n_filt = snap.decode("chr20", qual=2, show_ref=False)
        assert filt == full == inplace and n_filt == n_full == n_inplace
        assert n_all == n and n_full < n_all
        if ref_bias:
            assert n_full < 0.5 * n_all, (n_full, n_all)        # (most candidates are reference calls, as with a trained model)
        else:
            assert n_full > 100
        assert e.call_rows_text("chr20", qual=2, show_ref=False)[0] == full
    finally:
        e.close()
