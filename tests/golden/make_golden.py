#!/usr/bin/env python3
"""Generate golden vectors G1..G4 by calling the reference's own Python functions.

Run ONLY in the build container (needs /root/reference):

    python tests/golden/make_golden.py [g1|g2|g3|g4|all]

Outputs (committed, small):
    tests/golden/g1_columns.json     generate_tensor           (src/create_tensor_pileup.py:85-302)
    tests/golden/g2_streams.json     CreateTensorPileup driver (src/create_tensor_pileup.py:333-657)
    tests/golden/g3_batches.json     tensor_generator_from     (clair3_rna/utils.py:64-138)
    tests/golden/g4_decode.json      batch_output -> VCF rows  (clair3_rna/call_variants.py:1077-1392)

G2b/G5 (fixtures that need the oracle / torch) are produced by make_golden_e2e.py.
The inputs are synthetic and seeded; the outputs are whatever the reference computes.
"""
import gzip
import json
import os
import random
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refharness as rh  # noqa: E402

SEED = 20240422
BASES = "ACGT"


# --------------------------------------------------------------------------- column text generator
def rand_seq(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n))


def rand_column(rng, depth, ref_base, phased=False, p_alt=0.08, p_ins=0.04, p_del=0.04, p_star=0.03,
                p_skip=0.05, p_n=0.01, p_iupac=0.0, p_head=0.05, p_tail=0.05, fwd_frac=0.5, alt_base=None,
                max_indel=12):
    """One samtools-mpileup style BASES string (no -f, --reverse-del) with `depth` reads."""
    out = []
    hp = []
    mapq_chars = "+-<>$*#^AaI]~!5"
    for _ in range(depth):
        fwd = rng.random() < fwd_frac
        s = ""
        if rng.random() < p_head:
            s += "^" + rng.choice(mapq_chars)
        r = rng.random()
        if r < p_skip:
            s += ">" if fwd else "<"
        elif r < p_skip + p_star:
            s += "*" if fwd else "#"
        elif r < p_skip + p_star + p_n:
            s += "N" if fwd else "n"
        elif r < p_skip + p_star + p_n + p_iupac:
            c = rng.choice("RYSWKMBDHV")
            s += c if fwd else c.lower()
        else:
            if rng.random() < p_alt:
                b = alt_base if alt_base else rng.choice([x for x in BASES if x != ref_base] or BASES)
            else:
                b = ref_base if ref_base in BASES else rng.choice(BASES)
            s += b if fwd else b.lower()
        r2 = rng.random()
        if r2 < p_ins:
            n = 1 if rng.random() < 0.6 else rng.randint(2, max_indel)
            seq = rand_seq(rng, n, "ACGT" if rng.random() < 0.9 else "ACGTN")
            s += "+%d%s" % (n, seq if fwd else seq.lower())
        elif r2 < p_ins + p_del:
            n = 1 if rng.random() < 0.6 else rng.randint(2, max_indel)
            s += "-%d%s" % (n, ("N" if fwd else "n") * n)
        if rng.random() < p_tail:
            s += "$"
        out.append(s)
        hp.append(rng.choice(["1", "2", "*", "1", "2"]) if phased else None)
    return "".join(out), (hp if phased else None)


def handmade_columns():
    """(bases, ref_base, hp) triples exercising each token type."""
    cases = []
    for c in "+-<>$*#^A":
        cases.append(("^" + c + "T", "T", None))
    cases += [
        ("AATT", "T", None), ("TTAA", "T", None),
        ("AA" + "T" * 23, "T", None), ("A" + "T" * 12, "T", None),
        ("T+1A" * 3 + "T" * 17, "T", None), ("T+1A" * 2 + "T" * 12, "T", None),
        ("T-12NNNNNNNNNNNNt-2nn*#", "T", None),
        ("T+2NAt+2na", "T", None),
        ("NnRrTT", "T", None),
        (">><<^]>", "T", None),
        ("gGaAcCT+1ATt", "T", None),
        ("", "A", None),
        ("*" * 95 + "AAA" + "TT", "T", None),          # top allele != ref with AF < snp_min_af
        ("TTTAAA", "T", None), ("AAATTT", "T", None),  # ties, first-seen decides
        ("T+1AT+1AAAA", "T", None),                    # I ties with X
        ("A+10ACGTACGTACa+10acgtacgtacA-3NNNa-3nnn", "A", None),
        ("ACGTacgt" * 3, "N", None), ("ACGTacgt" * 3, "R", None),
        ("C$c$^~C^!c", "C", None),
        ("G-1NG-1NG-2NNg-1ng-2nng-2nn", "G", None),
        ("T+1AT+1CT+1At+1at+1a", "T", None),
        ("T+3ACGT+3ACGT+3ACTt+3acg", "T", None),
        ("*+2AG#+2ag>+1A", "C", None),                # indel after del / refskip
    ]
    # phased
    cases += [
        ("TtAa*>", "T", ["1", "2", "1", "2", "1", "2"]),
        ("T+1Aa", "T", ["1", "2"]),
        ("A-2NNa-2nnCcGgTt", "A", ["1", "2", "1", "2", "*", "1", "2", "2"]),
        ("T+1AT+1At+1aTTtt", "T", ["1", "1", "2", "2", "1", "*", "2"]),
        ("<>*#Nn", "G", ["1", "2", "1", "2", "1", "2"]),
    ]
    return cases


def gen_g1():
    ctp = rh.load_create_tensor()
    rng = random.Random(SEED)
    ref_start = 101
    ref_seq = rand_seq(rng, 400)
    cases = []

    def run(bases, pos, hp, snp_af, indel_af, ref_base_override=None):
        ref_base = ref_base_override if ref_base_override else ref_seq[pos - ref_start]
        t, alt, af, depth, pass_af, plist, mdl, msk = ctp.generate_tensor(
            pos=pos, pileup_bases=bases, reference_sequence=ref_seq, reference_start=ref_start,
            reference_base=ref_base, minimum_af_for_candidate=0.08, minimum_snp_af_for_candidate=snp_af,
            minimum_indel_af_for_candidate=indel_af, platform="ont", fast_mode=False, call_snp_only=False,
            phasing_info=hp)
        cases.append(dict(bases=bases, pos=pos, ref_base=ref_base, hp=hp, snp_af=snp_af, indel_af=indel_af,
                          out=dict(tensor=list(t), alt=[[k, v] for k, v in alt.items()], af=af, depth=depth,
                                   pass_af=bool(pass_af), pileup_list=[[k, v] for k, v in plist],
                                   max_del_length=mdl, max_skip_count=msk)))

    for bases, rb, hp in handmade_columns():
        run(bases, 150, hp, 0.08, 0.15, ref_base_override=rb)
    # AF thresholds exactly at the boundary
    for n_alt, n_tot in [(2, 25), (1, 13), (2, 26), (4, 50), (8, 100), (7, 100), (3, 37), (3, 38)]:
        run("A" * n_alt + "T" * (n_tot - n_alt), 150, None, 0.08, 0.15, ref_base_override="T")
    for n_ind, n_tot in [(3, 20), (2, 14), (15, 100), (14, 100), (3, 21)]:
        run("T+1A" * n_ind + "T" * (n_tot - n_ind), 150, None, 0.08, 0.15, ref_base_override="T")
        run("T-1N" * n_ind + "t" * (n_tot - n_ind), 150, None, 0.08, 0.15, ref_base_override="T")
    # random columns, several regimes
    for i in range(260):
        depth = rng.choice([0, 1, 2, 3, 4, 5, 8, 12, 20, 20, 30, 60, 144, 250, 500])
        pos = rng.randint(ref_start + 20, ref_start + 330)
        phased = (i % 3 == 0)
        regime = i % 5
        kw = {}
        if regime == 1:
            kw = dict(p_alt=0.5, p_ins=0.2, p_del=0.2)
        elif regime == 2:
            kw = dict(p_alt=0.02, p_ins=0.01, p_del=0.01, p_skip=0.3)
        elif regime == 3:
            kw = dict(p_star=0.5, p_alt=0.3)
        elif regime == 4:
            kw = dict(p_iupac=0.0 if phased else 0.05, fwd_frac=rng.random())
        bases, hp = rand_column(rng, depth, ref_seq[pos - ref_start], phased=phased, **kw)
        snp_af, indel_af = rng.choice([(0.08, 0.15), (0.08, 0.15), (0.0, 0.15), (0.2, 0.3)])
        run(bases, pos, hp, snp_af, indel_af)
    out = dict(ref_seq=ref_seq, ref_start=ref_start, cases=cases,
               note="generate_tensor(platform='ont', fast_mode=False, call_snp_only=False, min_af=0.08)")
    with open(os.path.join(HERE, "g1_columns.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("g1: %d cases" % len(cases))


# --------------------------------------------------------------------------- stream cases (G2)
def make_rows(rng, ctg, contig_seq, pos_ranges, depth, phased=False, snp_sites=(), **colkw):
    rows = []
    for (a, b) in pos_ranges:
        for pos in range(a, b + 1):
            ref_base = contig_seq[pos - 1]
            kw = dict(colkw)
            if pos in snp_sites:
                kw.update(p_alt=0.5)
            bases, hp = rand_column(rng, depth(pos) if callable(depth) else depth, ref_base, phased=phased, **kw)
            n = sum(1 for c in bases if c in "ACGTNacgtn*#<>")
            quals = "I" * n
            cols = [ctg, str(pos), "N", str(n), bases if bases else "*", quals if quals else "*"]
            if phased:
                cols.append(",".join(hp) if hp else "*")
            rows.append("\t".join(cols))
    return rows


def gen_g2():
    rng = random.Random(SEED + 2)
    ctg = "chr20"
    contig_seq = rand_seq(rng, 3000)
    contig_seq = contig_seq[:1200] + "N" + contig_seq[1201:1300] + "R" + contig_seq[1301:]
    cases = []

    def run(name, rows, argv, **extra):
        lines, cmds = rh.run_create_tensor(rows, contig_seq, ctg, argv, **extra)
        mp = [c for c in cmds if len(c) > 1 and c[1] == "mpileup"]
        cases.append(dict(name=name, rows=rows, argv=argv, lines=lines, mpileup_cmd=mp[0] if mp else None,
                          fai_len=extra.get("fai_len")))
        print("  g2 %-28s rows=%5d lines=%4d" % (name, len(rows), len(lines)))

    snps = set(rng.sample(range(1, 3000), 260))
    base_args = ["--ctgStart", "1", "--ctgEnd", "3000", "--minCoverage", "4"]
    ht = ["--enable_variant_calling_at_sequence_head_and_tail", "True"]
    sp = ["--enable_padding_in_splice_junction_regions", "True"]

    r = make_rows(rng, ctg, contig_seq, [(100, 700)], 20, snp_sites=snps, p_alt=0.03)
    run("contiguous", r, base_args)
    r = make_rows(rng, ctg, contig_seq, [(100, 200), (205, 300), (340, 420), (421, 470), (520, 530), (600, 700)],
                  12, snp_sites=snps, p_alt=0.05)
    run("gaps", r, base_args)
    run("gaps_headtail", r, base_args + ht)
    r = make_rows(rng, ctg, contig_seq, [(1, 79)], 10, snp_sites=snps | {5, 17, 18, 40}, p_alt=0.03)
    run("contig_start", r, ["--ctgStart", "1", "--ctgEnd", "100", "--minCoverage", "4"])
    run("contig_start_headtail", r, ["--ctgStart", "1", "--ctgEnd", "100", "--minCoverage", "4"] + ht)
    r = make_rows(rng, ctg, contig_seq, [(2940, 3000)], 10, snp_sites=snps | {2990, 2995, 3000}, p_alt=0.03)
    run("contig_end", r, ["--ctgStart", "2900", "--ctgEnd", "3000", "--minCoverage", "4"])
    run("contig_end_headtail", r, ["--ctgStart", "2900", "--ctgEnd", "3000", "--minCoverage", "4"] + ht)
    # non-ACGT reference bases inside the windows (pos 1201 'N', 1301 'R')
    r = make_rows(rng, ctg, contig_seq, [(1150, 1350)], 16, snp_sites=snps | {1201, 1301, 1202, 1290}, p_alt=0.04)
    run("ref_iupac", r, base_args)
    # introns: ref-skip heavy columns; with and without splice padding
    def d_intron(pos):
        return 25 if (pos < 1600 or pos > 1660) else 6
    r = make_rows(rng, ctg, contig_seq, [(1500, 1800)], d_intron, snp_sites=snps, p_alt=0.04, p_skip=0.0)
    r2 = []
    rr = random.Random(7)
    for row in r:
        cols = row.split("\t")
        pos = int(cols[1])
        if 1600 <= pos <= 1660:
            cols[4] = cols[4] + "".join(rr.choice("<>") for _ in range(19))
        r2.append("\t".join(cols))
    run("intron", r2, base_args)
    run("intron_splicepad", r2, base_args + sp)
    run("intron_splicepad_headtail", r2, base_args + sp + ht)
    # high / low depth, minCoverage gate
    r = make_rows(rng, ctg, contig_seq, [(2000, 2100)], lambda p: 3 if p % 7 else 5, snp_sites=snps, p_alt=0.3)
    run("low_depth", r, base_args)
    run("low_depth_mincov2", r, ["--ctgStart", "1", "--ctgEnd", "3000", "--minCoverage", "2"])
    r = make_rows(rng, ctg, contig_seq, [(2200, 2290)], 260, snp_sites=snps, p_alt=0.02)
    run("depth260", r, base_args)
    # AF == 0 forces every covered site to be a candidate
    r = make_rows(rng, ctg, contig_seq, [(2400, 2470)], 8, p_alt=0.0, p_ins=0, p_del=0)
    run("af_zero", r, base_args + ["--snp_min_af", "0.0"])
    # phased (30 channels)
    r = make_rows(rng, ctg, contig_seq, [(800, 950)], 18, phased=True, snp_sites=snps, p_alt=0.05)
    run("phased", r, base_args + ["--add_phasing_feature", "True"])
    run("phased_headtail", r, base_args + ["--add_phasing_feature", "True"] + ht)
    # chunking from the .fai (A4): contig length 3000, 3 chunks
    r = make_rows(rng, ctg, contig_seq, [(900, 1100)], 10, snp_sites=snps, p_alt=0.05)
    run("chunk2of3", r, ["--chunk_id", "2", "--chunk_num", "3", "--minCoverage", "4"])
    run("chunk1of7", make_rows(rng, ctg, contig_seq, [(380, 470)], 10, snp_sites=snps, p_alt=0.05),
        ["--chunk_id", "1", "--chunk_num", "7", "--minCoverage", "4"])
    # BED: confident bed + extend bed
    wd = "/tmp/c3r_golden"
    os.makedirs(wd, exist_ok=True)
    bed = os.path.join(wd, "conf.bed")
    with open(bed, "w") as f:
        f.write("chr20\t950\t1000\nchr20\t1020\t1021\nchr20\t1040\t1100\nchrX\t1\t5\n")
    ebed = os.path.join(wd, "ext.bed")
    with open(ebed, "w") as f:
        f.write("chr20\t934\t1116\n")
    r = make_rows(rng, ctg, contig_seq, [(920, 1130)], 14, snp_sites=snps, p_alt=0.06, p_del=0.08)
    run("bed", r, ["--bed_fn", bed, "--extend_bed", ebed, "--chunk_id", "1", "--chunk_num", "1", "--minCoverage", "4"])
    cases[-1]["bed"] = [["chr20", 950, 1000], ["chr20", 1020, 1021], ["chr20", 1040, 1100], ["chrX", 1, 5]]
    cases[-1]["extend_bed"] = [["chr20", 934, 1116]]
    # genotyping VCF mode
    vcf = os.path.join(wd, "sites.vcf")
    sites = [1010, 1015, 1050, 1051, 1100, 940, 2000]
    with open(vcf, "w") as f:
        f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\n")
        for s in sites:
            f.write("chr20\t%d\t.\tA\tG\n" % s)
        f.write("chr1\t1000\t.\tA\tG\n")
    r = make_rows(rng, ctg, contig_seq, [(920, 1130)], 9, snp_sites=snps, p_alt=0.02)
    run("genotyping_vcf", r, ["--vcf_fn", vcf, "--chunk_id", "1", "--chunk_num", "1", "--minCoverage", "4"])
    cases[-1]["vcf_sites"] = [["chr20", s] for s in sites] + [["chr1", 1000]]
    run("empty", [], base_args)
    out = dict(ctg=ctg, contig_seq=contig_seq, cases=cases)
    with gzip.open(os.path.join(HERE, "g2_streams.json.gz"), "wt") as f:
        json.dump(out, f, separators=(",", ":"))
    print("g2: %d cases" % len(cases))


# --------------------------------------------------------------------------- batch cases (G3)
_G3_CHILD = r"""
import sys, json, io
sys.path.insert(0, %(here)r)
import refharness as rh
u = rh.load_utils()
lines = json.load(open(%(inp)r))
sys.stdin = io.StringIO("".join(l + "\n" for l in lines))
out = []
for X, pos, alt in u.tensor_generator_from("PIPE", %(bs)d, True, "ont"):
    out.append(dict(shape=list(X.shape), dtype=str(X.dtype), X=X.reshape(-1).tolist(), positions=pos, alt_info=alt))
json.dump(out, open(%(outp)r, "w"))
"""


def fake_line(rng, ctg, pos, C, depth, amp):
    ref33 = rand_seq(rng, 33)
    vals = [rng.randint(-amp, amp) for _ in range(33 * C)]
    alt = "%d-XA %d RT %d" % (depth, depth // 3, depth - depth // 3)
    return "%s\t%d\t%s\t%s\t%s" % (ctg, pos, ref33, " ".join(str(v) for v in vals), alt)


def gen_g3():
    rng = random.Random(SEED + 3)
    cases = []
    for name, C, bs in [("c18", 18, 5), ("c30", 30, 4)]:
        lines = []
        for i, depth in enumerate([20, 144, 216, 217, 218, 250, 500, 2000, 7, 289, 1000]):
            lines.append(fake_line(rng, "chr20", 1000 + i, C, depth, max(3, depth)))
        # explicit rounding probes: 250 -> 72 at depth 500, 7@217 -> 4, -7@217 -> -4
        probe = [250, -250, 7, -7, 1, -1, 0, 216, 217, -217] + [0] * (33 * C - 10)
        lines.append("chr20\t5000\t%s\t%s\t%s" % ("A" * 33, " ".join(map(str, probe)), "500-XA 250 RT 250"))
        lines.append("chr20\t5001\t%s\t%s\t%s" % ("C" * 33, " ".join(map(str, probe)), "217-XA 7 RC 210"))
        inp, outp = "/tmp/c3r_g3_in.json", "/tmp/c3r_g3_out.json"
        json.dump(lines, open(inp, "w"))
        code = _G3_CHILD % dict(here=HERE, inp=inp, outp=outp, bs=bs)
        subprocess.check_call([sys.executable, "-c", code])  # separate process: param.input_shape is mutated
        cases.append(dict(name=name, C=C, batch_size=bs, lines=lines, batches=json.load(open(outp))))
    with open(os.path.join(HERE, "g3_batches.json"), "w") as f:
        json.dump(dict(cases=cases), f, separators=(",", ":"))
    print("g3: %d cases" % len(cases))


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("g1", "all"):
        gen_g1()
    if what in ("g2", "all"):
        gen_g2()
    if what in ("g3", "all"):
        gen_g3()
    if what in ("g4", "all"):
        from make_golden_decode import gen_g4
        gen_g4()
