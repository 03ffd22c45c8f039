#!/usr/bin/env python3
"""Golden G6: per-chunk VCF files -> merged VCF, through the reference's own sort_vcf_from (src/sort_vcf.py:123-292).

Run only in the build container:  python tests/golden/make_golden_sortvcf.py
Writes tests/golden/g6_sortvcf.json.gz = list of cases {files: {name: text}, listing: [names in the order the
reference saw them], contigs: [...], args: {...}, rediportal: text|None, out: text, out_no_tagging: text|None}.
Only inputs and outputs are recorded; bgzip/tabix (absent here) are switched off with --compress_vcf False.
"""
import gzip
import json
import os
import random
import shutil
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

HDR = ("##fileformat=VCFv4.2\n##FILTER=<ID=PASS,Description=\"All filters passed\">\n##FILTER=<ID=LowQual,Description=\"Low quality variant\">\n"
       "##FILTER=<ID=RefCall,Description=\"Reference call\">\n{extra}##contig=<ID=chr1,length=1000000>\n##contig=<ID=chr11,length=900000>\n"
       "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\n")


def row(rng, ctg, pos, kind=None):
    ref = rng.choice("ACGT")
    kind = kind or rng.choice(["snp", "snp", "snp", "ref", "ins", "del", "multi"])
    if kind == "ref":
        return "%s\t%d\t.\t%s\t.\t%.2f\tRefCall\t.\tGT:GQ:DP:AD:AF\t0/0:%d:%d:%d:%.4f\n" % (ctg, pos, ref, rng.uniform(0, 30), 12, 20, 19, 0.95)
    if kind == "snp":
        alt = rng.choice([b for b in "ACGT" if b != ref])
    elif kind == "ins":
        alt = ref + "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 4)))
    elif kind == "del":
        ref = ref + "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 4)))
        alt = ref[0]
    else:
        alt = ",".join(rng.sample([b for b in "ACGT" if b != ref], 2))
    q = rng.choice([0.0, 1.99, 2.0, 2.01, 7.99, 8.0, 8.5, rng.uniform(0, 40)])
    filt = "PASS" if q >= 2 else "LowQual"
    gt = rng.choice(["0/1", "1/1", "1/2"])
    return "%s\t%d\t.\t%s\t%s\t%.2f\t%s\t.\tGT:GQ:DP:AD:AF\t%s:%d:%d:%d,%d:%.4f\n" % (ctg, pos, ref, alt, q, filt, gt, int(q), 30, 12, 18, 0.6)


def make_case(rng, n_chunks, extra_header_in_some, with_empty_contig):
    files = {}
    # chr1 chunks overlap by a few positions (adjacent chunks re-call the +-33 bp halo): same position, possibly different rows
    for ctg, prefix in (("chr1", "pileup_chr1_"), ("chr11", "pileup_chr11_")):
        pos_pool = sorted(rng.sample(range(100, 90000), 60))
        per = len(pos_pool) // n_chunks
        for c in range(n_chunks):
            mine = pos_pool[c * per:(c + 1) * per + (3 if c + 1 < n_chunks else 0)]      # 3 shared with the next chunk
            rng.shuffle(mine) if rng.random() < 0.3 else None
            body = "".join(row(rng, ctg, p) for p in mine)
            extra = "##cmdline=run %d\n" % c if (extra_header_in_some and c % 2) else ""
            if body:
                files["%s%d.vcf" % (prefix, c + 1)] = HDR.format(extra=extra) + body
    files["pileup_chr1_99.vcf"] = HDR.format(extra="")                  # header-only file
    files["other_chr1_1.vcf"] = HDR.format(extra="") + row(rng, "chr1", 5)   # wrong prefix: ignored
    files["pileup_chr1_7.txt"] = "junk\n"                               # wrong suffix: ignored
    return files


def main():
    import refharness as rh
    rh._install_stubs()
    import src.sort_vcf as sv
    rng = random.Random(20240422 + 6)
    work = "/tmp/c3r_golden_sortvcf"
    cases = []
    arg_sets = [
        dict(qual=2, show_ref=False, tag=False),
        dict(qual=8, show_ref=True, tag=False),
        dict(qual=None, show_ref=False, tag=False),
        dict(qual=8, show_ref=False, tag=True, filter_tag=None),
        dict(qual=2, show_ref=True, tag=True, filter_tag="A:D"),
        dict(qual=2, show_ref=False, tag=True, filter_tag=None, missing_db=True),
    ]
    for ci, a in enumerate(arg_sets):
        shutil.rmtree(work, ignore_errors=True)
        os.makedirs(os.path.join(work, "in"))
        files = make_case(rng, n_chunks=3 + ci % 2, extra_header_in_some=bool(ci % 2), with_empty_contig=True)
        for fn, text in files.items():
            open(os.path.join(work, "in", fn), "w").write(text)
        listing = sorted(files.keys())
        rng.shuffle(listing)
        sv.os = types.SimpleNamespace(path=os.path, listdir=lambda d, _l=listing: list(_l))   # pin the directory order
        contigs = ["chr11", "chr1", "chrUn_x"] if ci % 2 else ["chr1", "chr11"]
        open(os.path.join(work, "CONTIGS"), "w").write("\n".join(contigs) + "\n")
        redi_text = None
        redi_fn = None
        if a.get("tag"):
            # REDIportal table: header row, then contig, pos, ref, alt, strand, db, ...
            rows_all = [l.split("\t") for t in files.values() for l in t.split("\n") if l and l[0] != "#"]
            picks = rng.sample(rows_all, 25)
            lines = ["Region\tPosition\tRef\tEd\tStrand\tdb\ttype"]
            for k, r in enumerate(picks):
                ref, alt = r[3], r[4]
                if k % 5 == 0:
                    alt = "G" if alt != "G" else "A"              # position known, different edit: no tag
                lines.append("%s\t%s\t%s\t%s\t+\t%s\tALU" % (r[0], r[1], ref, alt, rng.choice(["A", "D", "R"])))
            lines.append("chrZ\t5\tA\tG\t+\tA\tALU")               # contig not processed
            lines.append("chr1\tnotanumber\tA\tG\t+\tA\tALU")      # unparsable position is skipped
            redi_text = "\n".join(lines) + "\n"
            redi_fn = os.path.join(work, "redi.txt.gz")
            if not a.get("missing_db"):
                with gzip.open(redi_fn, "wt") as f:
                    f.write(redi_text)
        args = types.SimpleNamespace(output_fn=os.path.join(work, "out.vcf"), input_dir=os.path.join(work, "in"), vcf_fn_prefix="pileup",
                                     vcf_fn_suffix=".vcf", sample_name="S1", ref_fn=None, contigs_fn=os.path.join(work, "CONTIGS"),
                                     compress_vcf=False, qual=a["qual"], output_no_tagging_fn=os.path.join(work, "out_nt.vcf"),
                                     show_ref=a["show_ref"], cmd_fn=None, tag_variant_using_readiportal=a.get("tag") or None,
                                     readiportal_source_fn=redi_fn, readiportal_database_filter_tag=a.get("filter_tag"))
        sv.sort_vcf_from(args)
        out = open(args.output_fn).read()
        out_nt = open(args.output_no_tagging_fn).read() if a.get("tag") else None
        cases.append(dict(files=files, listing=listing, contigs=contigs, args=a, rediportal=None if a.get("missing_db") else redi_text,
                          out=out, out_no_tagging=out_nt))
        print("case %d: %d files, %d output lines, %d RNAEditing" % (ci, len(files), out.count("\n"), out.count("RNAEditing")))
    # stdin mode (src/sort_vcf.py:85-121)
    import io as _io
    text = "".join(cases[1]["files"][k] for k in cases[1]["listing"] if k.startswith("pileup") and k.endswith(".vcf"))
    sv.stdin = _io.StringIO(text)
    args = types.SimpleNamespace(output_fn=os.path.join(work, "out_stdin.vcf"))
    sv.sort_vcf_from_stdin(args)
    cases.append(dict(stdin=text, out=open(args.output_fn).read()))
    with gzip.open(os.path.join(HERE, "g6_sortvcf.json.gz"), "wt") as f:
        json.dump(cases, f)
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
