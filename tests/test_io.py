"""Host I/O: BAM/BGZF round trip, FASTA slices, BED semantics, VCF header contract (shared/utils.py:261-316)."""
import gzip
import os

import numpy as np

from clair3_rna_amd import bam, io, synth, vcf
from clair3_rna_amd.reads import ReadSet


def test_bam_roundtrip(tmp_path):
    ref, rs, _ = synth.small_case(seed=41, ref_len=20000, n_genes=4, depth=12, phased=True)
    rs2 = ReadSet.from_records([(7, "3M70000N4M", "ACGTACG", 16, 3, 0)])
    p = str(tmp_path / "x.bam")
    bam.write_bam(p, [("chr20", len(ref)), ("chrM", 100000)], {"chr20": rs, "chrM": rs2})
    got = bam.read_contig(p, "chr20")
    for f in ("pos", "n_cigar", "l_seq", "flag", "mapq", "hp"):
        assert np.array_equal(got.reads[f], rs.reads[f]), f
    assert np.array_equal(got.cigar, rs.cigar) and np.array_equal(got.seq, rs.seq)
    m = bam.read_contig(p, "chrM")
    assert len(m) == 1 and m.reads["pos"][0] == 7 and m.reads["flag"][0] == 16 and m.cigar.tolist() == rs2.cigar.tolist()
    assert len(bam.read_contig(p, "chrX")) == 0
    assert open(p, "rb").read()[-28:] == bam._BGZF_EOF


def test_fasta_slices_and_bed(tmp_path):
    seq = synth.random_reference(1234, 5)
    fa = str(tmp_path / "r.fa")
    io.write_fasta(fa, [("c1", "acgtn" * 20), ("chr20", seq)], width=50)
    assert io.fetch_reference(fa, "chr20", 1, 1234) == seq
    assert io.fetch_reference(fa, "chr20", 49, 152) == seq[48:152]
    assert io.fetch_reference(fa, "chr20", -5, 10) == seq[:10]
    assert io.fetch_reference(fa, "chr20", 1200, 99999) == seq[1199:]
    assert io.fetch_reference(fa, "c1", 3, 12) == ("ACGTN" * 20)[2:12]
    bed = str(tmp_path / "b.bed.gz")
    with gzip.open(bed, "wt") as f:
        f.write("#c\nchr20\t100\t200\nchr20\t300\t300\nchrX\t1\t2\nchr20\t5000\t6000\n")
    iv, s, e = io.read_bed(bed, "chr20")
    assert iv == [(100, 200), (300, 301), (5000, 6000)] and (s, e) == (100, 6000)
    iv, _, _ = io.read_bed(bed, "chr20", keep_start=150, keep_end=400)
    assert iv == [(100, 200), (300, 301)]


def test_vcf_header_contract(tmp_path):
    fa = str(tmp_path / "r.fa")
    io.write_fasta(fa, [("chr20", "ACGT" * 40), ("chrX", "TTGA" * 30)])
    cmd = str(tmp_path / "CMD")
    open(cmd, "w").write("run_clair3_rna --bam_fn x --ref_fn y\n")
    h = vcf.header(fa, cmd, "S1").split("\n")
    assert h[:3] == ["##fileformat=VCFv4.2", "##source=Clair3-RNA", "##clair3_rna_version=0.2.2"]
    assert h[3] == "##cmdline=run_clair3_rna --bam_fn x --ref_fn y" and h[4] == "##reference=" + fa
    assert [x[:9] for x in h[5:9]] == ["##FILTER="] * 4 and sum(x.startswith("##FORMAT=") for x in h) == 5
    assert h[-3:] == ["##contig=<ID=chr20,length=160>", "##contig=<ID=chrX,length=120>",
                      "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1"]
    out = str(tmp_path / "o.vcf")
    assert vcf.write_chunk_vcf(out, "\n".join(h), []) is False and not os.path.exists(out)
    assert vcf.write_chunk_vcf(out, "\n".join(h), ["chr20\t5\t.\tA\tG\t9.00\tPASS\t.\tGT\t0/1"]) and os.path.exists(out)
