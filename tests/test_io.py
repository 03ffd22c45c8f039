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


def test_native_fasta_fetch_equals_the_python_slice(tmp_path):
    """c3r_fasta_fetch (parallel pread over the .fai geometry) against io.fetch_reference: random slices of a contig that spans
    several pread pieces, lower-case input, CRLF line ends, a last line without its newline; a .fai that does not match the
    file is an error, not a shifted sequence."""
    import pytest
    from clair3_rna_amd import bamio
    rng = np.random.default_rng(8)
    big = "".join(rng.choice(list("ACGTacgtNn"), size=9_000_001).tolist())
    fa = str(tmp_path / "r.fa")
    io.write_fasta(fa, [("c1", "acgtn" * 21), ("big", big), ("tail", "ACGTTGCA" * 9 + "AC")], width=70)
    with open(fa, "rb+") as f:                                  # drop the file's last newline
        f.seek(-1, 2)
        assert f.read(1) == b"\n"
        f.seek(-1, 2)
        f.truncate()
    fai = {r[0]: r for r in io.read_fai(fa)}
    assert bamio.fasta_fetch(fa, fai["big"]).tobytes().decode() == big.upper()
    assert bamio.fasta_fetch(fa, fai["big"], upper=False).tobytes().decode() == big
    assert bamio.fasta_fetch(fa, fai["tail"]).tobytes() == b"ACGTTGCA" * 9 + b"AC"
    assert bamio.fasta_fetch(fa, fai["c1"], 2, 12).tobytes().decode() == ("ACGTN" * 21)[2:12]
    assert bamio.fasta_fetch(fa, fai["c1"], 50, 50).size == 0
    for _ in range(40):
        a = int(rng.integers(0, len(big)))
        b = int(min(len(big), a + rng.integers(1, 300_000)))
        for thr in (1, 3):
            assert bamio.fasta_fetch(fa, fai["big"], a, b, threads=thr).tobytes().decode() == io.fetch_reference(fa, "big", a + 1, b)
    # CRLF line ends
    fa2 = str(tmp_path / "crlf.fa")
    seq = "".join(rng.choice(list("ACGT"), size=1000).tolist())
    with open(fa2, "wb") as f:
        f.write(b">x\r\n")
        off = f.tell()
        for i in range(0, len(seq), 60):
            f.write(seq[i:i + 60].encode() + b"\r\n")
    row = ("x", len(seq), off, 60, 62)
    assert bamio.fasta_fetch(fa2, row).tobytes().decode() == seq
    assert bamio.fasta_fetch(fa2, row, 59, 121).tobytes().decode() == seq[59:121]
    # an index that does not describe the file
    for bad in (("x", len(seq), off, 61, 63), ("x", len(seq), off, 60, 61), ("x", len(seq) + 500, off, 60, 62), ("x", len(seq), off + 1, 60, 62)):
        with pytest.raises(IOError):
            bamio.fasta_fetch(fa2, bad)
