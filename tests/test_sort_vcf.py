"""F4 merge step (clair3_rna_amd/sort_vcf.py) against golden G6, produced by the reference's own sort_vcf_from /
sort_vcf_from_stdin (tests/golden/make_golden_sortvcf.py).  CPU-only."""
import gzip
import json
import os

import pytest

from clair3_rna_amd import sort_vcf

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(gzip.open(os.path.join(HERE, "golden", "g6_sortvcf.json.gz"), "rt"))


@pytest.mark.parametrize("ci", range(len(CASES) - 1))
def test_directory_merge_matches_reference(ci, tmp_path):
    case = CASES[ci]
    a = case["args"]
    d = tmp_path / "in"
    d.mkdir()
    for fn, text in case["files"].items():
        (d / fn).write_text(text)
    table = None
    if a.get("tag"):
        table = {}
        if case["rediportal"] is not None:
            redi = tmp_path / "redi.txt.gz"
            with gzip.open(str(redi), "wt") as f:
                f.write(case["rediportal"])
            tags = set(a["filter_tag"].split(":")) if a.get("filter_tag") else None
            table = sort_vcf.load_rediportal(str(redi), case["contigs"], tags)
    out, out_nt = str(tmp_path / "o.vcf"), str(tmp_path / "o_nt.vcf")
    sort_vcf.merge_chunk_vcfs(str(d), out, case["contigs"], "pileup", ".vcf", a["qual"], a["show_ref"], table, out_nt,
                              listing=case["listing"], log=lambda *_: None)
    assert open(out).read() == case["out"]
    if case["out_no_tagging"] is not None:
        assert open(out_nt).read() == case["out_no_tagging"]
        if case["rediportal"] is not None:
            assert "RNAEditing" in case["out"] and "RNAEditing" not in case["out_no_tagging"]


def test_stdin_mode_matches_reference(tmp_path):
    case = CASES[-1]
    out = str(tmp_path / "o.vcf")
    sort_vcf.merge_stream(case["stdin"].splitlines(keepends=True), out)
    assert open(out).read() == case["out"]


def test_cli_empty_inputs_and_bgzf_output(tmp_path):
    d = tmp_path / "in"
    d.mkdir()
    (tmp_path / "CONTIGS").write_text("chr1\n")
    out = str(tmp_path / "o.vcf")
    argv = ["--input_dir", str(d), "--vcf_fn_prefix", "pileup", "--output_fn", out, "--contigs_fn", str(tmp_path / "CONTIGS")]
    assert sort_vcf.main(argv) == 0 and open(out).read() == ""               # nothing to merge: empty file, like the reference
    hdr = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS\n"
    (d / "pileup_chr1_1.vcf").write_text(hdr + "chr1\t7\t.\tA\t.\t3.00\tRefCall\t.\tGT\t0/0\n")
    assert sort_vcf.main(argv) == 0 and open(out).read() == ""               # only RefCall rows and no --show_ref: empty as well
    (d / "pileup_chr1_2.vcf").write_text(hdr + "chr1\t9\t.\tA\tG\t1.50\tPASS\t.\tGT\t0/1\nchr1\t8\t.\tC\tT\t30.00\tPASS\t.\tGT\t1/1\n")
    assert sort_vcf.main(argv + ["--compress_vcf", "True", "--qual", "2"]) == 0
    assert not os.path.exists(out)
    text = gzip.open(out + ".gz", "rt").read()                               # BGZF is a valid multi-member gzip
    assert text == hdr + "chr1\t8\t.\tC\tT\t30.00\tPASS\t.\tGT\t1/1\nchr1\t9\t.\tA\tG\t1.50\tLowQual\t.\tGT\t0/1\n"
    assert open(out + ".gz", "rb").read()[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    with pytest.raises(SystemExit):
        sort_vcf.main(["--input_dir", str(tmp_path / "nope"), "--output_fn", out, "--contigs_fn", str(tmp_path / "CONTIGS")])
