"""Per-chunk VCF text: header contract of shared/utils.py:261-316 (get_header) and file handling of
clair3_rna/call_variants.py:1474-1475,1594-1599 (header first, file removed when it holds no record)."""
import os

from .io import read_fai

VERSION = "0.2.2"          # shared/param_p.py:2 — stamped as ##clair3_rna_version like the reference does

_FIXED = [
    '##fileformat=VCFv4.2',
    '##source=Clair3-RNA',
    '##clair3_rna_version=%s' % VERSION,
    '##FILTER=<ID=PASS,Description="All filters passed">',
    '##FILTER=<ID=LowQual,Description="Low quality variant">',
    '##FILTER=<ID=RefCall,Description="Reference call">',
    '##FILTER=<ID=RNAEditing,Description="RNA editing site tagged by REDIportal dataset">',
    '##INFO=<ID=A,Number=0,Type=Flag,Description="RNA editing site from ATLAS dataset in REDIportal">',
    '##INFO=<ID=R,Number=0,Type=Flag,Description="RNA editing site from RADAR dataset in REDIportal">',
    '##INFO=<ID=D,Number=0,Type=Flag,Description="RNA editing site from DARNED dataset in REDIportal">',
    '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
    '##FORMAT=<ID=GQ,Number=1,Type=Integer,Description="Genotype Quality">',
    '##FORMAT=<ID=DP,Number=1,Type=Integer,Description="Approximate read depth (reads with MQ<5 or selected by \'samtools view -F 2316\' are filtered)">',
    '##FORMAT=<ID=AD,Number=R,Type=Integer,Description="Allelic depths for the ref and alt alleles in the order listed">',
    '##FORMAT=<ID=AF,Number=1,Type=Float,Description="Observed allele frequency in reads, for each ALT allele, in the same order as listed, or the REF allele for a RefCall">',
]


def header(ref_fn=None, cmd_fn=None, sample_name="SAMPLE"):
    lines = list(_FIXED)
    if ref_fn is not None and os.path.exists(ref_fn):
        lines.insert(3, "##reference=%s" % ref_fn)
    if cmd_fn is not None and os.path.exists(cmd_fn):
        cmd = open(cmd_fn).read().rstrip()
        if cmd:
            lines.insert(3, "##cmdline=%s" % cmd)      # inserted after ##reference => ends up before it
    text = "\n".join(lines) + "\n"
    if ref_fn is not None:
        for name, length, _o, _b, _w in read_fai(ref_fn):
            text += "##contig=<ID=%s,length=%s>\n" % (name, length)
        text += "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t%s" % sample_name
    return text


def write_chunk_vcf(path, header_text, rows):
    """Header + rows; the file is deleted again when there is no record (call_variants.py:1594-1599).
    `rows` is a list of row strings or one bytes blob of newline-terminated rows (Engine.call_rows_text)."""
    if isinstance(rows, (bytes, bytearray)):
        with open(path, "wb") as f:
            f.write(header_text.encode() + b"\n")
            f.write(rows)
    else:
        with open(path, "w") as f:
            print(header_text, file=f)
            for r in rows:
                print(r, file=f)
    if not rows:
        os.remove(path)
        return False
    return True
