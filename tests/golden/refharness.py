"""Harness that imports the *reference* Clair3-RNA Python modules to generate golden vectors.

This file only works in the build container, where the upstream tree is mounted read-only at
/root/reference.  It is never imported by tests, by the product, by bench.py or by smoke(): the
committed JSON fixtures under tests/golden/ are what travel.  Nothing from the reference is
copied; we call its functions and record (input, output) pairs.

Recipe follows SURVEY.md Appendix G.
"""
import io
import os
import sys
import types

REF_ROOT = "/root/reference"


def _install_stubs():
    sys.dont_write_bytecode = True
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    if "cffi" not in sys.modules:  # src/utils.py:31 imports cffi for the gvcf-only inline C
        m = types.ModuleType("cffi")
        m.FFI = object
        sys.modules["cffi"] = m
    if "tensorflow" not in sys.modules:  # clair3_rna/call_variants.py:34
        sys.modules["tensorflow"] = types.ModuleType("tensorflow")


def load_create_tensor():
    _install_stubs()
    import src.create_tensor_pileup as ctp  # noqa
    return ctp


def load_utils():
    _install_stubs()
    import clair3_rna.utils as u  # noqa
    return u


def load_call_variants():
    _install_stubs()
    import clair3_rna.call_variants as cv  # noqa
    import shared.param_p as p
    cv.param = p
    return cv


class _FakeProc(object):
    def __init__(self, text):
        self.stdout = io.StringIO(text)
        self.returncode = 0

    def wait(self):
        return 0

    def poll(self):
        return 0


class _Sink(io.StringIO):
    def close(self):  # TensorStdout.__del__ closes sys.stdout (create_tensor_pileup.py:305-310)
        pass


def run_create_tensor(rows, contig_seq, ctg_name, argv_extra, fai_len=None, workdir="/tmp/c3r_golden"):
    """Run the reference's create_tensor_pileup main() over fake `samtools mpileup` rows.

    rows        : list of mpileup text rows (no trailing newline)
    contig_seq  : full contig sequence (1-based position p is contig_seq[p-1])
    argv_extra  : list of CLI args (e.g. ['--ctgStart','1','--ctgEnd','400'])
    returns (output_lines, observed_samtools_cmds)
    """
    ctp = load_create_tensor()
    os.makedirs(workdir, exist_ok=True)
    fa = os.path.join(workdir, "ref.fa")
    with open(fa, "w") as f:
        f.write(">%s\n%s\n" % (ctg_name, contig_seq))
    L = fai_len if fai_len is not None else len(contig_seq)
    with open(fa + ".fai", "w") as f:
        f.write("%s\t%d\t%d\t%d\t%d\n" % (ctg_name, L, len(ctg_name) + 2, L, L + 1))

    cmds = []

    def fake_popen(args, **kw):
        cmds.append(list(args))
        if len(args) > 1 and args[1] == "mpileup":
            return _FakeProc("".join(r + "\n" for r in rows))
        # everything else (gzip -fdc for vcf_fn) runs for real
        from subprocess import Popen, PIPE
        return Popen(args, stdout=PIPE, universal_newlines=True)

    def fake_ref(samtools_execute_command, fasta_file_path, regions):
        reg = regions[0]
        if ":" in reg:
            se = reg.split(":")[1]
            s, e = se.split("-")
            return contig_seq[int(s) - 1:int(e)].upper()
        return contig_seq.upper()

    ctp.subprocess_popen = fake_popen
    ctp.reference_sequence_from = fake_ref
    old_stdout, old_argv = sys.stdout, sys.argv
    sink = _Sink()
    sys.stdout = sink
    sys.argv = ["create_tensor_pileup", "--bam_fn", "fake.bam", "--ref_fn", fa, "--ctgName", ctg_name] + list(argv_extra)
    try:
        ctp.main()
    finally:
        sys.stdout = old_stdout
        sys.argv = old_argv
    text = sink.getvalue()
    lines = text.split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    return lines, cmds
