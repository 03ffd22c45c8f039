"""Which `samtools mpileup` printer the tensor build restates (c3r_params_t.mpileup_compat).

The reference pipes whatever `--samtools` points at and only demands version >= 1.10 (run_clair3_rna:159,166); its Docker image
installs an unpinned bioconda `clair3` environment (Dockerfile:56), which resolves to samtools >= 1.11 today.  The text of a column
differs between the two families for one CIGAR pattern: from 1.11 on (htslib bam_plp_insertion) an insertion that is followed at once
by a deletion prints `+2TT-1N` and the pads inside a run of I ops print as '*' / '#'; up to 1.10 the column shows `+2TT` only.  The
reference's parser (src/create_tensor_pileup.py:151-163) makes two tokens of the first form, so D / d, D1 / d1, the indel AF gate and
alt_info of such a column depend on the samtools the user has.  The drivers therefore ask that samtools:

    --mpileup_compat auto   (default; C3R_MPILEUP_COMPAT overrides the default)
        run `<--samtools> --version` in a child process BEFORE anything touches the GPU and take 1 for >= 1.11, 0 for <= 1.10;
        when the binary is missing or prints no version: 1 (what the reference's own image would run)
    --mpileup_compat 0 | 1  the <= 1.10 / >= 1.11 printer, whatever is installed
"""
import os
import re
import subprocess
import sys

CHOICES = ("auto", "0", "1")


def env_default():
    v = os.environ.get("C3R_MPILEUP_COMPAT", "auto")
    if v not in CHOICES:
        raise SystemExit("C3R_MPILEUP_COMPAT=%r: must be one of %s" % (v, ", ".join(CHOICES)))
    return v


def samtools_version(samtools):
    """(major, minor) of `<samtools> --version`, or None when it cannot be run or prints no version."""
    try:
        out = subprocess.run([samtools, "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=20).stdout.decode("utf-8", "replace")
    except (OSError, subprocess.SubprocessError, ValueError):
        return None
    m = re.search(r"samtools\s+(\d+)\.(\d+)", out)
    return (int(m.group(1)), int(m.group(2))) if m else None


def resolve(choice, samtools="samtools", log=None):
    """-> 0 or 1 for c3r_params_t.mpileup_compat; logs the decision (one line on stderr)."""
    log = log or (lambda m: print(m, file=sys.stderr))
    if choice in ("0", "1", 0, 1):
        v = int(choice)
        log("[INFO] mpileup_compat = %d (the samtools %s printer, as requested)" % (v, ">= 1.11" if v else "<= 1.10"))
        return v
    if choice != "auto":
        raise SystemExit("--mpileup_compat %r: must be one of %s" % (choice, ", ".join(CHOICES)))
    ver = samtools_version(samtools or "samtools")
    if ver is None:
        log("[INFO] mpileup_compat = 1: `%s --version` could not be run; taking the samtools >= 1.11 printer, which is what the reference's "
            "Docker image installs (--mpileup_compat 0 selects the <= 1.10 text)" % (samtools or "samtools"))
        return 1
    v = 1 if ver >= (1, 11) else 0
    log("[INFO] mpileup_compat = %d: %s is samtools %d.%d (%s)" % (v, samtools, ver[0], ver[1], ">= 1.11: `+<ins>-<del>` and padded insertions are shown"
                                                                   if v else "<= 1.10: an insertion hides the deletion behind it"))
    return v
