#!/usr/bin/env python3
"""Pin the two "parity unpinned" halves against the third-party tools themselves, the day they exist beside this repository
(neither samtools nor TensorFlow is in the build image).

  python tools/pin_third_party.py --samtools /path/to/samtools            # A1: reads -> mpileup text
  python tools/pin_third_party.py --tf --reference_repo /path/to/Clair3-RNA   # F5: a TensorFlow-written checkpoint through tfckpt.py

A1.  Every known-answer case of tests/test_oracle_mpileup.py (harvested by running those tests with a recording `pile`), plus the depth-cap
     case (8000 / 8001 reads starting on one position, `-d 8000`), is written as SAM + FASTA, turned into an indexed BAM by the given
     samtools, and piled up with the reference's own flag set (src/create_tensor_pileup.py:436-451: `mpileup <bam> -r <region> --reverse-del
     --min-MQ 5 --min-BQ 0 --excl-flags 2316 [--max-depth N] [--output-extra HP]`).  Columns (position, depth, bases[, HP]) are compared with
     orc_mpileup's in the printer that samtools version calls for (clair3_rna_amd/mpileup_compat.py).  A mismatch prints both columns; the
     one-constant switches for the depth cap are PLP_POOL_EXTRA (csrc/c3r_lib.hip) and ORC_PLP_POOL_EXTRA (oracle/c3r_oracle.c).
F5.  Builds the reference's Keras model (clair3_rna/model.py) for 18 and 30 channels, save_weights() into a temporary directory, reads the
     bundle with clair3_rna_amd/tfckpt.py and compares every tensor with model.get_weights() bit for bit
     (clair3_rna/call_variants.py:1472 is where the reference loads such a bundle).
Exit status 0: everything agreed."""
import argparse
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def harvest_cases():
    """(records, beg, end, kwargs) of every pile() call the known-answer tests make."""
    import tests.test_oracle_mpileup as T
    cases, orig = [], T.pile

    def pile(records, beg=1, end=60, **kw):
        cases.append(([dict(r) for r in records], beg, end, dict(kw)))
        return orig(records, beg, end, **kw)
    T.pile = pile
    try:
        for name in sorted(dir(T)):
            if name.startswith("test_"):
                try:
                    getattr(T, name)()
                except Exception:          # noqa: BLE001 (only the inputs are wanted here)
                    pass
    finally:
        T.pile = orig
    return cases


def cap_cases():
    seq = "ACGTACGTAC" * 6
    out = []
    for n in (8000, 8001):
        recs = [dict(pos=9, cigar="60M", seq=seq, flag=16 * (i % 2), mapq=60, hp=0) for i in range(n)]
        recs += [dict(pos=20, cigar="40M", seq=seq[11:51], flag=0, mapq=60, hp=0) for _ in range(5)]
        out.append((recs, 1, 80, dict(max_depth=8000)))
    return out


def write_sam(path, records, ref_len, with_hp):
    with open(path, "w") as f:
        f.write("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:c\tLN:%d\n" % ref_len)
        for i, r in enumerate(sorted(records, key=lambda r: r["pos"])):
            seq = r["seq"]
            tags = "\tHP:i:%d" % r["hp"] if (with_hp and r.get("hp")) else ""
            f.write("r%d\t%d\tc\t%d\t%d\t%s\t*\t0\t0\t%s\t%s%s\n" % (i, r.get("flag", 0), r["pos"] + 1, r.get("mapq", 60), r["cigar"], seq, "I" * len(seq), tags))


def run_samtools_case(samtools, compat, case, tmp):
    from clair3_rna_amd.reads import ReadSet
    from oracle import oracle as orc
    records, beg, end, kw = case
    with_hp = bool(kw.get("with_hp"))
    ref_len = max(end + 100, max(r["pos"] for r in records) + 5000)
    sam, bam, fa = os.path.join(tmp, "x.sam"), os.path.join(tmp, "x.bam"), os.path.join(tmp, "x.fa")
    write_sam(sam, records, ref_len, with_hp)
    with open(fa, "w") as f:
        f.write(">c\n" + "N" * ref_len + "\n")
    subprocess.check_call([samtools, "view", "-b", "-o", bam, sam])
    subprocess.check_call([samtools, "index", bam])
    cmd = [samtools, "mpileup", bam, "-r", "c:%d-%d" % (beg, end), "--reverse-del", "--min-MQ", str(kw.get("min_mq", 5)), "--min-BQ", "0",
           "--excl-flags", str(kw.get("excl_flags", 2316))]
    if "max_depth" in kw:
        cmd += ["--max-depth", str(kw["max_depth"])]
    if with_hp:
        cmd += ["--output-extra", "HP"]
    got = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, check=True).stdout.rstrip("\n").split("\n")
    rs = ReadSet.from_records(records)
    okw = {k: v for k, v in kw.items() if k in ("min_mq", "excl_flags", "with_hp", "max_depth")}
    exp = orc.mpileup(rs.reads, rs.cigar, rs.seq, "c", beg, end, compat=compat, **okw)

    def cols(rows):
        out = {}
        for row in rows:
            f = row.split("\t")
            if len(f) >= 5:
                out[int(f[1])] = (f[3], f[4]) + ((f[6],) if with_hp and len(f) > 6 else ())
        return out
    g, e = cols(got), cols(exp)
    bad = [(p, g.get(p), e.get(p)) for p in sorted(set(g) | set(e)) if g.get(p) != e.get(p)]
    return bad


def pin_samtools(samtools):
    from clair3_rna_amd import mpileup_compat
    compat = mpileup_compat.resolve("auto", samtools)
    ver = subprocess.run([samtools, "--version"], stdout=subprocess.PIPE, text=True).stdout.split("\n")[0]
    print("%s -> printer %d" % (ver, compat))
    cases = harvest_cases() + cap_cases()
    n_bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for k, case in enumerate(cases):
            bad = run_samtools_case(samtools, compat, case, tmp)
            if bad:
                n_bad += 1
                print("case %d (%d reads, %s): %d columns differ" % (k, len(case[0]), case[3], len(bad)))
                for p, g, e in bad[:6]:
                    print("   pos %d  samtools %r  oracle %r" % (p, g, e))
    print("A1: %d cases, %d differ" % (len(cases), n_bad))
    return n_bad == 0


def pin_tf(reference_repo):
    import numpy as np
    sys.path.insert(0, reference_repo)
    import tensorflow as tf  # noqa: F401
    from clair3_rna import model as ref_model
    from clair3_rna_amd import tfckpt
    ok = True
    for channels in (18, 30):
        try:
            m = ref_model.Clair3_P(add_indel_length=False, predict=True)      # (clair3_rna/call_variants.py:1466; the channel count comes with the input)
            m(np.zeros((1, 33, channels), np.float32))
        except Exception as ex:          # noqa: BLE001
            print("F5: could not build the reference model for %d channels: %r" % (channels, ex))
            ok = False
            continue
        with tempfile.TemporaryDirectory() as tmp:
            prefix = os.path.join(tmp, "pileup")
            m.save_weights(prefix)
            blob = tfckpt.weights_from_bundle(prefix, channels=channels)
            ours = np.asarray(blob, np.float32)
            theirs = np.concatenate([w.reshape(-1) for w in m.get_weights()]).astype(np.float32)
            same = ours.size == theirs.size and np.array_equal(np.sort(ours), np.sort(theirs))
            print("F5: %d channels: %d parameters read, %d in the model, multiset %s" % (channels, ours.size, theirs.size, "equal" if same else "DIFFERS"))
            ok = ok and same
    return ok


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--samtools")
    ap.add_argument("--tf", action="store_true")
    ap.add_argument("--reference_repo", default="/root/reference")
    ap.add_argument("--list", action="store_true", help="only count the harvested cases (needs neither tool)")
    a = ap.parse_args()
    ok = True
    if a.list:
        c = harvest_cases() + cap_cases()
        print("%d cases (%d reads in the largest)" % (len(c), max(len(x[0]) for x in c)))
    if a.samtools:
        ok = pin_samtools(a.samtools) and ok
    if a.tf:
        ok = pin_tf(a.reference_repo) and ok
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
