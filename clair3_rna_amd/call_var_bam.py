"""Drop-in per-chunk driver: accepts the flag set run_clair3_rna passes to `clair3_rna.py call_var_bam`
(run_clair3_rna:684-705, :808-831; parser clair3_rna/call_var_bam.py:336-518) and writes the same
`pileup_{ctg}_{chunk}.vcf` file (header + one row per candidate incl. RefCall rows, file removed when empty)
that src/sort_vcf.py consumes.  Where the reference spawns `pypy create_tensor_pileup | python call_variants`
(clair3_rna/call_var_bam.py:288-295) this runs tensor build + network on one MI355X through libc3r.so and the
decode on the host.  Exit codes: 0 ok, non-zero with a message on stderr on failure (a failing chunk fails the
GNU-parallel step exactly like the reference, run_clair3_rna:868-872).

    python -m clair3_rna_amd.call_var_bam --chkpnt_fn W --bam_fn reads.npz|x.bam --ref_fn ref.fa --call_fn out.vcf \
        --ctgName chr20 --chunk_id 1 --chunk_num 13 --platform ont --pileup ...
"""
import argparse
import os
import sys
from time import time

import numpy as np

from . import altinfo, decode, io, mpileup_compat, vcf


def _env_precision():
    from . import capi           # (ctypes declarations only: libc3r.so is opened when an Engine is made)
    return capi.env_precision()


_TRUE_WORDS = frozenset(["yes", "true", "t", "y", "1", "ture"])      # "ture" / "flase": misspellings the reference's own parser takes
_FALSE_WORDS = frozenset(["no", "false", "f", "n", "0", "flase"])     # (shared/utils.py:143-153); run scripts in the wild may rely on them


def str2bool(v):
    """argparse type of the reference's boolean flags (`--flag True`)."""
    if isinstance(v, bool):
        return v
    word = str(v).strip().lower()
    if word in _TRUE_WORDS:
        return True
    if word in _FALSE_WORDS:
        return False
    raise argparse.ArgumentTypeError("Boolean value expected, got %r" % (v,))


def str_none(v):
    return None if v is None or v.upper() == "NONE" else v


def existing(path):
    """file_path_from semantics (shared/utils.py:81-97): a missing optional file is treated as None."""
    return path if (isinstance(path, str) and os.path.isfile(path)) else None


def chunk_region(contig_len, chunk_id, chunk_num, bed_start=None, bed_end=None):
    """src/create_tensor_pileup.py:380-397 (chunk_id 1-based as on the CLI)."""
    cid = chunk_id - 1
    if bed_start is None:
        size = contig_len // chunk_num + 1 if contig_len % chunk_num else contig_len // chunk_num
        start = size * cid
    else:
        span = bed_end - bed_start
        size = span // chunk_num + 1 if span % chunk_num else span // chunk_num
        start = bed_start + 1 + size * cid
    return start, start + size


def resolve_region(contig_len, ctg, chunk_id, chunk_num, ctg_start=None, ctg_end=None, bed_fn=None, extend_bed=None, vcf_fn=None):
    """(ctg_start, ctg_end, sites or None, extend-BED intervals) of one CHUNK_LIST row, 1-based inclusive, exactly as
    src/create_tensor_pileup.py:375-422 derives them; `sites` == [] means the chunk holds no known site (nothing to do)."""
    ext_iv, bed_start, bed_end = (io.read_bed(extend_bed, ctg) if extend_bed else ([], None, None))
    sites = None
    if bed_fn is None and chunk_id is not None:
        ctg_start, ctg_end = chunk_region(contig_len, chunk_id, chunk_num)
    if bed_fn is not None and chunk_id is not None:
        if bed_start is None:
            sys.exit("[ERROR] ctg_name %s not exists in bed file(%s)." % (ctg, bed_fn))
        ctg_start, ctg_end = chunk_region(0, chunk_id, chunk_num, bed_start, bed_end)
    if vcf_fn is not None and chunk_id is not None:
        all_sites = io.read_vcf_sites(vcf_fn, ctg)
        n = len(all_sites)
        size = n // chunk_num if n % chunk_num == 0 else n // chunk_num + 1
        sites = all_sites[(chunk_id - 1) * size:(chunk_id - 1) * size + size]
        if not sites:
            return ctg_start, ctg_end, sites, ext_iv
        ctg_start, ctg_end = min(sites), max(sites)
    if ctg_start is None or ctg_end is None:
        ctg_start, ctg_end = 1, contig_len
    return ctg_start, ctg_end, sites, ext_iv


def build_parser():
    p = argparse.ArgumentParser(description="Clair3-RNA per-chunk pileup calling on MI355X (drop-in for call_var_bam)")
    a = p.add_argument
    a('--platform', type=str, default="ont")
    a('--bam_fn', type=str, required=True)
    a('--chkpnt_fn', type=str, required=True)
    a('--ref_fn', type=str, required=True)
    a('--call_fn', type=str, default=None)
    a('--vcf_fn', type=str, default=None)
    a('--ctgName', type=str, default=None)
    a('--ctgStart', type=int, default=None)
    a('--ctgEnd', type=int, default=None)
    a('--bed_fn', type=str, nargs='?', default=None)
    a('--sampleName', type=str, nargs='?', default="SAMPLE")
    a('--min_af', type=float, default=None)
    a('--snp_min_af', type=float, default=0.08)
    a('--indel_min_af', type=float, default=0.08)     # call_var_bam.py's own default; run_clair3_rna passes 0.15
    a('--qual', type=int, default=None)
    a('--samtools', type=str, default="samtools")     # never piped (the CIGAR walk runs on the GPU); asked for its version: --mpileup_compat auto
    a('--mpileup_compat', type=str, default=mpileup_compat.env_default(), choices=list(mpileup_compat.CHOICES),
      help="which samtools mpileup text the tensor build restates: auto = ask `--samtools --version` (>= 1.11 -> 1, <= 1.10 -> 0, not "
           "runnable -> 1); 0 = samtools <= 1.10; 1 = samtools >= 1.11 (`+<ins>-<del>`, pads inside insertions).  Default: $C3R_MPILEUP_COMPAT, else auto")
    a('--pypy', type=str, default="pypy3")
    a('--python', type=str, default="python3")
    a('--enable_phasing_model', type=str2bool, default=False)
    a('--minCoverage', type=int, default=4)
    a('--minMQ', type=int, default=5)
    a('--minBQ', type=int, default=0)
    a('--enable_variant_calling_at_sequence_head_and_tail', type=str2bool, default=False)
    a('--enable_padding_in_splice_junction_regions', type=str2bool, default=False)
    a('--extend_bed', nargs='?', type=str, default=None)
    a('--pileup', action='store_true')
    a('--chunk_num', type=int, default=None)
    a('--chunk_id', type=int, default=None)
    a('--show_ref', action='store_false')              # as in the reference: showRef is ON unless given
    a('--cmd_fn', type=str_none, default=None)
    a('--delay', type=int, default=0)
    a('--use_gpu', type=str2bool, default=True)
    a('--gpu_id', type=int, default=int(os.environ.get("C3R_DEVICE", "0")))
    a('--gpu_precision', type=str, default=_env_precision(), choices=["f32", "f16x3", "f16+f8", "auto"],
      help="network arithmetic (include/c3r.h, c3r_set_precision): f16x3 = fp32-equivalent split-f16 (default); auto = the faster fp8-corrected "
           "path where a calibration run through the loaded weights agrees with f16x3 to 4e-5, else f16x3")
    a('--tensor_dump_fn', type=str, default=None, help="DEBUG: also write the create_tensor text lines here")
    for flag in ('--gvcf', '--fast_mode', '--call_snp_only', '--enable_long_indel', '--keep_iupac_bases'):
        a(flag, type=str2bool, default=False)
    for flag in ('--haploid_precise', '--haploid_sensitive', '--add_indel_length', '--debug', '--output_for_ensemble',
                 '--phasing_info_in_bam', '--need_phasing', '--is_from_tables', '--bp_resolution'):
        a(flag, action='store_true')
    a('--need_realignment', action='store_false')
    a('--full_aln_regions', type=str, nargs='?', default=None)
    a('--tensorflow_threads', type=int, default=4)
    a('--temp_file_dir', type=str, default='./')
    a('--base_err', type=float, default=0.001)
    a('--gq_bin_size', type=int, default=5)
    return p


def Run(args, engine=None):
    from . import capi
    t0 = time()
    if args.platform not in ('ont', 'hifi', 'ilmn'):
        sys.exit("[ERROR] Provided platform are not in support platform list [ont, hifi, ilmn]")
    if args.ctgName is None:
        sys.exit("--ctgName must be specified. You can call variants on multiple chromosomes simultaneously.")
    for flag in ('gvcf', 'add_indel_length', 'haploid_precise', 'haploid_sensitive', 'enable_long_indel', 'fast_mode',
                 'call_snp_only', 'keep_iupac_bases', 'output_for_ensemble'):
        if getattr(args, flag):
            sys.exit("[ERROR] --%s is not supported by the MI355X pileup path" % flag)
    if not args.pileup:
        sys.exit("[ERROR] only --pileup calling is implemented by the MI355X path")
    for need in (args.bam_fn, args.ref_fn):
        if not os.path.isfile(need):
            sys.exit("[ERROR] file %s not found" % need)
    ctg = args.ctgName
    bed_fn, vcf_fn, extend_bed, cmd_fn = existing(args.bed_fn), existing(args.vcf_fn), existing(args.extend_bed), existing(args.cmd_fn)
    channels = 30 if args.enable_phasing_model else 18
    fai = {n: L for n, L, _o, _b, _w in io.read_fai(args.ref_fn)}

    # ---- A4: chunk -> coordinates (src/create_tensor_pileup.py:375-422)
    ctg_start, ctg_end, sites, ext_iv = resolve_region(fai.get(ctg, 0), ctg, args.chunk_id, args.chunk_num, args.ctgStart, args.ctgEnd,
                                                       bed_fn, extend_bed, vcf_fn)
    if sites is not None and not sites:
        return 0
    extend_start, extend_end = max(1, ctg_start - 33), ctg_end + 33
    ref_start = max(1, ctg_start - 1000)
    ref_seq = io.fetch_reference(args.ref_fn, ctg, ref_start, ctg_end + 1000)
    if not ref_seq:
        sys.exit("[ERROR] Failed to load reference sequence from file (%s)." % args.ref_fn)

    # ---- which samtools the column text follows: asked in a child process before anything touches the GPU
    compat = mpileup_compat.resolve(getattr(args, "mpileup_compat", "auto"), args.samtools)

    # ---- GPU: tensor build + network
    eng = engine or capi.Engine(args.gpu_id)
    eng.params = capi.default_params()
    eng.set_bed(0, ext_iv if extend_bed else None)
    cbed = io.read_bed(bed_fn, ctg, extend_start, extend_end)[0] if bed_fn else None
    eng.set_bed(1, cbed)
    if sites is not None:
        eng.set_sites(sites)
    eng.set_params(channels=channels, min_mq=args.minMQ, min_coverage=args.minCoverage, snp_min_af=args.snp_min_af,
                   indel_min_af=args.indel_min_af, head_tail=int(args.enable_variant_calling_at_sequence_head_and_tail),
                   splice_padding=int(args.enable_padding_in_splice_junction_regions), genotyping_mode=int(sites is not None),
                   mpileup_compat=compat)
    rs = io.load_reads(args.bam_fn, ctg, extend_start - 1, extend_end)      # mpileup -r ctg:extend_start-extend_end
    eng.load_reads(rs)
    eng.set_reference(ref_start, ref_seq)
    eng.load_weights(io.load_weights(args.chkpnt_fn, channels), channels)
    eng.set_precision(getattr(args, "gpu_precision", "f16x3"))
    n = eng.scan(ctg_start, ctg_end)
    rows = b""
    if n:
        eng.infer(fetch=False)
        qual = args.qual if args.qual is not None else 2            # call_variants.py:1827 default
        # A8 on host threads inside libc3r: ordered alt_info from the per-read tokens, decode, row text
        rows, _n_rows = eng.call_rows_text(ctg, qual=qual, show_ref=args.show_ref)
        if args.tensor_dump_fn:
            sites_out, toks = eng.sites(), eng.tokens()
            raw = eng.tensors(rescaled=False)
            with open(args.tensor_dump_fn, "w") as f:
                for line in altinfo.format_lines(ctg, sites_out, raw, toks, rs, ref_seq, ref_start, padins=eng.pad_insertions()):
                    f.write(line + "\n")
    if args.call_fn:
        vcf.write_chunk_vcf(args.call_fn, vcf.header(args.ref_fn, cmd_fn, args.sampleName), rows)
    if args.chunk_id is not None:
        print("Total processed positions in {} (chunk {}/{}) : {}".format(ctg, args.chunk_id, args.chunk_num, n), file=sys.stderr)
    else:
        print("Total processed positions in {} : {}".format(ctg, n), file=sys.stderr)
    print("Total time elapsed: %.2f s" % (time() - t0), file=sys.stderr)
    if engine is None:
        eng.close()
    return 0


def main(argv=None):
    args = build_parser().parse_args(argv)
    try:
        return Run(args)
    except SystemExit:
        raise
    except Exception as e:   # any failure must fail the chunk loudly (no CPU fallback)
        print("[ERROR] call_var_bam (MI355X path) failed: %s" % e, file=sys.stderr)
        return 1


if __name__ == "__main__":
    sys.exit(main())
