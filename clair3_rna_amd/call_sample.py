"""Whole-sample pileup calling in ONE process: STEP 1 + STEP 2 of run_clair3_rna without the per-chunk processes and files.

The reference runs `parallel ... clair3_rna.py call_var_bam ... :::: tmp/CHUNK_LIST` (run_clair3_rna:678-708: one Python +
pypy + samtools process group per (contig, chunk_id, chunk_num) row, each writing tmp/pileup_output/pileup_{ctg}_{chunk}.vcf)
and then `sort_vcf` over that directory (run_clair3_rna:710-726).  Here the same CHUNK_LIST (contig selection and chunk
counts of run_clair3_rna:310-449) is walked contig by contig: all chunks of a contig go through the tensor build in one
c3r_pileup_scan_regions call and through the network as one batch, the rows are decoded by libc3r's host threads and merged
in memory by sort_vcf.SampleMerger.  The result — `<output_dir>/<output_prefix>.vcf.gz` (+ .tbi, + `_no_tagging`) — is
byte-identical to running call_var_bam per CHUNK_LIST row and merging with sort_vcf (tests/test_gpu_sample.py).

One thing is pinned that the reference leaves open.  Adjacent chunks overlap by the 33 bp halo; the seam position itself is
emitted by both, and with head/tail calling both also emit the candidates inside the halo — from different windows (one
chunk's stream ends there, the other's begins).  sort_vcf keeps the row of whichever per-chunk file os.listdir returns last
(src/sort_vcf.py:204-236), i.e. an arbitrary one.  Here the LATER chunk's row is kept, always.

The host stages overlap on threads (the GIL is released inside libc3r / libc3r_io):
    fetch    BAM region fetch + reference slice of the next contigs          (libc3r_io.so, FASTA read; --fetch_threads)
    context  per contig, on one of --contexts GPU contexts, each with its own thread and HIP stream: read normalisation +
             uploads, tensor build, network, alt_info + genotype decode + row text (libc3r.so) — while one context waits
             for its kernels the others prepare or decode
    merge    per-record rules + order, in calling order as the contigs come out (libc3r_io.so)

Not covered (use the reference's own orchestration around call_var_bam for these): whatshap/longphase phasing between the
two passes (external tools), gVCF.  `--enable_phasing_model` here expects an already haplotagged BAM (HP tags) and runs the
30-channel pass only — the second half of run_clair3_rna:729-852.

    python -m clair3_rna_amd.call_sample --bam_fn x.bam --ref_fn ref.fa --pileup_model_path W --output_dir out
"""
import argparse
import os
import sys
import threading
from concurrent.futures import ThreadPoolExecutor
from time import time

import numpy as np

from . import io, mpileup_compat, sort_vcf, vcf
from .reads import READ_DTYPE, ReadSet
from .call_var_bam import existing, resolve_region


def _env_precision():
    from . import capi           # (ctypes declarations only: libc3r.so is opened when an Engine is made)
    return capi.env_precision()

MAJOR_CONTIGS_ORDER = ["chr" + str(a) for a in list(range(1, 23)) + ["X", "Y"]] + [str(a) for a in list(range(1, 23)) + ["X", "Y"]]
CHUNK_SIZE = 5000000            # shared/param_p.py:91
EXPAND = 33                     # param.no_of_positions: split_extend_bed widens every interval by one window


def _bed_contigs(bed_fn):
    names = []
    with io._open_text(bed_fn) as f:
        for row in f:
            if row.strip() and row[0] != "#":
                c = row.split()[0]
                if c not in names:
                    names.append(c)
    return names


def _vcf_contigs(vcf_fn):
    names = set()
    with io._open_text(vcf_fn) as f:
        for row in f:
            if row[0] != "#":
                names.add(row.split(None, 1)[0])
    return names


def plan_chunks(ref_fn, ctg_name=None, include_all_ctgs=False, bed_fn=None, vcf_fn=None, chunk_size=CHUNK_SIZE, chunk_num=None):
    """-> ([contig, ...] in calling order, {contig: chunk_num}): run_clair3_rna:310-389 + :441-449 (CHUNK_LIST)."""
    listed = set(ctg_name.split(",")) if ctg_name else None
    in_bed = set(_bed_contigs(bed_fn)) if bed_fn else None
    in_vcf = _vcf_contigs(vcf_fn) if vcf_fn else None
    contig_set = set(listed) if listed else set()
    if listed:
        if in_bed is not None:
            contig_set &= in_bed
        if in_vcf is not None:
            contig_set &= in_vcf
    else:
        if in_bed is not None:
            contig_set |= in_bed
        if in_vcf is not None:
            contig_set |= in_vcf
    restricted = bool(in_bed is not None or listed or in_vcf is not None)
    chunks = {}
    for name, length, _o, _b, _w in io.read_fai(ref_fn):
        if not include_all_ctgs and not restricted and name not in MAJOR_CONTIGS_ORDER:
            continue
        if in_bed is not None and name not in in_bed:
            continue
        if (listed or in_vcf is not None) and name not in contig_set:
            continue
        contig_set.add(name)
        n = length // chunk_size + 1 if length % chunk_size else length // chunk_size
        chunks[name] = chunk_num if chunk_num else max(n, 1)
    order = MAJOR_CONTIGS_ORDER + sorted(contig_set - set(MAJOR_CONTIGS_ORDER))
    contigs = sorted((c for c in contig_set if c in chunks), key=order.index)
    return contigs, chunks


def split_extend_bed(bed_fn, out_dir, contig_set):
    """run_clair3_rna:268-296: per-contig BED files with every interval widened by 33 bp on both sides."""
    per = {}
    with io._open_text(bed_fn) as f:
        for i, row in enumerate(f):
            if not row.strip() or row[0] == "#":
                continue
            c = row.strip().split()
            if contig_set and c[0] not in contig_set:
                continue
            s, e = int(c[1]), int(c[2])
            if e < s or s < 0 or e < 0:
                sys.exit("[ERROR] Invalid BED input at the %d-th row %s %d %d" % (i + 1, c[0], s, e))
            per.setdefault(c[0], []).append("%s %d %d" % (c[0], max(0, s - EXPAND), max(0, e + EXPAND)))
    os.makedirs(out_dir, exist_ok=True)
    for name, rows in per.items():
        with open(os.path.join(out_dir, name), "w") as f:
            f.write("\n".join(rows))


def build_parser():
    p = argparse.ArgumentParser(description="Clair3-RNA pileup calling of a whole sample on MI355X (in-process STEP 1 + STEP 2 of run_clair3_rna)")
    a = p.add_argument
    a("-b", "--bam_fn", type=str, required=True)
    a("-f", "--ref_fn", type=str, required=True)
    a("-o", "--output_dir", type=str, required=True)
    a("-p", "--platform", type=str, default="ont")
    a("--pileup_model_path", type=str, required=True, help="checkpoint prefix (…/variables), as passed to call_var_bam --chkpnt_fn")
    a("--phased_pileup_model_path", type=str, default=None)
    a("--enable_phasing_model", action="store_true", help="30-channel pass on an already haplotagged BAM")
    a("-c", "--ctg_name", type=str, default=None)
    a("--bed_fn", type=str, default=None)
    a("--genotyping_mode_vcf_fn", type=str, default=None)
    a("-q", "--qual", type=int, default=None)
    a("--snp_min_af", type=float, default=0.08)
    a("--indel_min_af", type=float, default=0.15)
    a("--min_coverage", type=int, default=4)
    a("--min_mq", type=int, default=5)
    a("--chunk_size", type=int, default=CHUNK_SIZE)
    a("--chunk_num", type=int, default=None)
    a("-s", "--sample_name", type=str, default="SAMPLE")
    a("--output_prefix", type=str, default="output")
    a("--include_all_ctgs", action="store_true")
    a("--print_ref_calls", action="store_true")
    a("--enable_variant_calling_at_sequence_head_and_tail", action="store_true")
    a("--enable_padding_in_splice_junction_regions", action="store_true")
    a("--tag_variant_using_readiportal", action="store_true")
    a("--readiportal_source_fn", type=str, default=None)
    a("--readiportal_database_filter_tag", type=str, default=None)
    a("--no_compress", action="store_true", help="leave <prefix>.vcf uncompressed (tests)")
    a("--samtools", type=str, default="samtools", help="as in run_clair3_rna; never piped, only asked for its version (--mpileup_compat auto)")
    a("--mpileup_compat", type=str, default=mpileup_compat.env_default(), choices=list(mpileup_compat.CHOICES),
      help="which samtools mpileup text the tensor build restates: auto = ask `--samtools --version` (>= 1.11 -> 1, <= 1.10 -> 0, not "
           "runnable -> 1); 0 = samtools <= 1.10; 1 = samtools >= 1.11.  Default: $C3R_MPILEUP_COMPAT, else auto")
    a("--gpu_id", type=int, default=None, help="default: $C3R_DEVICE, else LOCAL_RANK under torch.distributed.run, else 0")
    a("--gpu_precision", type=str, default=_env_precision(), choices=["f32", "f16x3", "f16+f8", "auto"],
      help="network arithmetic (c3r_set_precision): f16x3 = fp32-equivalent split-f16 (default); auto = the faster fp8-corrected path where a "
           "calibration run through the loaded weights agrees with f16x3 to 4e-5, else f16x3")
    a("--fetch_threads", type=int, default=8, help="threads that fetch alignments (long contigs as several position ranges, BGZF inflate on C3R_FETCH_INFLATE threads each) and reference slices ahead of the contexts; capped by the rank's share of the cores under torch.distributed")
    a("--contexts", type=int, default=2, help="GPU contexts (each with its own host thread and HIP stream) working side by side: while one waits for its kernels the other normalises reads or decodes (every context sizes its own device buffers on its first contig; first uses take turns)")
    return p


def _all_ranks_ok(dist, world, err, what):
    """Rendezvous of all ranks that also carries a failure flag: a rank that failed still arrives, and then every rank raises.
    (A rank that simply returned would leave the others in dist.barrier() until the gloo timeout.)"""
    if world > 1:
        import torch
        flag = torch.tensor([1 if err is not None else 0], dtype=torch.int32)
        dist.all_reduce(flag)
        if err is None and int(flag.item()):
            raise RuntimeError("another rank failed while %s" % what)
    if err is not None:
        raise err


class _Fetcher(object):
    """Stage 1: a contig's alignments + reference slice, one BAM handle per worker thread.  A long contig is fetched as several
    position ranges on several threads (`plan` / `part` / `join`): the first contig of a sample is then ready after a fraction of
    the 0.4 s a cold handle needs for a whole chromosome, and the contexts start that much earlier."""

    PART_BP = 24_000_000                 # shortest range worth a fetch of its own

    def __init__(self, bam_fn, ref_fn):
        self.bam_fn, self.ref_fn, self.tls = bam_fn, ref_fn, threading.local()
        self.handles, self.fai = [], None
        self.lock = threading.Lock()

    def _handle(self):
        from . import bamio
        bf = getattr(self.tls, "bf", None)
        if bf is None:
            bf = self.tls.bf = bamio.BamFile(self.bam_fn, threads=int(os.environ.get("C3R_FETCH_INFLATE", "8")))
            with self.lock:
                self.handles.append(bf)
        return bf

    def plan(self, length, max_parts):
        """-> [(beg0, end0), ...] position ranges of one contig, in order (one range: the whole contig)."""
        if self.bam_fn.endswith(".npz") or max_parts <= 1:
            return [(0, None)]
        n = max(1, min(max_parts, length // int(os.environ.get("C3R_FETCH_PART_BP", self.PART_BP))))      # (the variable: tests split small contigs)
        if n > 1 and not self._handle().has_index:
            n = 1
        edges = [length * k // n for k in range(n + 1)]
        return [(edges[k], edges[k + 1] if k + 1 < n else None) for k in range(n)]

    def part(self, ctg, beg0, end0):
        """The alignments that START in [beg0, end0) (a fetch returns everything that overlaps the range: what started before it
        belongs to the range before) -> ReadSet views, offsets relative to this part."""
        if self.bam_fn.endswith(".npz"):
            return io.load_reads(self.bam_fn, ctg)
        rs = self._handle().fetch(ctg, beg0, end0)
        if beg0 > 0 and len(rs.reads):
            i0 = int(np.searchsorted(rs.reads["pos"], beg0, side="left"))
            if i0:
                if i0 >= len(rs.reads):
                    return _empty_reads()
                c0, s0 = int(rs.reads["cigar_off"][i0]), int(rs.reads["seq_off"][i0])
                out = ReadSet.__new__(ReadSet)
                out.reads, out.cigar, out.seq = rs.reads[i0:], rs.cigar[c0:], rs.seq[s0:]
                out.base = (c0, s0)
                return out
        return rs

    def reference(self, ctg, length):
        from . import bamio
        with self.lock:
            if self.fai is None:
                self.fai = {r[0]: r for r in io.read_fai(self.ref_fn)}
        # the whole contig, upper-cased, line ends dropped, by parallel pread (c3r_fasta_fetch) — a 250-Mb chromosome went
        # through Python's bytes.replace in 0.4 s with the GIL held, which stalled every other thread of the process
        return bamio.fasta_fetch(self.ref_fn, self.fai[ctg], 0, length)

    @staticmethod
    def join(parts):
        """Parts in position order -> one ReadSet (arrays from c3r_io_alloc, offsets moved)."""
        parts = [p_ for p_ in parts if len(p_.reads)]
        if not parts:
            return _empty_reads()
        if len(parts) == 1 and getattr(parts[0], "base", (0, 0)) == (0, 0):
            return parts[0]
        from . import bamio
        out = ReadSet.__new__(ReadSet)
        out.reads = bamio.huge_empty(sum(len(p_.reads) for p_ in parts), READ_DTYPE)
        out.cigar = bamio.huge_empty(sum(len(p_.cigar) for p_ in parts), np.uint32)
        out.seq = bamio.huge_empty(sum(len(p_.seq) for p_ in parts), np.uint8)
        r0 = c0 = s0 = 0
        for p_ in parts:
            nr, nc, ns = len(p_.reads), len(p_.cigar), len(p_.seq)
            bc, bs = getattr(p_, "base", (0, 0))
            out.reads[r0:r0 + nr] = p_.reads
            out.reads["cigar_off"][r0:r0 + nr] += np.uint32((c0 - bc) & 0xffffffff)      # (modular: the sum is the new offset)
            out.reads["seq_off"][r0:r0 + nr] += np.uint64((s0 - bs) & 0xffffffffffffffff)
            out.cigar[c0:c0 + nc] = p_.cigar
            out.seq[s0:s0 + ns] = p_.seq
            r0 += nr; c0 += nc; s0 += ns
        return out

    def __call__(self, ctg, length):
        """The whole contig on the calling thread -> (ReadSet, reference array or b"", seconds)."""
        t0 = time()
        rs = self.join([self.part(ctg, 0, None)])
        ref = self.reference(ctg, length) if len(rs.reads) else b""
        return rs, ref, time() - t0

    def close(self):
        for bf in self.handles:
            bf.close()


def _empty_reads():
    out = ReadSet.__new__(ReadSet)
    out.reads, out.cigar, out.seq = np.zeros(0, READ_DTYPE), np.zeros(0, np.uint32), np.zeros(0, np.uint8)
    return out


def Run(args, log=None):
    from . import capi
    log = log or (lambda m: print(m, file=sys.stderr))
    t_all = time()
    # which samtools the column text follows: asked in a child process before anything touches the GPU (rank 0's line is the one printed)
    compat = mpileup_compat.resolve(getattr(args, "mpileup_compat", "auto"), getattr(args, "samtools", "samtools"),
                                    log if int(os.environ.get("RANK", "0")) == 0 else (lambda m: None))
    # one process per GPU (`python -m torch.distributed.run --nproc-per-node N -m clair3_rna_amd.call_sample ...`): contigs are
    # dealt to the ranks largest-first, every rank leaves the merged records of its contigs under tmp/parts/, rank 0 puts the
    # file together.  No data-path collective: torch.distributed (gloo) is the barrier, nothing else.
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            dist.init_process_group("gloo")
        if args.gpu_id is None:
            args.gpu_id = int(os.environ.get("LOCAL_RANK", str(rank)))
        # one process per GPU shares the node's cores with its peers: this rank's slice of them (its GPU's NUMA node when the
        # topology says which), and thread counts cut to match — decode (C3R_THREADS), fetch, compression
        from . import shard
        n_thr, _cpus = shard.host_budget(apply=True)
        args.fetch_threads = max(1, min(args.fetch_threads, n_thr // 2 or 1))
        os.environ.setdefault("C3R_FETCH_INFLATE", str(max(1, n_thr // args.fetch_threads)))    # inflate threads per fetch handle
    elif os.environ.get("C3R_HOST_SLICE"):
        # one process on the host slice a rank of an N-GPU run would get (tools/host_slice.py)
        from . import shard
        n_thr, _cpus = shard.host_budget(apply=True)
        args.fetch_threads = max(1, min(args.fetch_threads, n_thr // 2 or 1))
        os.environ.setdefault("C3R_FETCH_INFLATE", str(max(1, n_thr // args.fetch_threads)))
    if args.gpu_id is None:
        args.gpu_id = int(os.environ.get("C3R_DEVICE", "0"))
    for need in (args.bam_fn, args.ref_fn):
        if not os.path.isfile(need):
            sys.exit("[ERROR] file %s not found" % need)
    bed_fn, vcf_fn = existing(args.bed_fn), existing(args.genotyping_mode_vcf_fn)
    channels = 30 if args.enable_phasing_model else 18
    model = args.phased_pileup_model_path if args.enable_phasing_model else args.pileup_model_path
    if model is None:
        sys.exit("[ERROR] --phased_pileup_model_path is required with --enable_phasing_model")
    out_dir = args.output_dir
    os.makedirs(os.path.join(out_dir, "tmp"), exist_ok=True)
    suffix = "_enable_phasing" if args.enable_phasing_model else ""
    out_fn = os.path.join(out_dir, args.output_prefix + suffix + ".vcf")
    out_nt_fn = os.path.join(out_dir, args.output_prefix + "_no_tagging" + suffix + ".vcf")

    contigs, chunk_nums = plan_chunks(args.ref_fn, args.ctg_name, args.include_all_ctgs, bed_fn, vcf_fn, args.chunk_size, args.chunk_num)
    if not contigs:
        log("[WARNING] Exit calling because no contig was found in BAM!")
        return 0
    fai = {n: L for n, L, _o, _b, _w in io.read_fai(args.ref_fn)}
    priv = os.path.join(out_dir, "tmp") if rank == 0 else os.path.join(out_dir, "tmp", "rank%d" % rank)
    os.makedirs(priv, exist_ok=True)
    split_dir = os.path.join(priv, "split_beds")
    if bed_fn:
        split_extend_bed(bed_fn, split_dir, set(contigs))
    cmd_fn = os.path.join(out_dir, "tmp", "CMD")                        # run_clair3_rna:613-667: stamped into the VCF header
    if rank == 0 and not os.path.exists(cmd_fn):
        with open(cmd_fn, "w") as f:
            f.write(" ".join(sys.argv) + "\n")
    bam_fn = args.bam_fn
    if bam_fn.endswith(".bam"):
        from . import bamio
        with bamio.BamFile(bam_fn) as probe:
            indexed = probe.has_index
        if not indexed:
            # without an index every contig would cost a pass over the whole file: build one next to a link in tmp/
            # (run_clair3_rna insists on an existing index, :469-477; samtools is not a dependency here).  Rank 0 alone builds it,
            # under temporary names that are renamed into place, and the other ranks open the BAM only after the barrier: nobody
            # ever sees a missing link or a half-written .bai
            link = os.path.join(out_dir, "tmp", "input.bam")
            build_err = None
            if rank == 0:
                try:
                    tmp_link, tmp_bai = link + ".tmp%d" % os.getpid(), link + ".bai.tmp%d" % os.getpid()
                    if os.path.lexists(tmp_link):
                        os.remove(tmp_link)
                    os.symlink(os.path.abspath(bam_fn), tmp_link)
                    log("[INFO] %s has no .bai: building %s.bai" % (bam_fn, link))
                    try:
                        bamio.index_build(tmp_link, tmp_bai)
                        os.replace(tmp_bai, link + ".bai")
                        os.replace(tmp_link, link)
                    finally:
                        for t_ in (tmp_link, tmp_bai):
                            if os.path.lexists(t_):
                                os.remove(t_)
                except Exception as e:           # the other ranks are waiting at the barrier: reach it, then fail together
                    build_err = e
            _all_ranks_ok(dist, world, build_err, "building the BAM index")
            bam_fn = link
    all_contigs = contigs
    parts_dir = os.path.join(out_dir, "tmp", "parts")
    if world > 1:
        from . import shard
        os.makedirs(parts_dir, exist_ok=True)
        dist.barrier()                                                  # CMD is in place before anybody builds the header
        # contigs are dealt by the WORK they hold ("by read count ... from the read index"): the mapped reads the BAM
        # index reports per contig (pseudo-bin 37450), else the compressed bytes its records span, else its length.  RNA coverage is
        # nowhere near proportional to length (chr19 against chr13), so a deal by length balances the wrong thing.
        costs, basis = shard.contig_costs(bam_fn, all_contigs, fai)
        plan = shard.lpt_assign(costs, world)
        mine = plan[rank]
        if rank == 0:
            log("[INFO] %d contigs dealt to %d ranks by %s: load imbalance %.3f" % (len(all_contigs), world, basis, shard.imbalance(costs, plan)))
        contigs = [all_contigs[i] for i in mine]

    table = None
    if args.tag_variant_using_readiportal:
        src = args.readiportal_source_fn
        if src is None or src.upper() == "NONE" or not os.path.exists(src):
            log("[WARNING] Enabled tagging variant using readiportal, but --readiportal_source_fn %s file not found, skip tagging!" % src)
            table = {}
        else:
            tags = set(args.readiportal_database_filter_tag.split(":")) if args.readiportal_database_filter_tag is not None else None
            table = sort_vcf.load_rediportal(src, contigs, tags)

    weights = io.load_weights(model, channels)
    n_ctx = max(1, args.contexts)
    engines = [None] * n_ctx
    engine_ready = [threading.Event() for _ in range(n_ctx)]
    engine_errs = []
    engine_threads = []

    def make_engines():
        """The GPU contexts: created, given the weights and the arithmetic once the first fetches are under way (0.1-0.2 s that used to come
        before the first BAM byte was read).  Each context on a thread of its own: context 0 takes the first contig the moment IT is ready
        — HIP start-up and its own weights, 0.2 s in a fresh process — while the others' weights are still being packed (they are not needed
        before the second contig has been fetched).  Returns at once; context_worker waits for its engine."""
        def bring_up(k):
            try:
                if k > 0:
                    # (the process' first c3r_create starts HIP, and two contexts that pack and calibrate their weights at once take twice as long
                    # each: context 0 first — it is the one the first contig waits for —, the others are ready long before their contig is fetched)
                    engine_ready[0].wait()
                t0_ = time()
                e = capi.Engine(args.gpu_id)
                engines[k] = e
                mark("ctx%d" % k, "create", t0_)
                t0_ = time()
                e.load_weights(weights, channels)             # (packing the weights into the kernels' layouts is host work: side by side)
                mark("ctx%d" % k, "weights", t0_)
                t0_ = time()
                e.set_precision(args.gpu_precision)
                mark("ctx%d" % k, "precision", t0_)
                if k == 0 and args.gpu_precision != "f16x3":
                    log("[INFO] network arithmetic: %s -> %s (calibration max |dP| %s)" % ((args.gpu_precision,) + tuple(e.precision())))
            except BaseException as ex:
                engine_errs.append(ex)
            finally:
                engine_ready[k].set()
        for k in range(n_ctx):
            t_ = threading.Thread(target=bring_up, args=(k,), name="c3r-bringup%d" % k, daemon=True)
            engine_threads.append(t_)
            t_.start()

    def engine_of(k):
        """Context k's engine, once it is up (raises what its bring-up raised)."""
        engine_ready[k].wait()
        if engine_errs:
            raise engine_errs[0]
        return engines[k]
    qual_rows = args.qual if args.qual is not None else 2              # call_variants.py:1827 (STEP 1 never passes --qual)
    qual_merge = args.qual if args.qual is not None else 2             # sort_vcf's own default
    header = vcf.header(args.ref_fn, cmd_fn, args.sample_name) + "\n"
    # (compressed output is written as it is produced — bgzip blocks and the tabix index grow contig by contig — so that no
    # compression pass is left after the last contig)
    merger = sort_vcf.SampleMerger(out_fn, header, qual_merge, args.print_ref_calls, table, out_nt_fn, stream_gz=not args.no_compress) if rank == 0 else None

    def device_stage(eng, ctg, rs, ref):
        """-> number of candidates left resident in `eng` (rows are produced by decode_stage)."""
        extend_bed = existing(os.path.join(split_dir, ctg)) if bed_fn else None
        regions, site_sets, ext_iv = [], [], []
        for k in range(1, chunk_nums[ctg] + 1):
            a, b, sites, ext_iv = resolve_region(fai[ctg], ctg, k, chunk_nums[ctg], None, None, bed_fn, extend_bed, vcf_fn)
            if sites is not None and not sites:
                continue
            regions.append((a, b))
            site_sets.append(sites)
        if not regions:
            return 0
        eng.params = capi.default_params()
        eng.set_bed(0, ext_iv if extend_bed else None)
        eng.set_bed(1, io.read_bed(bed_fn, ctg)[0] if bed_fn else None)
        eng.set_params(channels=channels, min_mq=args.min_mq, min_coverage=args.min_coverage, snp_min_af=args.snp_min_af,
                       indel_min_af=args.indel_min_af, head_tail=int(args.enable_variant_calling_at_sequence_head_and_tail),
                       splice_padding=int(args.enable_padding_in_splice_junction_regions), genotyping_mode=int(vcf_fn is not None),
                       mpileup_compat=compat)
        t = [time()]
        eng.load_reads(rs); t.append(time())
        eng.set_reference(1, ref, upper_view=not isinstance(ref, (bytes, str))); t.append(time())      # (the fetcher's array: upper-cased, used in place)
        if vcf_fn is None:
            eng.begin_batch()
            n = eng.scan_regions(regions)
            eng.end_batch()
            n = eng.n_candidates
            t.append(time())
            if n:
                # (launched as soon as the tensors exist, side by side with the other context's pass: making the contexts take
                # turns for the network — one copies while the other computes — measured 3.1-3.3 s against 2.7-2.8 s)
                eng.infer(fetch=False)
            t.append(time())
            if os.environ.get("C3R_TIMING"):
                log("[device_stage %s] load_reads %.0f ms, set_reference %.0f ms, scan %.0f ms, infer launch %.0f ms (%d reads, %d sites)"
                    % (ctg, 1e3 * (t[1] - t[0]), 1e3 * (t[2] - t[1]), 1e3 * (t[3] - t[2]), 1e3 * (t[4] - t[3]), len(rs.reads), n))
            return n
        return [(r, s) for r, s in zip(regions, site_sets)]            # genotyping mode: one scan per chunk (its own site list)

    def decode_stage(eng, ctg, todo):
        if isinstance(todo, list):
            rows = b""
            for (a, b), sites in todo:
                eng.set_sites(sites)
                if eng.scan(a, b):
                    eng.infer(fetch=False)
                    rows += eng.call_rows_text(ctg, qual=qual_rows, show_ref=args.print_ref_calls)[0]
            return rows
        rows = eng.call_rows_text(ctg, qual=qual_rows, show_ref=args.print_ref_calls)[0] if todo else b""
        dump = getattr(args, "debug_dump", None)
        if dump is not None and todo:                                  # tests / debugging: what the rows were made from
            dump[ctg] = dict(sites=eng.sites(), tokens=eng.tokens(), probs=eng.fetch_probs(todo), tensors=eng.tensors(), rows=rows)
        return rows

    part_counts = {}
    edits_cache, edits_lock = [], threading.Lock()

    def edits_by_contig():
        """The REDIportal table as per-contig entry lists, built once for all the per-contig mergers of this rank."""
        with edits_lock:
            if not edits_cache:
                d = {}
                for (c, pos), hit in (table or {}).items():
                    d.setdefault(c, []).append((pos, hit[0], hit[1]))
                edits_cache.append(d)
        return edits_cache[0]

    def merge_contig(ctg, rows):
        if world == 1:
            merger.add_contig(ctg, rows)
        else:                                 # this contig's records on their own; rank 0 concatenates in calling order
            k = all_contigs.index(ctg)
            m = sort_vcf.SampleMerger(os.path.join(parts_dir, "%05d.vcf" % k), "", qual_merge, args.print_ref_calls, table,
                                      os.path.join(parts_dir, "%05d_nt.vcf" % k))
            if table:
                m._edits = edits_by_contig()
            m.add_contig(ctg, rows)
            m.out.close()
            if m.out_nt:
                m.out_nt.close()
            part_counts[k] = (m.n_read, m.n_kept, m.n_tagged)

    fetcher = _Fetcher(bam_fn, args.ref_fn)
    t_setup = time() - t_all
    # contigs fetched (or being fetched) but not yet through their context: every context busy + every fetch thread running ahead.
    # (n_ctx + 2 starved the contexts: a fetch takes ~100 ms, a context needs a new contig every ~35 ms)
    slots = threading.BoundedSemaphore(n_ctx + max(2, args.fetch_threads))
    stats = dict(fetch=0.0, dev=0.0, sites=0)
    lock = threading.Lock()

    timeline = os.environ.get("C3R_TIMING") is not None
    def mark(ctg, what, t0):
        if timeline:
            log("[timeline] %-6s %-8s %7.3f -> %7.3f s" % (ctg, what, t0 - t_all, time() - t_all))


    def submit_fetch(pool, ctg):
        """A contig's fetch as tasks of the fetch pool — one per position range (_Fetcher.plan) and one for the reference — joined
        by whichever finishes last.  -> Future of (ReadSet, reference, seconds)."""
        from concurrent.futures import Future
        t0 = time()
        ranges = fetcher.plan(fai[ctg], max(1, args.fetch_threads))
        out = Future()
        state = dict(left=len(ranges) + 1, parts=[None] * len(ranges), ref=b"", err=None)
        lk = threading.Lock()

        def done_one():
            with lk:
                state["left"] -= 1
                last = state["left"] == 0
            if not last:
                return
            try:
                if state["err"] is not None:
                    raise state["err"]
                rs = fetcher.join(state["parts"])
                mark(ctg, "fetch", t0)
                if timeline:
                    log("[timeline-fetch %s] %d range(s): %d reads, %d CIGAR ops, %.0f MB of bases, %.0f Mb of reference" % (ctg, len(ranges), len(rs.reads), len(rs.cigar), len(rs.seq) / 1e6, len(state["ref"]) / 1e6))
                out.set_result((rs, state["ref"] if len(rs.reads) else b"", time() - t0))
            except BaseException as e:
                out.set_exception(e)

        def part_task(k, beg, end):
            try:
                state["parts"][k] = fetcher.part(ctg, beg, end)
            except BaseException as e:
                state["err"] = e
            done_one()

        def ref_task():
            try:
                state["ref"] = fetcher.reference(ctg, fai[ctg])
            except BaseException as e:
                state["err"] = e
            done_one()

        for k, (beg, end) in enumerate(ranges):
            pool.submit(part_task, k, beg, end)
        pool.submit(ref_task)
        return out

    def context_task(eng, ctg, fut):
        """One contig on the context whose thread took it from the queue (contigs are taken in calling order by whichever context
        is free): host preparation + uploads, tensor build, network, snapshot.  Contexts work side by side — while one waits for
        its kernels another queues its uploads and scans."""
        try:
            rs, ref, dt = fut.result()
            if not len(rs.reads):
                return None
            # (A context sizes its device buffers on its first contig.  Round 2 made first uses take turns, each on a quiet GPU,
            # because a multi-GB hipMalloc under another context's kernels took 0.3-0.5 s; with the network's buffers reserved
            # ahead (c3r_reserve) what is left is cheaper than the wait: 1.38-1.47 s against 1.58-1.60 s on 22 half-length contigs.)
            t0 = time()
            todo = device_stage(eng, ctg, rs, ref)
            t1 = time()
            mark(ctg, "device", t0)
            with lock:
                stats["fetch"] += dt
                stats["dev"] += t1 - t0
                stats["sites"] += todo if isinstance(todo, int) else 0
            detach = isinstance(todo, int) and todo and hasattr(eng, "rows_begin") and getattr(args, "debug_dump", None) is None
            if detach:
                # the decode inputs leave the context as a host snapshot (c3r_rows_begin): this thread goes on to the next contig
                # while a decode worker turns the snapshot into rows and — single process — merges them (c3r_vcf_merge)
                # (snapshots hold ~0.3 GB of staging and their contig's reference: the contexts may not run further ahead of the decode
                # pool than it has workers + 2)
                snap_slots.acquire()
                try:
                    # (without --print_ref_calls the sites the decoder's early RefCall exit would drop anyway stay on the device; the decoder
                    # reads inserted bases from the fetched arrays in place — the snapshot keeps them alive)
                    snap = eng.rows_begin(drop_ref_calls=not args.print_ref_calls)
                except BaseException:
                    snap_slots.release()
                    raise
                mark(ctg, "snapshot", t1)
                fut_d = decode_pool.submit(decode_task, snap, ctg)     # (the look-ahead slot is free: the fetched arrays are done with)
                eng_decodes[engines.index(eng)].append(fut_d)
                return fut_d
            rows = decode_stage(eng, ctg, todo)
            mark(ctg, "decode", t1)
            return rows
        finally:
            slots.release()

    def decode_task(snap, ctg):
        try:
            t0 = time()
            # (rows stay a uint8 array from here to the compressed piece: no 100-MB bytes objects built under the GIL)
            rows = snap.decode(ctg, qual=qual_rows, show_ref=args.print_ref_calls, as_array=True)[0]
            mark(ctg, "decode", t0)
            t1 = time()
            if world == 1:
                res = ("merged", merger.merge_only(ctg, rows))
            else:
                merge_contig(ctg, rows)                    # this rank's part files of the contig, written here on the worker
                res = ("merged", None)
            mark(ctg, "merge", t1)
            return res
        finally:
            snap_slots.release()

    def context_worker(k):
        """Thread of context k: contigs from the shared queue until the end marker, then — once the decodes that still read this
        context's snapshots are through — the context is released: device and page-locked memory go back while the last contigs
        are still being decoded and written."""
        try:
            eng = engine_of(k)
        except BaseException as e:           # the bring-up failed: every contig this worker takes fails with that
            while True:
                item = ctx_queue.get()
                if item is None:
                    return
                if item[2].set_running_or_notify_cancel():
                    item[2].set_exception(e)
        if reserve_sites and hasattr(eng, "reserve"):
            # the network's buffers (8.9 GB for a full slice: 0.25-0.4 s of a first hipMalloc) while the first fetch is under way
            t0 = time()
            try:
                eng.reserve(reserve_sites)
            except Exception:
                pass                      # (the first infer() sizes them, and reports whatever is wrong)
            mark("ctx%d" % k, "reserve", t0)
        while True:
            item = ctx_queue.get()
            if item is None:
                break
            ctg, fut, out = item
            if not out.set_running_or_notify_cancel():          # (cancelled by the failure path, which released its slot)
                continue
            try:
                out.set_result(context_task(eng, ctg, fut))
            except BaseException as e:
                out.set_exception(e)
        t0 = time()
        for f in list(eng_decodes[k]):
            try:
                f.result()
            except BaseException:
                pass                      # (reported by the main loop, which holds the same future)
        if not stop.is_set():
            eng.close()
            mark("ctx%d" % k, "close", t0)

    work_err = None
    t_merge = 0.0
    results = []
    n_sites = t_fetch = t_dev = 0
    called = []
    import queue
    from concurrent.futures import Future
    # candidates per kb of contig: ~3 on the synthetic GRCh38-sized samples, so a contig past ~50 Mb fills a whole network slice
    reserve_sites = min(262144, int(max(fai[c] for c in contigs) * 0.005)) if contigs else 0
    ctx_queue = queue.Queue()                                            # (contig, fetch future, result future) in calling order; None ends a worker
    ctx_threads = [threading.Thread(target=context_worker, args=(k,), name="c3r-ctx%d" % k, daemon=True) for k in range(n_ctx)]
    # snapshots -> rows (-> merged records), beside the contexts.  A large contig's merge is 0.2-0.7 s on one thread when every
    # candidate is a record: with n_ctx + 1 workers the snapshots queued up behind three merges (full-length GRCh38 timeline)
    n_dec = int(os.environ.get("C3R_DECODE_WORKERS", "0")) or max(2, min(8, (n_thr if world > 1 else (os.cpu_count() or 8)) // 4))
    decode_pool = ThreadPoolExecutor(n_dec)
    snap_slots = threading.Semaphore(n_dec + 2)                                # snapshots taken but not decoded yet
    eng_decodes = [[] for _ in range(n_ctx)]                                  # decode futures per context (its finish task waits for them)
    stop = threading.Event()

    def stop_workers():
        """Failure path: nothing may still be inside a c3r_* call (or queued to make one) when the engines are destroyed.  Queued
        contigs are cancelled, the running ones finish, and the feeder — possibly parked on a look-ahead slot that a cancelled task
        will never release — is told to stop and woken."""
        stop.set()
        while True:                       # contigs no context has taken yet are cancelled; the running ones finish
            try:
                item = ctx_queue.get_nowait()
            except queue.Empty:
                break
            if item is not None:
                item[2].cancel()
                try:
                    slots.release()
                except ValueError:
                    pass
        for _ in range(n_dec + n_ctx + 4):   # (a context parked on a snapshot slot whose decode will be cancelled below)
            snap_slots.release()
        for _t in ctx_threads:
            ctx_queue.put(None)
        for t_ in ctx_threads:
            if t_.is_alive():
                t_.join()
        decode_pool.shutdown(wait=True, cancel_futures=True)
        for _ in range(len(contigs) + n_ctx + args.fetch_threads + 2):
            try:
                slots.release()
            except ValueError:           # (bounded semaphore: every slot is free again)
                break

    try:
        with ThreadPoolExecutor(max(1, args.fetch_threads)) as fetch_pool:
            # A feeder takes the look-ahead slot BEFORE it submits a contig's fetch, strictly in calling order.  (Taken inside the
            # fetch workers, a later contig could grab the last slot while the one its context needs next was still waiting for
            # one — every context consumes its contigs in order, so nothing would ever have released a slot again.)
            tasks = [None] * len(contigs)
            submitted = [threading.Event() for _ in contigs]
            feeder_err = []

            def feeder():
                try:
                    for i, c in enumerate(contigs):
                        slots.acquire()
                        if stop.is_set():
                            break
                        fut = submit_fetch(fetch_pool, c)
                        tasks[i] = Future()
                        ctx_queue.put((c, fut, tasks[i]))
                        submitted[i].set()
                    for _k in range(n_ctx):
                        ctx_queue.put(None)
                except BaseException as e:          # (e.g. the pools were shut down by a failure below)
                    feeder_err.append(e)
                finally:
                    for ev in submitted:
                        ev.set()

            feed = threading.Thread(target=feeder, name="c3r-feeder", daemon=True)
            feed.start()

            def close_fetcher():
                """The BAM handles (the mapped file, the record buffers of the largest contig each thread fetched) go as soon as
                the last fetch is through, beside the contexts' last contigs."""
                feed.join()
                fetch_pool.shutdown(wait=True)          # (every fetch has been submitted; the with-block's own shutdown is then a no-op)
                t0_ = time()
                fetcher.close()
                mark("all", "bam_close", t0_)
            closer = threading.Thread(target=close_fetcher, name="c3r-closer", daemon=True)
            closer.start()
            try:
                t0 = time()
                make_engines()                                             # (the first fetches are running)
                mark("all", "engines", t0)
                for t_ in ctx_threads:
                    t_.start()
                for i, ctg in enumerate(contigs):                          # merge in calling order as the contigs come out
                    submitted[i].wait()
                    if tasks[i] is None:
                        raise RuntimeError("contig %s was never submitted: %r" % (ctg, feeder_err[:1]))
                    rows = tasks[i].result()
                    tasks[i] = None
                    if rows is None:
                        log("[WARNING] Contig name %s provided but no mapped reads found in BAM, skip!" % ctg)
                        continue
                    if hasattr(rows, "result"):                        # decode (and merge) detached to a worker
                        rows = rows.result()
                    t0 = time()
                    if isinstance(rows, tuple) and rows[0] == "merged":
                        if rows[1] is not None:
                            merger.write_merged(rows[1])               # merged on the worker: only the ordered write is left
                    else:
                        merge_contig(ctg, rows)
                    t_merge += time() - t0
                    mark(ctg, "write" if isinstance(rows, tuple) else "merge", t0)
                    results.append((ctg, None))
            except BaseException:
                stop_workers()           # before the fetch pool's own shutdown waits for fetches nobody will consume
                raise
        t0 = time()
        for t_ in ctx_threads:
            t_.join()
        decode_pool.shutdown()
        n_sites, t_fetch, t_dev = stats["sites"], stats["fetch"], stats["dev"]
        closer.join()
        mark("all", "shutdown", t0)
        called = [c for c, _f in results]
    except Exception as e:               # with several ranks: reach the rendezvous first, then every rank fails
        work_err = e
        stop_workers()                   # (idempotent) no thread is inside libc3r any more
        if merger is not None:
            merger.discard()             # no truncated output.vcf.gz (without EOF block) beside a stale .tbi
        if world == 1:
            for t_ in engine_threads:
                t_.join()
            for e_ in engines:
                if e_ is not None:
                    e_.close()
            raise
    for t_ in engine_threads:
        t_.join()
    for e in engines:
        if e is not None:
            e.close()
    if world > 1:
        import json
        with open(os.path.join(parts_dir, "rank%d.json" % rank), "w") as f:
            json.dump(dict(counts={str(k): v for k, v in part_counts.items()}, called=called, n_sites=n_sites), f)
        _all_ranks_ok(dist, world, work_err, "calling its contigs")
        if rank != 0:
            _all_ranks_ok(dist, world, None, "assembling the output")  # leave only when rank 0 has written the result
            return 0
    fin_err = None
    try:
        if world > 1:
            called_set, n_sites = set(), 0
            for r in range(world):
                j = json.load(open(os.path.join(parts_dir, "rank%d.json" % r)))
                called_set.update(j["called"])
                n_sites += j["n_sites"]
                for k, (a, b, c) in j["counts"].items():
                    merger.n_read += a; merger.n_kept += b; merger.n_tagged += c
            called = [c for c in all_contigs if c in called_set]
            todo = []
            for k, c in enumerate(all_contigs):
                fn = os.path.join(parts_dir, "%05d.vcf" % k)
                if c in called_set and os.path.exists(fn) and os.path.getsize(fn):
                    todo.append((fn, os.path.join(parts_dir, "%05d_nt.vcf" % k)))

            def load_part(fns):
                """One contig's merged records as the ranks left them -> what the writer takes: with compressed output a piece
                compressed and indexed here, on a pool thread (the ordered step is then a file append), else the text."""
                out = []
                for fn_, want in ((fns[0], True), (fns[1], merger.out_nt is not None)):
                    if not want or not os.path.exists(fn_):
                        out.append(None)
                    elif merger.stream_gz:
                        from . import bamio as _b
                        data = np.fromfile(fn_, dtype=np.uint8)
                        out.append(_b.VcfPiece(data) if data.size else None)
                    else:
                        out.append(open(fn_).read())
                return out
            with ThreadPoolExecutor(max(1, min(8, n_thr // 4 or 1))) as part_pool:
                for main_part, nt_part in part_pool.map(load_part, todo):
                    merger._header()
                    for out_, m_ in ((merger.out, main_part), (merger.out_nt, nt_part)):
                        if out_ is None or m_ is None:
                            continue
                        if hasattr(m_, "free"):
                            out_.append(m_)
                        else:
                            out_.write(m_)
        t0 = time()
        n_read, n_kept, n_tag = merger.close(log)
        mark("all", "close_out", t0)
        # tmp/CONTIGS and tmp/CHUNK_LIST as run_clair3_rna leaves them (:436-449): contigs without reads are dropped by its
        # `samtools idxstats` check (:184-210) before they are written; here that is known once the contig has been fetched
        with open(os.path.join(out_dir, "tmp", "CONTIGS"), "w") as f:
            f.write("\n".join(called))
        with open(os.path.join(out_dir, "tmp", "CHUNK_LIST"), "w") as f:
            for c in called:
                for k in range(1, chunk_nums[c] + 1):
                    f.write("%s %d %d\n" % (c, k, chunk_nums[c]))
        t0 = time()
        if not args.no_compress and not getattr(merger, "streamed", False):      # (streamed: <out>.vcf.gz + .tbi are complete already)
            sort_vcf.compress_vcf(out_fn)
            if table is not None and n_kept:
                sort_vcf.compress_vcf(out_nt_fn)
        if table is not None:
            log("[INFO] Dataset size:%d, total variants tagged by REDIportal dataset: %d" % (len(table), n_tag))
        log("[INFO] %d contigs, %d candidate sites, %d records written to %s%s" % (len(called), n_sites, n_kept, out_fn, "" if args.no_compress else ".gz"))
        t_gz = time() - t0
        log("[INFO] set-up %.2f s, fetch %.2f s (overlapped), device stage %.2f s, merge %.2f s, bgzip+tabix %.2f s, total %.2f s"
            % (t_setup, t_fetch, t_dev, t_merge, t_gz, time() - t_all))
    except Exception as e:               # rank 0 still meets the others at the last rendezvous, and all of them fail
        fin_err = e
        if merger is not None:
            merger.discard()
        if world == 1:
            raise
    if world > 1:
        _all_ranks_ok(dist, world, fin_err, "assembling the output")
    return 0


def main(argv=None):
    args = build_parser().parse_args(argv)
    try:
        try:
            return Run(args)
        finally:
            try:                          # the 8.9-GB blocks libc3r.so keeps for a process's next context: this process has none
                from . import capi
                capi.trim()
            except Exception:
                pass
    except SystemExit:
        raise
    except Exception as e:       # fail loudly: there is no CPU fallback
        print("[ERROR] call_sample (MI355X path) failed: %s" % e, file=sys.stderr)
        return 1
    finally:
        # (the command-line run owns the process group Run() created: taken down here rather than by the interpreter's exit)
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            try:
                import torch.distributed as dist
                if dist.is_initialized():
                    dist.destroy_process_group()
            except Exception:
                pass


if __name__ == "__main__":
    sys.exit(main())
