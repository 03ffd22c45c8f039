"""Merge the per-chunk VCFs into the final call set: SURVEY §8(f) row F4, the `sort_vcf` step of the reference
(src/sort_vcf.py:85-292; invoked by run_clair3_rna:711-726 and :838-853 with the flag set accepted below).

Behaviour kept, including the parts that are accidents of the reference's implementation, because downstream files
must be identical (fixture tests/golden/g6_sortvcf.json.gz was produced by the reference's own function):
  * chunk files are chosen by name: prefix, suffix, and the contig name being a SUBSTRING of the file name; a file is
    read until its first record of another contig (that is how `chr1` skips the `chr11` files);
  * duplicate positions (adjacent chunks overlap by 33 bp) are resolved last-file-wins in directory-listing order;
  * header lines are collected as a set-in-order from the files read so far and written once, at the end of the first
    contig for which any exist;
  * RefCall rows are dropped unless --show_ref; a variant row is re-labelled LowQual when QUAL <= --qual;
  * REDIportal tagging sets FILTER=RNAEditing when (contig, pos, ref, alt) is in the table and the row mentions neither
    "Germline" nor "RefCall"; the untagged copy is the same text with RNAEditing -> PASS;
  * no input rows at all, or no row kept: the output file is left EMPTY (not even a header).
--compress_vcf writes <out>.gz as BGZF and <out>.gz.tbi with this package's own writers (bgzip / tabix are not dependencies).
"""
import gzip
import os
import sys
from argparse import ArgumentParser

MAJOR_CONTIGS = ["chr%s" % c for c in list(range(1, 23)) + ["X", "Y"]] + [str(c) for c in list(range(1, 23)) + ["X", "Y"]]


def _str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "ture", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "flase", "false", "f", "n", "0"):
        return False
    raise ValueError("Boolean value expected.")


def _contig_order(names, extra):
    order = MAJOR_CONTIGS + list(extra)
    return sorted(names, key=order.index)


def load_rediportal(path, contigs, filter_tags=None):
    """REDIportal table (gzip or plain TSV: contig, pos, ref, edited base, strand, db, ...) -> {(contig, pos): (ref, alt, db)}."""
    table = {}
    opener = gzip.open if open(path, "rb").read(2) == b"\x1f\x8b" else open
    with opener(path, "rt") as f:
        for i, line in enumerate(f):
            if i == 0:
                continue
            c = line.rstrip().split("\t", 6)
            if len(c) < 6 or (contigs and c[0] not in contigs):
                continue
            try:
                key = (c[0], int(c[1]))
            except ValueError:
                continue
            if filter_tags is not None and c[5] not in filter_tags:
                continue
            table[key] = (c[2], c[3], c[5])
    return table


def _relabel(row, label):
    f = row.split("\t")
    f[6] = label
    return "\t".join(f)


def _filter_row(row, c, contig, qual, show_ref, rediportal):
    """One record of a per-chunk VCF -> (pos, row as written, tagged 0/1), or None when sort_vcf drops it
    (src/sort_vcf.py:204-236: RefCall rows unless --show_ref, LowQual relabel at QUAL <= --qual, REDIportal tagging)."""
    pos, q, ref, alt = int(c[1]), float(c[5]), c[3], c[4]
    is_ref = alt == "." or ref == alt
    if is_ref and not show_ref:
        return None
    if not is_ref and qual and q <= qual:
        row = _relabel(row, "LowQual")
    tagged = 0
    hit = rediportal.get((contig, pos)) if rediportal is not None else None
    if hit is not None and "Germline" not in row and "RefCall" not in row:
        f9 = row.split("\t", 8)
        if f9[3] == hit[0] and f9[4] == hit[1]:
            f9[6] = "RNAEditing"
            row = "\t".join(f9)
            tagged = 1
    return pos, row, tagged


def merge_chunk_vcfs(input_dir, output_fn, contigs, prefix="pileup", suffix=".vcf", qual=2, show_ref=False,
                     rediportal=None, output_no_tagging_fn=None, listing=None, log=print):
    """Returns (rows_read, rows_written, rows_tagged).  `listing` pins the directory order (tests); default os.listdir."""
    names = list(listing) if listing is not None else os.listdir(input_dir)
    if prefix is not None:
        names = [n for n in names if n.startswith(prefix)]
    if suffix is not None:
        names = [n for n in names if n.endswith(suffix)]
    if not names:
        open(output_fn, "w").close()
        log("[WARNING] No vcf file found with prefix/suffix %s/%s*%s, output empty vcf file" % (input_dir, prefix, suffix))
        return 0, 0, 0
    tagging = rediportal is not None
    out = open(output_fn, "w")
    out_nt = open(output_no_tagging_fn, "w") if tagging and output_no_tagging_fn else None
    header, header_done = [], False
    n_read = n_kept = n_tagged = 0
    for contig in _contig_order(contigs, contigs):
        by_pos = {}
        for name in (n for n in names if contig in n):
            with open(os.path.join(input_dir, name)) as f:
                for row in f:
                    n_read += 1
                    if row[0] == "#":
                        if row not in header:
                            header.append(row)
                        continue
                    c = row.strip().split(None, 6)
                    if c[0] != contig:
                        break                      # a file of another contig whose name merely contains this one
                    kept = _filter_row(row, c, contig, qual, show_ref, rediportal)
                    if kept is None:
                        continue
                    by_pos[kept[0]] = kept[1]
                    n_tagged += kept[2]
                    n_kept += 1
        if not header_done and header:
            out.write("".join(header))
            if out_nt:
                out_nt.write("".join(header))
            header_done = True
        for pos in sorted(by_pos):
            out.write(by_pos[pos])
            if out_nt:
                out_nt.write(by_pos[pos].replace("RNAEditing", "PASS"))
    out.close()
    if out_nt:
        out_nt.close()
    if n_read == 0 or n_kept == 0:
        open(output_fn, "w").close()
        log("[WARNING] No %s found, output empty vcf file" % ("vcf file" if n_read == 0 else "variant"))
    return n_read, n_kept, n_tagged


class SampleMerger(object):
    """In-process form of the same merge for the whole-sample driver (call_sample.py): contigs are added in output order,
    each as the newline-terminated rows its chunks produced (any order, duplicates at chunk seams allowed); the files written
    are byte-identical to merge_chunk_vcfs over the per-chunk files."""

    def __init__(self, output_fn, header_text, qual=2, show_ref=False, rediportal=None, output_no_tagging_fn=None, native=True,
                 stream_gz=False):
        """stream_gz: write <output_fn>.gz + .tbi directly, block by block as the contigs arrive (bamio.VcfGzWriter: the bytes
        compress_vcf would make of the finished file), instead of the plain file that compress_vcf turns into them afterwards."""
        self.output_fn, self.header = output_fn, header_text
        self.native, self._edits = native, None
        import threading
        self._edits_lock = threading.Lock()
        self.qual, self.show_ref, self.rediportal = qual, show_ref, rediportal
        self.stream_gz = bool(stream_gz) and native is not False
        self.out_nt_fn = output_no_tagging_fn if rediportal is not None else None
        if self.stream_gz:
            from . import bamio
            self.out = bamio.VcfGzWriter(output_fn + ".gz")
            self.out_nt = bamio.VcfGzWriter(self.out_nt_fn + ".gz") if self.out_nt_fn else None
        else:
            self.out = open(output_fn, "w")
            self.out_nt = open(self.out_nt_fn, "w") if self.out_nt_fn else None
        self.n_read = self.n_kept = self.n_tagged = 0
        self.header_done = False

    def discard(self):
        """Failure path: nothing half-written stays behind — the streamed .gz (no EOF block yet) and its writers go, and so does a
        .tbi / plain file of an EARLIER run in the same directory, which would otherwise sit beside a truncated result."""
        import os
        for w, fn in ((self.out, self.output_fn), (self.out_nt, self.out_nt_fn)):
            if w is None or fn is None:
                continue
            try:
                if self.stream_gz:
                    w.discard()
                else:
                    w.close()
            except Exception:
                pass
            for stale in (fn, fn + ".gz", fn + ".gz.tbi"):
                try:
                    os.remove(stale)
                except OSError:
                    pass

    def add_contig(self, contig, rows):
        """rows: bytes (or str) of newline-terminated records of `contig`.  The work is done by c3r_vcf_merge
        (csrc/vcfio.cpp); add_contig_py below is the same in Python and is what the golden tests compare it with."""
        if self.native is False:
            return self.add_contig_py(contig, rows)
        from . import bamio
        blob = rows.encode() if isinstance(rows, str) else (rows if hasattr(rows, "ctypes") else bytes(rows))     # (uint8 array: by pointer)
        if not len(blob):
            return
        edits = None
        if self.rediportal:
            if self._edits is None:                       # one pass over the table: per-contig entry lists
                self._edits = {}
                for (c, pos), hit in self.rediportal.items():
                    self._edits.setdefault(c, []).append((pos, hit[0], hit[1]))
            edits = self._edits.get(contig)
        self.write_merged(bamio.vcf_merge(blob, self.qual, self.show_ref, edits, self.out_nt is not None))

    def merge_only(self, contig, rows):
        """The per-contig work of add_contig without touching the output (c3r_vcf_merge releases the GIL): callable from worker
        threads, several contigs at a time; hand the result to write_merged in output order."""
        if self.native is False:
            raise RuntimeError("merge_only needs the native merge")
        from . import bamio
        blob = rows.encode() if isinstance(rows, str) else (rows if hasattr(rows, "ctypes") else bytes(rows))     # (uint8 array: by pointer)
        if not len(blob):
            return None
        edits = None
        if self.rediportal:
            with self._edits_lock:
                if self._edits is None:
                    self._edits = {}
                    for (c, pos), hit in self.rediportal.items():
                        self._edits.setdefault(c, []).append((pos, hit[0], hit[1]))
            edits = self._edits.get(contig)
        merged, merged_nt, counts = bamio.vcf_merge(blob, self.qual, self.show_ref, edits, self.out_nt is not None)
        if self.stream_gz:
            # compressed and indexed here, on the worker (bamio.VcfPiece): write_merged only appends the finished blocks
            merged = bamio.VcfPiece(merged) if merged is not None and len(merged) else None
            merged_nt = bamio.VcfPiece(merged_nt) if merged_nt is not None and len(merged_nt) else None
        return merged, merged_nt, counts

    def write_merged(self, res):
        if res is None:
            return
        merged, merged_nt, (n_read, n_kept, n_tag) = res
        self._header()
        self.n_read += n_read
        self.n_kept += n_kept
        self.n_tagged += n_tag
        for out, m in ((self.out, merged), (self.out_nt, merged_nt)):
            if out is None or m is None:
                continue
            if hasattr(m, "free"):
                out.append(m)
            elif len(m):
                out.write(m if self.stream_gz else (m.tobytes() if hasattr(m, "tobytes") else m).decode())

    def _header(self):
        if not self.header_done:
            self.out.write(self.header)
            if self.out_nt:
                self.out_nt.write(self.header)
            self.header_done = True

    def add_contig_py(self, contig, rows):
        text = rows.decode() if isinstance(rows, (bytes, bytearray)) else rows
        if not text:
            return
        self._header()
        by_pos = {}
        for row in text.splitlines(True):
            self.n_read += 1
            kept = _filter_row(row, row.strip().split(None, 6), contig, self.qual, self.show_ref, self.rediportal)
            if kept is None:
                continue
            by_pos[kept[0]] = kept[1]
            self.n_tagged += kept[2]
            self.n_kept += 1
        for pos in sorted(by_pos):
            self.out.write(by_pos[pos])
            if self.out_nt:
                self.out_nt.write(by_pos[pos].replace("RNAEditing", "PASS"))

    def close(self, log=print):
        empty = self.n_read == 0 or self.n_kept == 0
        if self.stream_gz:
            # (an empty result is an EMPTY file, header dropped: written the plain way and left to compress_vcf like before)
            (self.out.discard if empty else self.out.close)()
            if self.out_nt:
                (self.out_nt.discard if (empty or not self.n_kept) else self.out_nt.close)()
            self.streamed = not empty
            if empty and self.out_nt_fn:
                open(self.out_nt_fn, "w").close()
        else:
            self.out.close()
            if self.out_nt:
                self.out_nt.close()
            self.streamed = False
        if empty:
            open(self.output_fn, "w").close()
            log("[WARNING] No %s found, output empty vcf file" % ("vcf file" if self.n_read == 0 else "variant"))
        return self.n_read, self.n_kept, self.n_tagged


def merge_stream(lines, output_fn):
    """stdin mode (src/sort_vcf.py:85-121): concatenated VCF text -> sorted VCF."""
    header, per = [], {}
    for row in lines:
        if row[0] == "#":
            if row not in header:
                header.append(row)
            continue
        c = row.strip().split(None, 3)
        per.setdefault(c[0], {})[int(c[1])] = row
    with open(output_fn, "w") as out:
        out.write("".join(header))
        for contig in _contig_order(list(per), list(per)):
            for pos in sorted(per[contig]):
                out.write(per[contig][pos])


def _reg2bin(beg, end):
    """UCSC binning scheme (SAM spec 5.3), 0-based half-open."""
    end -= 1
    if beg >> 14 == end >> 14: return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17: return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20: return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23: return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26: return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def compress_vcf(path, threads=0):
    """`bgzip -f` + `tabix -f -p vcf` (src/sort_vcf.py:70-75) through c3r_vcf_compress (csrc/vcfio.cpp: blocks deflated on
    threads); compress_vcf_py is the single-threaded Python twin whose bytes it must reproduce."""
    from . import bamio
    return bamio.vcf_compress(path, threads)


def compress_vcf_py(path):
    """`bgzip -f` + `tabix -f -p vcf` equivalent (src/sort_vcf.py:70-75): <path> -> <path>.gz (BGZF) + <path>.gz.tbi, original
    removed.  The index follows the tabix format description (TBI v1: VCF preset, UCSC bins, 16 kb linear index); tabix
    itself is not in the image, so it is checked against the format and by reading records back through it
    (tests/test_sort_vcf.py), not against tabix' own output."""
    import struct
    import zlib
    from .bam import _BGZF_EOF
    data = open(path, "rb").read()
    BLK = 0xff00
    coffs, out = [], bytearray()
    for i in range(0, len(data), BLK):
        chunk = data[i:i + BLK]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        cdata = co.compress(chunk) + co.flush()
        coffs.append(len(out))
        out += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(cdata) + 25)
        out += cdata + struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk))
    end_coff = len(out)
    out += _BGZF_EOF
    with open(path + ".gz", "wb") as f:
        f.write(out)

    def voff(u):            # virtual offset of uncompressed byte u (a position at a block's end belongs to the next block)
        if u >= len(data):
            return end_coff << 16
        return (coffs[u // BLK] << 16) | (u % BLK)

    names, idx = [], {}     # contig -> (bins {bin: [[beg, end], ...]}, linear [])
    u = 0
    for line in data.split(b"\n"):
        n = len(line) + 1
        if line and not line.startswith(b"#"):
            f = line.split(b"\t", 5)
            ctg, beg = f[0].decode(), int(f[1]) - 1
            end = beg + max(1, len(f[3]))
            if ctg not in idx:
                names.append(ctg)
                idx[ctg] = ({}, [])
            bins, lin = idx[ctg]
            v0, v1 = voff(u), voff(u + n)
            ch = bins.setdefault(_reg2bin(beg, end), [])
            if ch and ch[-1][1] == v0:
                ch[-1][1] = v1
            else:
                ch.append([v0, v1])
            w1 = (end - 1) >> 14
            if len(lin) <= w1:
                lin.extend([0] * (w1 + 1 - len(lin)))
            for w in range(beg >> 14, w1 + 1):
                if lin[w] == 0:
                    lin[w] = v0
        u += n
    nm = b"".join(x.encode() + b"\x00" for x in names)
    tbi = bytearray(b"TBI\x01" + struct.pack("<8i", len(names), 2, 1, 2, 0, ord("#"), 0, len(nm)) + nm)
    for ctg in names:
        bins, lin = idx[ctg]
        for w in range(1, len(lin)):
            if lin[w] == 0:
                lin[w] = lin[w - 1]
        tbi += struct.pack("<i", len(bins))
        for b_, chunks in sorted(bins.items()):
            tbi += struct.pack("<Ii", b_, len(chunks))
            for c0, c1 in chunks:
                tbi += struct.pack("<QQ", c0, c1)
        tbi += struct.pack("<i", len(lin)) + b"".join(struct.pack("<Q", v) for v in lin)
    from .bam import _bgzf_write
    with open(path + ".gz.tbi", "wb") as f:
        _bgzf_write(f, bytes(tbi))
        f.write(_BGZF_EOF)
    os.remove(path)
    return path + ".gz"


def build_parser():
    p = ArgumentParser(description="Sort and merge per-chunk VCF files by contig and position (drop-in for `sort_vcf`)")
    a = p.add_argument
    a("--output_fn", type=str, required=True)
    a("--input_dir", type=str, default=None)
    a("--vcf_fn_prefix", type=str, default=None)
    a("--vcf_fn_suffix", type=str, default=".vcf")
    a("--ref_fn", type=str, default=None)
    a("--sample_name", type=str, default="SAMPLE")
    a("--contigs_fn", type=str, default=None)
    a("--compress_vcf", type=_str2bool, default=False)
    a("--show_ref", type=_str2bool, default=False)
    a("--cmd_fn", type=str, default=None)
    a("--qual", type=int, default=2)
    a("--output_no_tagging_fn", type=str, default=None)
    a("--tag_variant_using_readiportal", type=_str2bool, default=None)
    a("--readiportal_source_fn", type=str, default=None)
    a("--readiportal_database_filter_tag", type=str, default=None)
    return p


def main(argv=None, listing=None):
    """`listing`: file names of --input_dir in the order to read them (tests; default os.listdir order like the reference)."""
    args = build_parser().parse_args(argv)
    if args.input_dir is None:
        merge_stream(sys.stdin, args.output_fn)
        return 0
    print("[INFO] Sorting VCFs...")
    if not os.path.exists(args.input_dir):
        sys.exit("[ERROR] Input directory: %s not exists!" % args.input_dir)
    if not (args.contigs_fn and os.path.exists(args.contigs_fn)):
        sys.exit("[ERROR] Cannot find contig file %s. Exit!" % args.contigs_fn)
    contigs = [l.rstrip() for l in open(args.contigs_fn)]
    table = None
    if args.tag_variant_using_readiportal:
        src = args.readiportal_source_fn
        if src is None or src.upper() == "NONE" or not os.path.exists(src):
            print("[WARNING] Enabled tagging variant using readiportal, but --readiportal_source_fn %s file not found, skip tagging!" % src)
            table = {}
        else:
            tags = set(args.readiportal_database_filter_tag.split(":")) if args.readiportal_database_filter_tag is not None else None
            table = load_rediportal(src, contigs, tags)
    n_read, n_kept, n_tag = merge_chunk_vcfs(args.input_dir, args.output_fn, contigs, args.vcf_fn_prefix, args.vcf_fn_suffix, args.qual,
                                             args.show_ref, table, args.output_no_tagging_fn, listing=listing)
    if args.compress_vcf:
        compress_vcf(args.output_fn)
        if table is not None and args.output_no_tagging_fn and n_kept:
            compress_vcf(args.output_no_tagging_fn)
    if table is not None:
        print("[INFO] Dataset size:%d, total variants tagged by REDIportal dataset: %d" % (len(table), n_tag))
    print("[INFO] Finished VCF sorting!")
    return 0


if __name__ == "__main__":
    sys.exit(main())
