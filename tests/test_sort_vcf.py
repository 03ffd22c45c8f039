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


def test_bgzf_and_tabix_index_read_back(tmp_path):
    """compress_vcf: the .tbi (TBI v1, VCF preset) must lead a reader from a region to exactly the records overlapping it."""
    import random
    rng = random.Random(5)
    hdr = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS\n"
    recs = []
    for ctg, n in (("chr1", 4000), ("chr2", 2500), ("chrM", 3)):
        pos = sorted(rng.sample(range(1, 3000000), n))
        for p in pos:
            ref = "ACGT"[p % 4] + "".join(rng.choice("ACGT") for _ in range(rng.choice([0, 0, 0, 1, 5, 40])))
            recs.append((ctg, p, ref, "%s\t%d\t.\t%s\tG\t%.2f\tPASS\t.\tGT:GQ\t0/1:%d\n" % (ctg, p, ref, rng.uniform(0, 40), rng.randint(0, 40))))
    path = str(tmp_path / "m.vcf")
    open(path, "w").write(hdr + "".join(r[3] for r in recs))
    gz = sort_vcf.compress_vcf(path)
    assert not os.path.exists(path) and gzip.open(gz, "rt").read() == hdr + "".join(r[3] for r in recs)
    _check_tabix_read_back(gz, recs, ["chr1", "chr2", "chrM"], rng)


def _check_tabix_read_back(gz, recs, names_want, rng, n_queries=40):
    """Parse <gz>.tbi (TBI v1, VCF preset) and follow it, BGZF block by block, from random regions to exactly the records overlapping them."""
    import struct
    import zlib
    raw = open(gz, "rb").read()
    tbi = gzip.open(gz + ".tbi", "rb").read()
    assert tbi[:4] == b"TBI\x01"
    n_ref, fmt, cs, cb, ce, meta, skip, l_nm = struct.unpack_from("<8i", tbi, 4)
    assert (fmt, cs, cb, ce, meta, skip) == (2, 1, 2, 0, ord("#"), 0)
    names = tbi[36:36 + l_nm].split(b"\x00")[:-1]
    assert [n.decode() for n in names] == names_want
    o = 36 + l_nm
    index = {}
    for nm in names:
        n_bin = struct.unpack_from("<i", tbi, o)[0]; o += 4
        bins = {}
        for _ in range(n_bin):
            b, n_chunk = struct.unpack_from("<Ii", tbi, o); o += 8
            bins[b] = [struct.unpack_from("<QQ", tbi, o + 16 * k) for k in range(n_chunk)]; o += 16 * n_chunk
        n_intv = struct.unpack_from("<i", tbi, o)[0]; o += 4
        lin = list(struct.unpack_from("<%dQ" % n_intv, tbi, o)); o += 8 * n_intv
        index[nm.decode()] = (bins, lin)
    assert o == len(tbi)

    def block(coff):
        bs = struct.unpack_from("<H", raw, coff + 16)[0] + 1
        return zlib.decompress(raw[coff + 18:coff + bs - 8], -15), coff + bs

    def read_from(v0, v1):
        coff, u = v0 >> 16, v0 & 0xffff
        out = b""
        while (coff << 16 | u) < v1:
            data, nxt = block(coff)
            stop = (v1 & 0xffff) if (v1 >> 16) == coff else len(data)
            out += data[u:stop]
            coff, u = nxt, 0
        return out

    def bins_of(beg, end):
        end -= 1
        out = [0]
        for sh, off in ((26, 1), (23, 9), (20, 73), (17, 585), (14, 4681)):
            out += list(range(off + (beg >> sh), off + (end >> sh) + 1))
        return out

    for _ in range(n_queries):
        ctg = rng.choice(names_want)
        beg = rng.randint(0, 3000000); end = beg + rng.choice([1, 100, 20000, 700000])
        bins, lin = index[ctg]
        lo = lin[min(beg >> 14, len(lin) - 1)] if lin else 0
        got = set()
        for b in bins_of(beg, end):
            for c0, c1 in bins.get(b, []):
                if c1 <= lo:
                    continue
                for line in read_from(max(c0, lo), c1).decode().split("\n"):
                    if line:
                        f = line.split("\t")
                        p0 = int(f[1]) - 1
                        if f[0] == ctg and p0 < end and p0 + len(f[3]) > beg:
                            got.add(line + "\n")
        want = set(r[3] for r in recs if r[0] == ctg and r[1] - 1 < end and r[1] - 1 + len(r[2]) > beg)
        assert got == want, (ctg, beg, end, len(got), len(want))


def _random_vcf(rng, n_scale=1):
    hdr = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS\n"
    recs = []
    for ctg, n in (("chr1", 4000 * n_scale), ("chr2", 2500 * n_scale), ("chrM", 3)):
        for p in sorted(rng.sample(range(1, 3000000), n)):
            ref = "ACGT"[p % 4] + "".join(rng.choice("ACGT") for _ in range(rng.choice([0, 0, 0, 1, 5, 40])))
            recs.append("%s\t%d\t.\t%s\t%s\t%.2f\tPASS\t.\tGT:GQ\t0/1:%d\n" % (ctg, p, ref, rng.choice("ACGT"), rng.random() * 40, rng.randint(0, 30)))
    return hdr, recs


@pytest.mark.parametrize("case", ["big", "empty", "header_only", "one_block_exact"])
def test_native_bgzip_tabix_reproduces_the_python_writer(case, tmp_path):
    """c3r_vcf_compress (threads) must write the very bytes compress_vcf_py writes: same blocks, same deflate stream, same .tbi."""
    import random
    rng = random.Random(11)
    hdr, recs = _random_vcf(rng)
    text = {"big": hdr + "".join(recs), "empty": "", "header_only": hdr,
            "one_block_exact": (hdr + "".join(recs))[:0xff00 * 2]}[case]
    if case == "one_block_exact":
        text = text[:text.rfind("\n") + 1]
        text += "chr2\t2999999\t.\t" + "A" * (0xff00 * 2 - len(text) - 60) + "\tC\t9.00\tPASS\t.\tGT\t0/1\n"
    a, b = str(tmp_path / "a.vcf"), str(tmp_path / "b.vcf")
    open(a, "w").write(text); open(b, "w").write(text)
    sort_vcf.compress_vcf_py(a)
    sort_vcf.compress_vcf(b, threads=5)
    assert not os.path.exists(a) and not os.path.exists(b)
    assert open(a + ".gz", "rb").read() == open(b + ".gz", "rb").read()
    assert open(a + ".gz.tbi", "rb").read() == open(b + ".gz.tbi", "rb").read()
    assert gzip.open(b + ".gz", "rt").read() == text


def test_native_merge_equals_python_merge_on_random_records(tmp_path):
    """c3r_vcf_merge against SampleMerger.add_contig_py: duplicates at seams, RefCall rows, LowQual boundary (QUAL == qual),
    tagging incl. rows that must not be tagged (RefCall / REF-ALT mismatch), every qual / show_ref combination."""
    import random
    rng = random.Random(3)
    rows = []
    for p in sorted(rng.sample(range(1, 200000), 3000)):
        ref = rng.choice("ACGT") + "".join(rng.choice("ACGT") for _ in range(rng.choice([0, 0, 0, 2])))
        kind = rng.random()
        if kind < 0.3:
            rows.append("chr7\t%d\t.\t%s\t.\t%.2f\tRefCall\t.\tGT:GQ:DP:AD:AF\t0/0:3:20:18:0.9000\n" % (p, ref[0], rng.choice([0.0, 2.0, 7.13])))
        else:
            q = rng.choice([0.0, 1.99, 2.0, 2.01, 8.0, 22.47])
            rows.append("chr7\t%d\t.\t%s\t%s\t%.2f\t%s\t.\tGT:GQ:DP:AD:AF\t0/1:%d:20:9,9:0.4500\n" % (p, ref, rng.choice(["A", "C", "G,T", "GA"]), q, "PASS" if q >= 2 else "LowQual", int(q)))
    dup = rng.sample(rows, 150)                                   # seam duplicates: the same positions again, later in the list, other content
    rows_in = rows[:2000] + [r.replace("\t0/1:", "\t1/1:").replace("\t0/0:", "\t0/0:9") for r in dup] + rows[2000:]
    blob = "".join(rows_in).encode()
    var = [r.split("\t") for r in rows if "RefCall" not in r]
    table = {("chr7", int(v[1])): (v[3], v[4], "A") for v in var[::7]}
    table.update({("chr7", int(v[1])): (v[3], "N", "A") for v in var[3::70]})          # wrong ALT: not tagged
    table.update({("chr8", int(v[1])): (v[3], v[4], "A") for v in var[1::9]})          # other contig
    hdr = "##h\n"
    for qual in (0, 2, 8):
        for show_ref in (False, True):
            for tab in (None, table, {}):
                outs = []
                for native in (True, False):
                    o, o_nt = str(tmp_path / ("m%d.vcf" % native)), str(tmp_path / ("m%d_nt.vcf" % native))
                    m = sort_vcf.SampleMerger(o, hdr, qual, show_ref, tab, o_nt, native=native)
                    m.add_contig("chr7", blob)
                    m.add_contig("chr9", b"")
                    counts = m.close(log=lambda *_: None)
                    outs.append((open(o).read(), open(o_nt).read() if tab is not None else None, counts))
                assert outs[0] == outs[1], (qual, show_ref, tab is not None)
    assert outs[0][0].count("\n") > 1500


def test_streaming_compressor_equals_whole_file_compressor(tmp_path):
    """bamio.VcfGzWriter fed piece by piece (pieces of every size, block boundaries inside lines, inside pieces, on piece ends) writes the
    .gz and .tbi bytes that compress_vcf makes of the concatenated file — and SampleMerger(stream_gz=True) the files of the plain
    merger + compress_vcf."""
    import random
    from clair3_rna_amd import bamio
    rng = random.Random(5)
    header = "##fileformat=VCFv4.2\n##contig=<ID=chr1,length=900000>\n##contig=<ID=chr2,length=900000>\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS\n"
    pieces = [header]
    for ctg in ("chr1", "chr2", "chrUn_x"):
        pos, rows = 0, []
        for _ in range(rng.choice([1, 3000, 9000])):
            pos += rng.randint(1, 120)
            ref = rng.choice(["A", "C", "G", "T", "ACGTACGTAC" * rng.randint(1, 40)])
            rows.append("%s\t%d\t.\t%s\t%s\t%.2f\tPASS\t.\tGT:GQ:DP:AD:AF\t0/1:%d:%d:%d,%d:%.4f\n"
                        % (ctg, pos, ref, rng.choice("ACGT"), rng.random() * 40, rng.randint(0, 40), rng.randint(4, 90), rng.randint(0, 40), rng.randint(0, 40), rng.random()))
        # one contig = several pieces of different sizes
        k = 0
        while k < len(rows):
            step = rng.choice([1, 2, 50, 700, 5000])
            pieces.append("".join(rows[k:k + step]))
            k += step
    text = "".join(pieces)
    assert len(text) > 5 * 0xff00
    plain = tmp_path / "whole.vcf"
    plain.write_text(text)
    sort_vcf.compress_vcf(str(plain))
    z = bamio.VcfGzWriter(str(tmp_path / "stream.vcf.gz"), threads=3)
    for p_ in pieces:
        z.write(p_)
    z.close()
    assert (tmp_path / "stream.vcf.gz").read_bytes() == (tmp_path / "whole.vcf.gz").read_bytes()
    assert (tmp_path / "stream.vcf.gz.tbi").read_bytes() == (tmp_path / "whole.vcf.gz.tbi").read_bytes()
    # a piece without its newline is refused; discard() leaves nothing behind
    z = bamio.VcfGzWriter(str(tmp_path / "bad.vcf.gz"))
    with pytest.raises(IOError):
        z.write("chr1\t5\t.\tA\tC")
    z.discard()
    assert not (tmp_path / "bad.vcf.gz").exists()
    # SampleMerger in streaming mode == plain merger + compress_vcf (records and an empty result)
    rows1 = "".join(r for r in pieces[1:] if r.startswith("chr1"))
    for label, rows in (("full", rows1), ("empty", "")):
        a, b = tmp_path / ("m_%s_a.vcf" % label), tmp_path / ("m_%s_b.vcf" % label)
        m1 = sort_vcf.SampleMerger(str(a), header, 2, False, None, None)
        m2 = sort_vcf.SampleMerger(str(b), header, 2, False, None, None, stream_gz=True)
        for m in (m1, m2):
            m.add_contig("chr1", rows)
            m.close(log=lambda _m: None)
        sort_vcf.compress_vcf(str(a))
        if not m2.streamed:
            sort_vcf.compress_vcf(str(b))
        assert (tmp_path / ("m_%s_a.vcf.gz" % label)).read_bytes() == (tmp_path / ("m_%s_b.vcf.gz" % label)).read_bytes()
        assert (tmp_path / ("m_%s_a.vcf.gz.tbi" % label)).read_bytes() == (tmp_path / ("m_%s_b.vcf.gz.tbi" % label)).read_bytes()


def test_pieces_compressed_on_their_own_give_the_same_text_and_a_valid_index(tmp_path):
    """bamio.VcfPiece + VcfGzWriter.append (a contig's records compressed and indexed on a worker, appended in order): the file
    inflates to the concatenated text, the .tbi leads from regions to exactly their records — across pieces mixed with plain
    write() calls, a piece holding two contigs, an empty piece, pieces ending on a block boundary — and SampleMerger's
    merge_only / write_merged (the whole-sample driver's path) writes the text of add_contig."""
    import random
    from clair3_rna_amd import bamio
    rng = random.Random(11)
    hdr = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS\n"
    recs, by_ctg = [], {}
    for ctg, n in (("chr1", 6000), ("chr2", 2500), ("chr3", 1), ("chr4", 3000), ("chr5", 1800)):
        for p in sorted(rng.sample(range(1, 3000000), n)):
            ref = "ACGT"[p % 4] + "".join(rng.choice("ACGT") for _ in range(rng.choice([0, 0, 0, 1, 5, 40])))
            recs.append((ctg, p, ref, "%s\t%d\t.\t%s\tG\t%.2f\tPASS\t.\tGT:GQ\t0/1:%d\n" % (ctg, p, ref, rng.uniform(3, 40), rng.randint(0, 40))))
            by_ctg.setdefault(ctg, []).append(recs[-1][3])
    # chr5: padded so that its text is a whole number of blocks (its last block is full, the next piece starts on the boundary)
    t5 = "".join(by_ctg["chr5"])
    gz = str(tmp_path / "p.vcf.gz")
    z = bamio.VcfGzWriter(gz, threads=2)
    z.write(hdr)
    c1 = by_ctg["chr1"]
    z.append(bamio.VcfPiece("".join(c1[:2500]), threads=3))           # one contig over two pieces and a plain write
    z.write("".join(c1[2500:2600]))
    z.append(bamio.VcfPiece("".join(c1[2600:])))
    z.append(bamio.VcfPiece(""))
    z.append(bamio.VcfPiece("".join(by_ctg["chr2"]) + "".join(by_ctg["chr3"])))     # two contigs in one piece
    z.write("".join(by_ctg["chr4"][:7]))
    z.write("".join(by_ctg["chr4"][7:]))
    z.append(bamio.VcfPiece(t5))
    z.close()
    with pytest.raises(IOError):
        bamio.VcfPiece("chr1\t5\t.\tA\tC")
    assert gzip.open(gz, "rt").read() == hdr + "".join(r[3] for r in recs)
    _check_tabix_read_back(gz, recs, ["chr1", "chr2", "chr3", "chr4", "chr5"], rng, n_queries=120)
    # a piece whose length is a whole number of blocks, followed by another piece
    line = "chr9\t%d\t.\tA\tG\t30.00\tPASS\t.\tGT\t0/1\n"
    rows, tot, k = [], 0, 1
    while tot + 3 * len(line % k) < 2 * 0xff00:
        rows.append(("chr9", k, "A", line % k)); tot += len(line % k); k += 1
    pre = "chr9\t%d\t.\tA\tG\t30.00\tPASS\t" % k
    rows.append(("chr9", k, "A", pre + "x" * (2 * 0xff00 - tot - len(pre) - 1) + "\n")); tot += len(rows[-1][3]); k += 1
    assert tot == 2 * 0xff00
    more = [("chr9", k + j, "A", line % (k + j)) for j in range(50)]
    gz2 = str(tmp_path / "q.vcf.gz")
    z = bamio.VcfGzWriter(gz2)
    z.append(bamio.VcfPiece("".join(r[3] for r in rows)))
    z.append(bamio.VcfPiece("".join(r[3] for r in more)))
    z.close()
    assert gzip.open(gz2, "rt").read() == "".join(r[3] for r in rows + more)
    _check_tabix_read_back(gz2, rows + more, ["chr9"], rng, n_queries=30)
    # the driver's path: merge_only on workers, write_merged in order == add_contig
    header = hdr
    outs = []
    kept = [r for r in recs if r[0] in ("chr1", "chr2", "chr4") and r[2] != "G"]       # (REF == ALT is a reference call: dropped by the merge)
    for mode in ("pieces", "plain"):
        fn = str(tmp_path / ("m_%s.vcf" % mode))
        m = sort_vcf.SampleMerger(fn, header, 2, False, None, None, stream_gz=True)
        for ctg in ("chr1", "chr2", "chr4"):
            rows_ = "".join(by_ctg[ctg])
            if mode == "pieces":
                res = m.merge_only(ctg, rows_)
                assert hasattr(res[0], "free")
                m.write_merged(res)
            else:
                m.add_contig(ctg, rows_)
        m.close(log=lambda _m: None)
        assert m.streamed
        outs.append(gzip.open(fn + ".gz", "rt").read())
        _check_tabix_read_back(fn + ".gz", kept, ["chr1", "chr2", "chr4"], rng, n_queries=40)
    assert outs[0] == outs[1] == header + "".join(r[3] for r in kept)
