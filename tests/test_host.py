"""CPU-side tests: the C-ABI library loads and exports every symbol include/c3r.h declares (no compute
without a GPU), host-side logic (read packing, alt_info reconstruction, chunk arithmetic) and the
oracle's network restatement against torch.nn.LSTM (golden G5 role: TensorFlow is absent)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_capi_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from clair3_rna_amd import capi
    hdr = open(os.path.join(ROOT, "include", "c3r.h")).read()
    declared = sorted(set(re.findall(r"\b(c3r_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 20
    lib = capi.load_library()
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(capi.EXPORTS) == declared
    assert lib.c3r_version().startswith(b"c3r")
    assert lib.c3r_weight_count(18) == 2072216      # parameter count at C=18
    p = capi.default_params()
    assert (p.channels, p.min_mq, p.excl_flags, p.min_coverage, p.max_depth_rescale) == (18, 5, 2316, 4, 144)
    assert (p.snp_min_af, p.indel_min_af) == (0.08, 0.15)


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from clair3_rna_amd import capi
    with pytest.raises(capi.C3RError):
        capi.Engine(0)


def test_struct_layouts_match_header():
    from clair3_rna_amd import capi, reads
    assert reads.READ_DTYPE.itemsize == 32 and capi.SITE_DTYPE.itemsize == 52 and capi.TOKEN_DTYPE.itemsize == 16
    assert reads.READ_DTYPE.fields["seq_off"][1] == 16 and reads.READ_DTYPE.fields["flag"][1] == 24


def test_read_packing_roundtrip():
    from clair3_rna_amd.reads import ReadSet, parse_cigar
    rs = ReadSet.from_records([(5, "2S3M1I2M", "ACGTNACG", 16, 30, 2), (2, "4M", "TTTT", 0, 60, 0)])
    assert rs.reads["pos"].tolist() == [2, 5]
    assert rs.read_bases(1, 0, 8) == "ACGTNACG" and rs.read_bases(0, 1, 2) == "TT" and rs.read_bases(1, 7, 3) == "GNN"
    assert parse_cigar("10M2I5N").tolist() == [(10 << 4) | 0, (2 << 4) | 1, (5 << 4) | 3]


def test_alt_info_from_tokens_matches_oracle_columns():
    """tokens (BAM order) -> ordered alt dict must equal generate_tensor's alt_dict for the same column."""
    from clair3_rna_amd import altinfo, capi
    from clair3_rna_amd.reads import ReadSet
    from oracle import oracle as orc
    ref = "ACGTACGTACGTACGTACGTACGTACGTAC"
    recs = [(4, "3M2I3M", "ACGTTGTA", 0, 60, 0), (4, "3M1D4M", "TCGACGT", 16, 60, 0), (5, "2M2I3M", "CGTTGTA", 16, 60, 0),
            (5, "2M", "CA", 0, 60, 0), (6, "1M2D2M", "GCG", 0, 60, 0), (2, "3M4N3M", "GTAGTA", 0, 60, 0)]
    rs = ReadSet.from_records(recs)
    rows = {int(r.split("\t")[1]): r.split("\t")[4] for r in orc.mpileup(rs.reads, rs.cigar, rs.seq, "c", 1, 30)}
    pos = 7      # 1-based column; reads 0,1,2,3 cover with ins / del tokens
    # hand-build the tokens the kernel would emit for column 7 (0-based 6), in BAM order
    toks = np.zeros(6, dtype=capi.TOKEN_DTYPE)
    order = np.argsort([r[0] for r in recs], kind="stable")
    k = 0
    for ridx, oi in enumerate(order):
        p0, cig, seq, flag = recs[oi][0], recs[oi][1], recs[oi][2], recs[oi][3]
        import re as _re
        x, y, tok = p0, 0, None
        ops = [(int(n), o) for n, o in _re.findall(r"(\d+)([MIDNS])", cig)]
        for i, (n, o) in enumerate(ops):
            if o in "MDN":
                if pos - 1 < x + n and pos - 1 >= x:
                    base = {"A": 1, "C": 2, "G": 4, "T": 8}[seq[y + pos - 1 - x]] if o == "M" else (16 if o == "D" else 17)
                    indel, q = 0, 0
                    if pos - 1 == x + n - 1 and i + 1 < len(ops):
                        if ops[i + 1][1] == "I":
                            indel, q = ops[i + 1][0], y + (n if o == "M" else 0)
                        elif ops[i + 1][1] == "D" and o != "D":
                            indel = -ops[i + 1][0]
                    tok = (ridx, indel, q, base, 1 if flag & 16 else 0, 0)
                    break
                x += n
                if o == "M":
                    y += n
            else:
                y += n
        if tok:
            toks[k] = tok
            k += 1
    alt, depth = altinfo.alt_dict_from_tokens(toks[:k], rs, ref, 1, pos, depth=altinfo.COUNT_DEPTH)
    o = orc.generate_tensor(rows[pos], ref[pos - 1], pos, ref, 1)
    assert [[a, b] for a, b in alt.items()] == o["alt"] and depth == o["depth"]


def test_bench_chunk_list_matches_reference_arithmetic():
    import bench
    from oracle import oracle as orc
    L = 64444167
    ch = bench.chunk_list(L)
    assert len(ch) == 13
    for i, (a, b) in enumerate(ch):
        r = orc.chunk_region(contig_len=L, chunk_id=i + 1, chunk_num=13)
        assert (a, b) == (r["ctg_start"], r["ctg_end"])
    assert ch[-1][1] >= L


def test_oracle_network_matches_torch_lstm():
    """Keras LSTM equations as restated in the oracle == torch.nn.LSTM(bidirectional) with weight_ih=K.T,
    weight_hh=R.T, bias_ih=b, bias_hh=0, and the dense/selu/softmax heads."""
    import torch
    from clair3_rna_amd import synth
    from oracle import oracle as orc
    torch.manual_seed(0)
    for C in (18, 30):
        w = synth.random_weights(C, seed=99 + C)
        X = np.random.RandomState(C).randint(-40, 40, size=(5, 33, C)).astype(np.int32)
        probs, y1, y2 = orc.forward(w, X, return_hidden=True)
        q = [0]

        def take(*shape):
            n = int(np.prod(shape))
            a = torch.from_numpy(w[q[0]:q[0] + n].reshape(shape).copy())
            q[0] += n
            return a

        x = torch.from_numpy(X.astype(np.float32))
        for (cin, H, yref) in ((C, 128, y1), (256, 160, y2)):
            lstm = torch.nn.LSTM(cin, H, batch_first=True, bidirectional=True)
            with torch.no_grad():
                for sfx in ("", "_reverse"):
                    K, Rm, b = take(cin, 4 * H), take(H, 4 * H), take(4 * H)
                    getattr(lstm, "weight_ih_l0" + sfx).copy_(K.t())
                    getattr(lstm, "weight_hh_l0" + sfx).copy_(Rm.t())
                    getattr(lstm, "bias_ih_l0" + sfx).copy_(b)
                    getattr(lstm, "bias_hh_l0" + sfx).zero_()
                x, _ = lstm(x)
            assert np.abs(x.numpy() - yref).max() < 2e-5
        selu = torch.nn.functional.selu
        W4, b4 = take(33 * 320, 128), take(128)
        a4 = selu(x.reshape(5, -1) @ W4 + b4)
        W51, b51, W52, b52 = take(128, 128), take(128), take(128, 128), take(128)
        Wg, bg, Wz, bz = take(128, 21), take(21), take(128, 3), take(3)
        p21 = torch.softmax(selu(selu(a4 @ W51 + b51) @ Wg + bg), 1)
        p3 = torch.softmax(selu(selu(a4 @ W52 + b52) @ Wz + bz), 1)
        ref = torch.cat([p21, p3], 1).numpy()
        assert q[0] == w.size
        assert np.abs(ref - probs).max() < 1e-5


def test_c_abi_from_plain_c(tmp_path):
    """include/c3r.h and include/c3r_io.h are C (not C++) headers; a C99 program resolves the entry points and runs the
    calls that need no GPU (defaults, weight count, struct sizes, a failing open with its message)."""
    import subprocess
    exe = str(tmp_path / "abi_check")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_check.c"), "-o", exe, "-ldl"])
    out = subprocess.run([exe, os.path.join(ROOT, "clair3_rna_amd", "libc3r.so"), os.path.join(ROOT, "clair3_rna_amd", "libc3r_io.so")],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert out.stdout.startswith("ok c3r")


def test_bin_counter_layout_is_a_permutation_that_keeps_groups_and_separates_neighbours(tmp_path):
    """K0's record counters (csrc/reads_kernels.hpp, cnt_at) are laid out so that the bins of one locus do not share cache lines: the layout must be
    a permutation of every block of 1024 bins, keep a group of four bins in four consecutive words (k_bin_scan's 16-byte load), and put neighbouring
    groups at least 128 bytes apart.  The same program checks the giant spans' slice arithmetic (giant_slices / giant_slice: 1..32 slices, a partition
    of the span's records in order, also for counts near 2^31).  Host-side checks of the very functions the kernels use (hipcc, no GPU)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    exe = str(tmp_path / "layout_check")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-Wno-unused-function",
                           os.path.join(ROOT, "tests", "c", "layout_check.hip"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("cnt_at ok") and "giant slices ok" in out.stdout, out.stdout + out.stderr


def test_phase1_bytes_counts_aligned_and_covered_positions():
    """bench.phase1_bytes (phase 1): input records + C * 4 B per position with an aligned base or a deletion of a passing read;
    positions that reads only span with a ref-skip are counted apart; filtered reads count for the input bytes only."""
    import bench
    from clair3_rna_amd.reads import ReadSet
    recs = [dict(pos=100, cigar="10M5D10M100N20M", seq="A" * 40, flag=0, mapq=60),          # aligned 100..124 and 225..244, spans 100..244
            dict(pos=110, cigar="30M", seq="C" * 30, flag=16, mapq=60),                       # aligned 110..139
            dict(pos=300, cigar="50M", seq="G" * 50, flag=256, mapq=60),                      # secondary: filtered
            dict(pos=400, cigar="50M", seq="T" * 50, flag=0, mapq=3)]                         # MAPQ below 5: filtered
    rs = ReadSet.from_records(recs)
    b, aligned, covered = bench.phase1_bytes(rs, 18)
    assert aligned == (140 - 100) + 20 and covered == 245 - 100
    assert b == rs.reads.nbytes + rs.cigar.nbytes + rs.seq.nbytes + 4 * 18 * aligned
    empty = ReadSet.from_records([])
    assert bench.phase1_bytes(empty, 18)[1:] == (0, 0)


def test_synth_expression_spread_is_opt_in_and_deterministic():
    from clair3_rna_amd import synth
    a = synth.generate_contig(contig_len=400000, seed=11, depth=20.0)
    b = synth.generate_contig(contig_len=400000, seed=11, depth=20.0, expr_sigma=0.0)
    assert a[0] == b[0] and np.array_equal(a[1].reads, b[1].reads) and np.array_equal(a[1].cigar, b[1].cigar)      # the default stream is untouched
    c = synth.generate_contig(contig_len=400000, seed=11, depth=20.0, expr_sigma=2.3, max_level=12000.0)
    d = synth.generate_contig(contig_len=400000, seed=11, depth=20.0, expr_sigma=2.3, max_level=12000.0)
    assert np.array_equal(c[1].reads, d[1].reads) and c[2]["n_reads"] != a[2]["n_reads"]


def test_third_party_pin_script_harvests_the_known_answer_cases():
    """tools/pin_third_party.py re-uses the inputs of tests/test_oracle_mpileup.py (and adds the depth-cap case): the harvest needs neither tool."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("pin_third_party", os.path.join(root, "tools", "pin_third_party.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    cases = m.harvest_cases()
    assert len(cases) >= 60 and all(len(c[0]) >= 1 for c in cases)
    caps = m.cap_cases()
    assert [len(c[0]) for c in caps] == [8005, 8006] and caps[0][3] == dict(max_depth=8000)
