# k_scan_tiles ablations on the chr20 pass (env C3R_SCAN_ABL bits: 1 no walk, 2 no indel events, 4 no coverage, 8 no column store,
# 16 skip intron-only tiles, 32 skip tiles with aligned bases):  python tools/scan_breakdown.py
# (the ablation / phase-clock code is compiled in only with -DC3R_SCAN_DIAG=1:  bash tools/build_variant.sh diag -DC3R_SCAN_DIAG=1, then
#  C3R_LIB=gpurun_variants/libc3r_diag.so; this script picks that library up by itself when it exists and C3R_LIB is not set)
import os, sys
if "C3R_LIB" not in os.environ:
    _d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_variants", "libc3r_diag.so")
    if os.path.exists(_d):
        os.environ["C3R_LIB"] = _d
        sys.stderr.write("using %s (built %s) — rebuild it after every change of csrc/: a stale one measures, and fails, like the code it was built from\n"
                         % (_d, __import__("time").strftime("%Y-%m-%d %H:%M", __import__("time").gmtime(os.path.getmtime(_d)))))
    else:
        sys.stderr.write("no gpurun_variants/libc3r_diag.so: C3R_SCAN_ABL / C3R_SCAN_DBG have no effect on the product build\n")
sys.path.insert(0, '.')
from clair3_rna_amd import capi, synth
import bench
ref, rs, info = synth.generate_contig(contig_len=synth.CHR20_LEN, seed=synth.SEED, depth=20.0)
chunks = bench.chunk_list(synth.CHR20_LEN)
eng = capi.Engine(0); eng.set_params(); eng.load_reads(rs); eng.set_reference(1, ref)
for abl in (0, 16, 32, 1, 2, 4, 8):
    os.environ["C3R_SCAN_ABL"] = str(abl)
    eng.begin_batch(); eng.scan_regions(chunks); eng.end_batch()
    eng.set_profiling(True); eng.reset_kernel_stats()
    for _ in range(3):
        eng.begin_batch(); n = eng.scan_regions(chunks); eng.end_batch()
    eng.set_profiling(False)
    ks = eng.kernel_stats()
    print("abl %2d  n=%6d  " % (abl, n) + "  ".join("%s %.3f" % (k.replace("k_", ""), v["total_ms"] / 3) for k, v in sorted(ks.items())))
