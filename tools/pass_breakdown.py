# Host-inclusive stage times of one chr20 pass on the GPU box (not part of bench.py's timed region):  python tools/pass_breakdown.py
import sys, time
sys.path.insert(0, '.')
import numpy as np
from clair3_rna_amd import capi, synth
ref, rs, info = synth.generate_contig(contig_len=synth.CHR20_LEN, seed=synth.SEED, depth=20.0)
eng = capi.Engine(0); eng.set_params()
w = synth.random_weights(18); eng.load_weights(w, 18)
for it in range(3):
    t0=time.perf_counter(); eng.load_reads(rs); t1=time.perf_counter(); eng.set_reference(1, ref); t2=time.perf_counter()
    size=(synth.CHR20_LEN + 12)//13
    chunks=[(a, min(a+size, synth.CHR20_LEN)) for a in range(0, synth.CHR20_LEN, size)]
    eng.begin_batch(); n=eng.scan_regions(chunks); eng.end_batch(); eng.synchronize(); t3=time.perf_counter()
    p=eng.infer(); t4=time.perf_counter()
    rows, nrows=eng.call_rows_text("chr20", qual=2, show_ref=True); t5=time.perf_counter()
    print("load_reads %.1f ms (%.1f MB)  set_reference %.1f ms (%.1f MB)  scan %.1f ms  infer+fetch %.1f ms (%d sites)  call_rows %.1f ms (%d rows)" % (
        1e3*(t1-t0), (rs.reads.nbytes+rs.cigar.nbytes+rs.seq.nbytes)/1e6, 1e3*(t2-t1), len(ref)/1e6, 1e3*(t3-t2), 1e3*(t4-t3), n, 1e3*(t5-t4), nrows))
