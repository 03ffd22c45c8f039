# The synthetic chr20 read set of bench.py as flat files for the stand-alone probes:  python tools/dump_reads.py /tmp/rs [contig_len]
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clair3_rna_amd import synth
base = sys.argv[1] if len(sys.argv) > 1 else "/tmp/rs"
L = int(sys.argv[2]) if len(sys.argv) > 2 else synth.CHR20_LEN
ref, rs, info = synth.generate_contig(contig_len=L, seed=synth.SEED, depth=20.0)
rs.reads.tofile(base + ".reads"); rs.cigar.tofile(base + ".cigar"); rs.seq.tofile(base + ".seq")
print(info, "max ops per read", int(rs.reads["n_cigar"].max()))
