"""clair3_rna_amd — MI355X-native pileup variant-calling hot path for Clair3-RNA.

Host-side Python mirror of the reference's per-chunk interface
(clair3_rna/call_var_bam.py -> src/create_tensor_pileup.py | clair3_rna/call_variants.py) over the
C-ABI of libc3r.so (include/c3r.h).  The compute path is HIP only; nothing here falls back to CPU.
"""
from .reads import ReadSet, READ_DTYPE, parse_cigar, pack_seq  # noqa: F401

__version__ = "0.1.0"
