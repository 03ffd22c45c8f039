"""Test launcher: clair3_rna_amd.call_sample with a stand-in for capi.Engine, so that the driver's orchestration — planning, fetch
threads, per-context threads, merge order, seam duplicates, the multi-rank hand-off under torch.distributed (gloo) — runs on a
machine without a GPU.  The stand-in "calls" one deterministic record at every 211th covered position of each region (and at
region ends, so that neighbouring chunks emit the same position twice, with different content).  Not a product path.

    python tests/support/fake_engine_sample.py <call_sample arguments>
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from clair3_rna_amd import call_sample, capi


class FakeEngine(object):
    def __init__(self, device=0, stream=None):
        self.params = None
        self.n_candidates = 0
        self._rows = []

    def close(self): pass
    def synchronize(self): pass
    def load_weights(self, blob, channels=None): self.w = float(np.asarray(blob).sum())
    def set_precision(self, mode): self.mode = mode
    def precision(self): return getattr(self, "mode", "f16x3"), -1.0
    def set_bed(self, which, iv): pass
    def set_params(self, **kw): self.kw = kw
    def set_sites(self, sites): self.sites = list(sites)
    def set_reference(self, start, seq, upper_view=False):
        self.ref = seq if isinstance(seq, bytes) else (seq.encode() if isinstance(seq, str) else bytes(seq))
        if os.environ.get("C3R_FAKE_FAIL_LEN") == str(len(self.ref)):       # test hook: this contig's device stage fails
            raise RuntimeError("stand-in engine: injected failure on the contig of length %d" % len(self.ref))

    def load_reads(self, rs):
        self.pos = rs.reads["pos"].astype(np.int64)
        self.cov = np.zeros(len(self.ref) + 2 if hasattr(self, "ref") else 1, dtype=bool)
        self.rs = rs

    def _covered(self):
        cov = np.zeros(len(self.ref) + 2, dtype=bool)
        for p, l in zip(self.pos, self.rs.reads["l_seq"].astype(np.int64)):
            cov[p + 1:min(len(cov), p + 1 + l)] = True
        return cov

    def begin_batch(self): self._rows = []
    def end_batch(self): pass

    def scan_regions(self, regions):
        cov = self._covered()
        for k, (a, b) in enumerate(regions):
            for p in range(max(1, a), min(b, len(self.ref)) + 1):
                if (p % 211 == 0 and cov[p]) or p == a or p == b:
                    base = chr(self.ref[p - 1]).upper()
                    alt = "ACGT"[(p + k) % 4]
                    q = (p * 7 + k) % 31
                    self._rows.append("%s\t%d\t.\t%s\t%s\t%d.00\t%s\t.\tGT:GQ\t0/1:%d\n" % ("{ctg}", p, base, alt if alt != base else ".", q,
                                                                                     "PASS" if q >= 2 else "LowQual", k))
        self.n_candidates = len(self._rows)
        return self.n_candidates

    def scan(self, a, b):
        self._rows = []
        return self.scan_regions([(a, b)])

    def infer(self, fetch=False): pass

    def call_rows_text(self, ctg, qual=2, show_ref=True):
        rows = [r.replace("{ctg}", ctg) for r in self._rows if show_ref or "\t.\t" not in r.split("\t", 4)[4][:2] and r.split("\t")[4] != "."]
        return "".join(rows).encode(), len(rows)


class FakeSnapshot(object):
    """Stand-in for capi.RowSnapshot: the rows of one contig, detached from the engine (decoded later, on a worker thread)."""

    def __init__(self, rows):
        self.rows = rows

    def decode(self, ctg, qual=2, show_ref=True, as_array=False):
        eng = FakeEngine()
        eng._rows = self.rows
        text, n = eng.call_rows_text(ctg, qual, show_ref)
        return (np.frombuffer(text, dtype=np.uint8) if as_array else text), n

    def free(self): pass


if os.environ.get("C3R_FAKE_SNAPSHOTS") == "1":          # the driver's detached path: c3r_rows_begin -> decode pool -> worker-side merge
    FakeEngine.rows_begin = lambda self, drop_ref_calls=False, host_reads=True: FakeSnapshot(list(self._rows))
    FakeEngine.reserve = lambda self, n: None
capi.Engine = FakeEngine
if __name__ == "__main__":
    sys.exit(call_sample.main())
