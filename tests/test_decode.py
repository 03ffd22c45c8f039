"""A8 decode (clair3_rna_amd/decode.py) against golden G4: rows the reference's batch_output printed
(clair3_rna/call_variants.py:1077-1392) for 700 seeded (probabilities, ref33, alt_info) triples."""
import json
import os

import numpy as np

from clair3_rna_amd import decode

G = os.path.join(os.path.dirname(__file__), "golden")


def test_g4_decode_rows_exact():
    g4 = json.load(open(os.path.join(G, "g4_decode.json")))
    bad = []
    for c in g4["cases"]:
        y = np.asarray(c["Y"], dtype=np.float32)
        row = decode.vcf_row("chr20", c["pos"], c["ref33"], c["alt_info"], y)
        got = [row] if row is not None else []
        if got != c["rows"]:
            bad.append((c["pos"], got, c["rows"], c["alt_info"]))
    assert not bad, (len(bad), bad[:3])
    assert len(g4["cases"]) >= 700


def test_quality_score_examples():
    assert decode.quality_score_from(np.float32(0.8) * np.float32(0.8)) == 12.5     # SURVEY.md Appendix E example
    assert decode.quality_score_from(0.0) == 0.0
    assert decode.parse_alt_info("20-XG 9 RA 11\n") == (20, {"XG": 9, "RA": 11})
    assert decode.parse_alt_info("0-\n") == (0, {})
