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


def test_g4_cpp_decoder_rows_exact():
    """The C++ decoder of libc3r (c3r_decode_text, host-only: no GPU needed) against the same golden G4."""
    import __graft_entry__ as g
    g.build()
    from clair3_rna_amd import capi
    g4 = json.load(open(os.path.join(G, "g4_decode.json")))
    cases = g4["cases"]
    rows = capi.decode_text("chr20", [c["pos"] for c in cases], [c["ref33"] for c in cases], [c["alt_info"] for c in cases],
                            np.asarray([c["Y"] for c in cases], dtype=np.float32))
    exp = [r for c in cases for r in c["rows"]]
    assert len(rows) == len(exp) == 700
    bad = [(a, b) for a, b in zip(rows, exp) if a != b]
    assert not bad, (len(bad), bad[:3])
    # --qual None and hidden RefCall rows
    some = cases[:50]
    r2 = capi.decode_text("chr20", [c["pos"] for c in some], [c["ref33"] for c in some], [c["alt_info"] for c in some],
                          np.asarray([c["Y"] for c in some], dtype=np.float32), qual=None, show_ref=False)
    py = decode.vcf_rows("chr20", [c["pos"] for c in some], [c["ref33"] for c in some], [c["alt_info"] for c in some],
                         np.asarray([c["Y"] for c in some], dtype=np.float32), qual_for_pass=None, show_ref=False)
    assert r2 == py and all("RefCall" not in r and "LowQual" not in r for r in r2)
