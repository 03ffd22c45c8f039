"""Host-side reconstruction of the per-candidate alt_info field and of the create_tensor text line.

The kernel emits, per candidate, what every covering read shows at the centre column in BAM order
(c3r_token_t).  From that the ordered alternative-allele dictionary of the reference is rebuilt:
    src/create_tensor_pileup.py:179     Counter(base_list)  -> first-seen order of distinct tokens
    src/create_tensor_pileup.py:221-261 alt_dict keys  X<base> / I<ref><SEQ> / D<refseq> / R<ref>
    src/create_tensor_pileup.py:595-605 line = ctg \t pos \t ref33 \t ints \t "<depth>-<k v k v ...>"
Order matters: clair3_rna/call_variants.py:144,151,187,196 break ties with max(dict, key=dict.get)
(first maximum wins).
"""
from collections import OrderedDict

from .reads import NT16

_ACGT = {1: "A", 2: "C", 4: "G", 8: "T"}


def evc_base(b):
    """evc_base_from (src/create_tensor_pileup.py:64-74) for an upper-case reference base."""
    return b if b in "ACGT" else "A"


def pad_table(padins):
    """{(read_idx, qpos): (total, pad_mask)} of a PADINS_DTYPE array (capi.Engine.pad_insertions), or None."""
    if padins is None or len(padins) == 0:
        return None
    return {(int(e["read_idx"]), int(e["qpos"])): (int(e["total"]), int(e["pad_mask"])) for e in padins}


def inserted_text(readset, read_idx, qpos, n_bases, rev, pads=None):
    """The inserted string as the reference's alt_dict keys hold it (upper case).  mpileup_compat = 1: samtools >= 1.11 prints the pads of
    the run of I ops between the bases, '*' on the forward and '#' on the reverse strand (--reverse-del); `pads` = pad_table()."""
    seq = readset.read_bases(read_idx, qpos, n_bases)
    e = pads.get((read_idx, qpos)) if pads else None
    if e is None:
        return seq
    total, mask = e
    out, j = [], 0
    for ch in range(total):
        if (mask >> ch) & 1:
            out.append("#" if rev else "*")
        else:
            out.append(seq[j])
            j += 1
    return "".join(out)


COUNT_DEPTH = "count"


def alt_dict_from_tokens(tokens, readset, ref_seq, ref_start, pos, pads=None, *, depth):
    """tokens: TOKEN_DTYPE slice for one site (BAM order). ref_seq[0] is 1-based ref_start.
    depth (required): the site record's depth — the engine writes tokens only for reads that show something other than the reference base
    or a ref-skip (since round 5), so the column's depth cannot be counted from them.  COUNT_DEPTH asks for the legacy stream by name:
    one token per covering read, the depth counted here.
    Returns (OrderedDict alt, depth)."""
    if depth is None:
        raise TypeError("alt_dict_from_tokens: depth is required (the site record's depth, or altinfo.COUNT_DEPTH for a token per covering read)")
    site_depth = None if depth == COUNT_DEPTH else depth
    ref_base = evc_base(ref_seq[pos - ref_start])
    alt = OrderedDict()
    depth = alt_count = ins_count = del_count = 0
    for tk in tokens:
        b = int(tk["base"])
        if b in _ACGT:
            depth += 1
            u = _ACGT[b]
            if u != ref_base:
                alt["X" + u] = alt.get("X" + u, 0) + 1
                alt_count += 1
        elif b == 16:          # '*' / '#'
            depth += 1
            del_count += 1
        ind = int(tk["indel"])
        if ind > 0:
            seq = inserted_text(readset, int(tk["read_idx"]), int(tk["qpos"]), ind, bool(tk["rev"]), pads)
            k = "I" + ref_base + seq
            alt[k] = alt.get(k, 0) + 1
            ins_count += 1
            da = int(tk["del_after"])
            if da:                 # mpileup_compat = 1: `+<ins>-<del>`, the deletion is the read's next token on this column
                a = pos - ref_start + 1
                k = "D" + ref_seq[a:a + da]
                alt[k] = alt.get(k, 0) + 1
                del_count += 1
        elif ind < 0:
            a = pos - ref_start + 1
            k = "D" + ref_seq[a:a + (-ind)]
            alt[k] = alt.get(k, 0) + 1
            del_count += 1
    if site_depth is not None:
        depth = int(site_depth)
    ref_count = max(0, depth - del_count - ins_count - alt_count)
    if ref_count > 0:
        alt["R" + ref_base] = alt.get("R" + ref_base, 0) + ref_count
    return alt, depth


def alt_info_string(depth, alt):
    return "%d-%s" % (depth, " ".join("%s %d" % kv for kv in alt.items()))


def format_lines(ctg, sites, tensors_raw, tokens, readset, ref_seq, ref_start, padins=None):
    """Reproduce the reference's create_tensor stdout lines (debug / parity only).  padins: capi.Engine.pad_insertions() when the scan ran
    with mpileup_compat = 1 on reads whose CIGARs hold pads."""
    out = []
    pads = pad_table(padins)
    for i, s in enumerate(sites):
        pos = int(s["pos"])
        tk = tokens[int(s["tok_off"]):int(s["tok_off"]) + int(s["n_tok"])]
        alt, _ = alt_dict_from_tokens(tk, readset, ref_seq, ref_start, pos, pads, depth=int(s["depth"]))
        ints = " ".join(str(v) for v in tensors_raw[i].reshape(-1).tolist())
        out.append("%s\t%d\t%s\t%s\t%s" % (ctg, pos, s["ref33"].decode(), ints, alt_info_string(int(s["depth"]), alt)))
    return out
