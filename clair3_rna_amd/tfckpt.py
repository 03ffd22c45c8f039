"""TensorFlow checkpoint bundle (`<prefix>.index` + `<prefix>.data-00000-of-00001`) -> the flat weight blob of include/c3r.h,
without TensorFlow.  SURVEY §8(f) row F5; replaces `m.load_weights(prefix)` (clair3_rna/call_variants.py:1472).

UNVERIFIED AGAINST A REAL CHECKPOINT: neither TensorFlow nor any Clair3-RNA model is in the build image.  The reader follows
the published formats — LevelDB table (data blocks with prefix-compressed keys and restart arrays, 1-byte compression type +
masked CRC32C trailer, index block, 48-byte footer with magic 0xdb4775248b80fb57) and tensor_bundle.proto (BundleHeaderProto
under key "", BundleEntryProto per variable) — and is exercised by a writer of the same formats (tests/test_tfckpt.py).
Variables are matched to layers by NAME FRAGMENT + SHAPE, not by exact key, because the object-graph key strings
(`LSTM1/forward_layer/cell/kernel/.ATTRIBUTES/VARIABLE_VALUE` ...) depend on the TF/Keras version; every expected tensor must
be found exactly once or the load fails loudly.  When TensorFlow is available, INTEGRATION.md §3's three-line conversion
is the verified route.
"""
import os
import struct

import numpy as np

_MAGIC = 0xdb4775248b80fb57
_DT_FLOAT = 1

# ---- CRC32C (Castagnoli), table driven; TF stores masked values
_T = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
    _T.append(_c)


def crc32c(data, crc=0):
    crc ^= 0xffffffff
    for b in data:
        crc = _T[(crc ^ b) & 0xff] ^ (crc >> 8)
    return crc ^ 0xffffffff


def _mask(crc):
    return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xffffffff


def _varint(buf, p):
    v = s = 0
    while True:
        b = buf[p]
        p += 1
        v |= (b & 0x7f) << s
        if not b & 0x80:
            return v, p
        s += 7


def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _read_block(data, off, size, verify=True):
    body, ctype = data[off:off + size], data[off + size]
    if verify:
        want = struct.unpack_from("<I", data, off + size + 1)[0]
        if _mask(crc32c(data[off:off + size + 1])) != want:
            raise ValueError("checkpoint index: block checksum mismatch at offset %d" % off)
    if ctype != 0:
        raise ValueError("checkpoint index: compressed table blocks (type %d) are not supported" % ctype)
    n_restarts = struct.unpack_from("<I", body, len(body) - 4)[0]
    end = len(body) - 4 - 4 * n_restarts
    p, key, out = 0, b"", []
    while p < end:
        shared, p = _varint(body, p)
        non_shared, p = _varint(body, p)
        vlen, p = _varint(body, p)
        key = key[:shared] + body[p:p + non_shared]
        p += non_shared
        out.append((key, body[p:p + vlen]))
        p += vlen
    return out


def _proto_fields(buf):
    p, out = 0, []
    while p < len(buf):
        tag, p = _varint(buf, p)
        f, wt = tag >> 3, tag & 7
        if wt == 0:
            v, p = _varint(buf, p)
        elif wt == 1:
            v = buf[p:p + 8]; p += 8
        elif wt == 2:
            n, p = _varint(buf, p)
            v = buf[p:p + n]; p += n
        elif wt == 5:
            v = buf[p:p + 4]; p += 4
        else:
            raise ValueError("protobuf wire type %d" % wt)
        out.append((f, wt, v))
    return out


def _entry(buf):
    e = dict(dtype=0, shape=[], shard=0, offset=0, size=0, crc=None, sliced=False)
    for f, wt, v in _proto_fields(buf):
        if f == 1: e["dtype"] = v
        elif f == 2:
            for f2, _w2, v2 in _proto_fields(v):
                if f2 == 2:
                    d = 0
                    for f3, _w3, v3 in _proto_fields(v2):
                        if f3 == 1: d = v3
                    e["shape"].append(d)
        elif f == 3: e["shard"] = v
        elif f == 4: e["offset"] = v
        elif f == 5: e["size"] = v
        elif f == 6: e["crc"] = struct.unpack("<I", v)[0]
        elif f == 7: e["sliced"] = True
    return e


def read_bundle(prefix, verify=True):
    """{variable key: float32 ndarray} for every DT_FLOAT variable of the bundle."""
    idx = open(prefix + ".index", "rb").read()
    if len(idx) < 48 or struct.unpack_from("<Q", idx, len(idx) - 8)[0] != _MAGIC:
        raise ValueError("%s.index is not a TensorFlow checkpoint index (bad table magic)" % prefix)
    foot = idx[-48:]
    _mo, p = _varint(foot, 0); _ms, p = _varint(foot, p)
    io_, p = _varint(foot, p); is_, p = _varint(foot, p)
    entries = []
    for _sep, handle in _read_block(idx, io_, is_, verify):
        bo, q = _varint(handle, 0); bs, q = _varint(handle, q)
        entries += _read_block(idx, bo, bs, verify)
    header = dict(num_shards=1, endianness=0)
    out, shards = {}, {}
    for key, val in entries:
        if key == b"":
            for f, _wt, v in _proto_fields(val):
                if f == 1: header["num_shards"] = v
                elif f == 2: header["endianness"] = v
            if header["endianness"] != 0:
                raise ValueError("big-endian checkpoint bundles are not supported")
            continue
        e = _entry(val)
        if e["dtype"] != _DT_FLOAT or e["sliced"]:
            continue
        if e["shard"] not in shards:
            fn = "%s.data-%05d-of-%05d" % (prefix, e["shard"], header["num_shards"])
            shards[e["shard"]] = np.memmap(fn, dtype=np.uint8, mode="r")
        raw = shards[e["shard"]][e["offset"]:e["offset"] + e["size"]]
        n = int(np.prod(e["shape"])) if e["shape"] else 1
        if len(raw) != e["size"] or e["size"] != 4 * n:
            raise ValueError("variable %r: %d bytes for shape %s" % (key, e["size"], e["shape"]))
        if verify and e["crc"] is not None and _mask(crc32c(bytes(raw))) != e["crc"]:
            raise ValueError("variable %r: data checksum mismatch" % key)
        out[key.decode()] = np.frombuffer(bytes(raw), dtype="<f4").reshape(e["shape"]).copy()
    return out


def _expected(channels):
    """(layer fragment, direction fragment or None, weight fragment, shape) in blob order (include/c3r.h, Keras order)."""
    spec = []
    for layer, cin, H in (("LSTM1", channels, 128), ("LSTM2", 256, 160)):
        for d in ("forward", "backward"):
            spec += [(layer, d, "recurrent_kernel", (H, 4 * H)), (layer, d, "kernel", (cin, 4 * H)), (layer, d, "bias", (4 * H,))]
    for layer, shp in (("L4", (33 * 320, 128)), ("L5_1", (128, 128)), ("L5_2", (128, 128)), ("Y_gt21_logits", (128, 21)), ("Y_genotype_logits", (128, 3))):
        spec += [(layer, None, "kernel", shp), (layer, None, "bias", (shp[1],))]
    return spec


def weights_from_bundle(prefix, channels=18, verify=True):
    """Flat fp32 blob in the layout c3r_load_weights takes.  Raises unless every tensor is found exactly once."""
    tensors = {k: v for k, v in read_bundle(prefix, verify).items() if "OPTIMIZER_SLOT" not in k and "optimizer" not in k.lower()}
    got = {}
    for layer, d, wname, shape in _expected(channels):
        hits = []
        for k, v in tensors.items():
            parts = k.split("/")
            if tuple(v.shape) != shape or layer not in parts:                 # the layer's attribute name is a path component
                continue
            if d is not None and not any(d in p for p in parts):               # forward_layer / forward_lstm / ...
                continue
            if wname not in parts:                                             # exact component: 'kernel' is not 'recurrent_kernel'
                continue
            hits.append(k)
        if len(hits) != 1:
            raise ValueError("checkpoint %s: expected exactly one variable for %s/%s/%s %s, found %s" % (prefix, layer, d or "-", wname, shape, hits))
        got[(layer, d, wname)] = tensors[hits[0]]
    blob = []
    for layer, cin, H in (("LSTM1", channels, 128), ("LSTM2", 256, 160)):
        for d in ("forward", "backward"):
            blob += [got[(layer, d, "kernel")], got[(layer, d, "recurrent_kernel")], got[(layer, d, "bias")]]
    for layer in ("L4", "L5_1", "L5_2", "Y_gt21_logits", "Y_genotype_logits"):
        blob += [got[(layer, None, "kernel")], got[(layer, None, "bias")]]
    return np.concatenate([b.reshape(-1) for b in blob]).astype(np.float32)


# ------------------------------------------------------------------------------------------------ writer (tests, demos)
def _block(entries, restart_interval=16):
    body, restarts, last = bytearray(), [], b""
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(body))
        else:
            while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                shared += 1
        body += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        body += struct.pack("<I", r)
    body += struct.pack("<I", len(restarts))
    return bytes(body)


def _string_tensor_bytes(strings):
    """tensor_bundle's layout of a DT_STRING tensor in the data file: the elements' lengths as varint64s, the masked CRC32C of those length
    bytes (4 bytes, little-endian), then the strings back to back."""
    lens = b"".join(_put_varint(len(x)) for x in strings)
    return lens + struct.pack("<I", _mask(crc32c(lens))) + b"".join(strings)


def write_bundle(prefix, tensors, per_block=7, block_size=None, extra=()):
    """Write {key: float32 array} as a one-shard bundle in the same formats read_bundle parses.
    block_size: flush a data block once its entries reach that many bytes (LevelDB's TableBuilder rule; restart points every 16 keys) instead
    of every `per_block` keys — 4096 is LevelDB's default block size, 262144 the one TensorFlow's table options carry.
    extra: (key, dtype enum, shape, raw bytes) entries of other types, as a Keras object-based checkpoint holds them: the serialized object
    graph under `_CHECKPOINTABLE_OBJECT_GRAPH` (DT_STRING = 7, scalar), `save_counter` and optimizer `iter` (DT_INT64 = 9)."""
    items = [(k.encode(), _DT_FLOAT, tuple(np.shape(tensors[k])), np.ascontiguousarray(tensors[k], dtype="<f4").tobytes()) for k in tensors]
    items += [(k.encode() if isinstance(k, str) else k, dt, tuple(shp), bytes(raw)) for k, dt, shp, raw in extra]
    items.sort(key=lambda it: it[0])                       # the table's keys are sorted bytewise ('_' sorts behind the upper-case layer names)
    data, ents = bytearray(), []
    for k, dt, shp, raw in items:
        shape = b"".join(b"\x12" + _put_varint(len(_put_varint(d)) + 1) + b"\x08" + _put_varint(d) for d in shp)
        e = b"\x08" + _put_varint(dt) + b"\x12" + _put_varint(len(shape)) + shape + (b"\x20" + _put_varint(len(data)) if len(data) else b"") + \
            b"\x28" + _put_varint(len(raw)) + b"\x35" + struct.pack("<I", _mask(crc32c(raw)))
        ents.append((k, e))
        data += raw
    header = b"\x08\x01" + b"\x1a\x02\x08\x01"          # num_shards = 1, (endianness LITTLE = 0: the proto3 default is not written,) version { producer: 1 }
    ents = [(b"", header)] + ents
    groups, cur, cur_bytes = [], [], 0
    for k, v in ents:
        cur.append((k, v)); cur_bytes += len(k) + len(v) + 3
        if (block_size is None and len(cur) >= per_block) or (block_size is not None and cur_bytes >= block_size):
            groups.append(cur); cur, cur_bytes = [], 0
    if cur:
        groups.append(cur)
    out, index = bytearray(), []
    for g in groups:
        blk = _block(g)
        off = len(out)
        out += blk + b"\x00" + struct.pack("<I", _mask(crc32c(blk + b"\x00")))
        index.append((g[-1][0], _put_varint(off) + _put_varint(len(blk))))
    meta = _block([])
    mo = len(out)
    out += meta + b"\x00" + struct.pack("<I", _mask(crc32c(meta + b"\x00")))
    ib = _block(index, restart_interval=1)
    io_ = len(out)
    out += ib + b"\x00" + struct.pack("<I", _mask(crc32c(ib + b"\x00")))
    foot = _put_varint(mo) + _put_varint(len(meta)) + _put_varint(io_) + _put_varint(len(ib))
    out += foot + b"\x00" * (40 - len(foot)) + struct.pack("<Q", _MAGIC)
    open(prefix + ".index", "wb").write(bytes(out))
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
