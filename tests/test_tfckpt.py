"""TF-free checkpoint bundle reader (clair3_rna_amd/tfckpt.py).  No TensorFlow-written file exists in the image, so this pins
the reader against a writer of the same published formats, against known CRC32C values, and pins the name/shape matching."""
import numpy as np
import pytest

from clair3_rna_amd import io, synth, tfckpt


def _keras_like(w, channels, style):
    """Split the flat blob into variables named the way object-based Keras save_weights names them (two observed styles)."""
    out, p = {}, 0

    def take(shape):
        nonlocal p
        n = int(np.prod(shape))
        a = w[p:p + n].reshape(shape)
        p += n
        return a
    for layer, cin, H in (("LSTM1", channels, 128), ("LSTM2", 256, 160)):
        for d in ("forward", "backward"):
            base = "%s/%s_layer/cell" % (layer, d) if style == 0 else "%s/%s_lstm/lstm_cell" % (layer, d)
            out[base + "/kernel/.ATTRIBUTES/VARIABLE_VALUE"] = take((cin, 4 * H))
            out[base + "/recurrent_kernel/.ATTRIBUTES/VARIABLE_VALUE"] = take((H, 4 * H))
            out[base + "/bias/.ATTRIBUTES/VARIABLE_VALUE"] = take((4 * H,))
    for layer, shp in (("L4", (10560, 128)), ("L5_1", (128, 128)), ("L5_2", (128, 128)), ("Y_gt21_logits", (128, 21)), ("Y_genotype_logits", (128, 3))):
        out[layer + "/kernel/.ATTRIBUTES/VARIABLE_VALUE"] = take(shp)
        out[layer + "/bias/.ATTRIBUTES/VARIABLE_VALUE"] = take((shp[1],))
    assert p == len(w)
    return out


def test_crc32c_known_answers():
    assert tfckpt.crc32c(b"123456789") == 0xe3069283                      # the standard CRC-32C check value
    assert tfckpt.crc32c(b"\x00" * 32) == 0x8a9136aa                       # RFC 3720 B.4
    assert tfckpt.crc32c(b"\xff" * 32) == 0x62a8ab43


@pytest.mark.parametrize("channels,style", [(18, 0), (30, 1)])
def test_bundle_roundtrip_and_layer_matching(tmp_path, channels, style):
    w = synth.random_weights(channels, seed=5 + channels)
    named = _keras_like(w, channels, style)
    named["optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE"] = np.zeros((1,), np.float32)
    named["LSTM1/forward_layer/cell/kernel/.OPTIMIZER_SLOT/optimizer/m/.ATTRIBUTES/VARIABLE_VALUE"] = np.ones((channels, 512), np.float32)
    prefix = str(tmp_path / "variables")
    tfckpt.write_bundle(prefix, named)
    back = tfckpt.read_bundle(prefix)
    assert set(back) == set(named) and all(np.array_equal(back[k], named[k]) for k in named)
    blob = tfckpt.weights_from_bundle(prefix, channels)
    assert blob.dtype == np.float32 and np.array_equal(blob, w)
    assert np.array_equal(io.load_weights(prefix, channels), w)          # the driver's loader finds the bundle
    # corruption is detected, missing tensors fail loudly
    d = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    d[100] ^= 1
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(d))
    with pytest.raises(ValueError):
        tfckpt.read_bundle(prefix)
    del named["L5_2/bias/.ATTRIBUTES/VARIABLE_VALUE"]
    tfckpt.write_bundle(prefix, named)
    with pytest.raises(ValueError):
        tfckpt.weights_from_bundle(prefix, channels)


# ---- an INDEPENDENT statement of the formats (LevelDB table_format.md, tensor_bundle.proto, RFC 3720 CRC32C): nothing below calls into
# tfckpt's own writer, so reader and writer cannot share one misreading.  Still "unverified against a TensorFlow-written file".
def _crc32c_bitwise(data):
    crc = 0xffffffff
    for b in data:
        crc ^= b
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82f63b78 if crc & 1 else 0)       # reflected Castagnoli polynomial 0x1EDC6F41
    return crc ^ 0xffffffff


def _masked(crc):
    return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xffffffff


def _vi(v):
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def test_crc32c_rfc3720_vectors_and_bitwise_twin():
    import struct
    vec = [(bytes(range(32)), 0x46dd794e), (bytes(range(31, -1, -1)), 0x113fdb5c),
           (bytes.fromhex("01c00000" "00000000" "00000000" "00000000" "14000000" "00000400" "00000014" "00000018" "28000000" "00000000" "02000000" "00000000"), 0xd9963a56)]
    for data, want in vec:                                                   # RFC 3720 B.4: incrementing, decrementing, an iSCSI read PDU
        assert tfckpt.crc32c(data) == want == _crc32c_bitwise(data)
    rng = np.random.RandomState(3)
    for n in (0, 1, 7, 64, 1000):
        d = rng.randint(0, 256, n).astype(np.uint8).tobytes()
        assert tfckpt.crc32c(d) == _crc32c_bitwise(d)
        assert tfckpt.crc32c(d[n // 2:], tfckpt.crc32c(d[:n // 2])) == tfckpt.crc32c(d)      # incremental form
    assert _masked(0) == 0xa282ead8 and struct.pack("<I", _masked(_crc32c_bitwise(b"a"))) != struct.pack("<I", _crc32c_bitwise(b"a"))


def test_hand_built_prefix_compressed_multi_block_table(tmp_path):
    """A bundle index written byte by byte from the format descriptions: sorted keys with long shared prefixes, restart interval 2
    (so most keys are stored as deltas), four data blocks, index keys that are short separators (not block keys), an empty
    metaindex block, a header entry under the empty key, masked CRC32C on every block and on every tensor."""
    import struct
    rng = np.random.RandomState(11)
    tensors = {"LSTM1/forward_layer/cell/bias/.ATTRIBUTES/VARIABLE_VALUE": rng.randn(512).astype("<f4"),
               "LSTM1/forward_layer/cell/kernel/.ATTRIBUTES/VARIABLE_VALUE": rng.randn(18, 512).astype("<f4"),
               "LSTM1/forward_layer/cell/recurrent_kernel/.ATTRIBUTES/VARIABLE_VALUE": rng.randn(128, 512).astype("<f4"),
               "L4/bias/.ATTRIBUTES/VARIABLE_VALUE": rng.randn(128).astype("<f4"),
               "L4/kernel/.ATTRIBUTES/VARIABLE_VALUE": rng.randn(33, 128).astype("<f4"),
               "_CHECKPOINTABLE_OBJECT_GRAPH": None,                                       # a DT_STRING entry: must be skipped
               "save_counter/.ATTRIBUTES/VARIABLE_VALUE": np.array([7], "<i8")}             # DT_INT64: skipped too
    data, entries = bytearray(), []
    hdr = b"\x08\x01" + b"\x10\x00" + b"\x1a\x02\x08\x01"                                  # num_shards = 1, LITTLE, version {producer 1}
    entries.append((b"", hdr))
    for key in sorted(tensors):
        a = tensors[key]
        if a is None:
            raw, dtype, shape = b"\x04graf", 7, []
        else:
            raw, dtype, shape = a.tobytes(), (1 if a.dtype == np.dtype("<f4") else 9), list(a.shape)
        shp = b"".join(b"\x12" + _vi(len(d)) + d for d in (b"\x08" + _vi(s) for s in shape))
        e = b"\x08" + _vi(dtype) + b"\x12" + _vi(len(shp)) + shp + b"\x20" + _vi(len(data)) + b"\x28" + _vi(len(raw)) + \
            b"\x35" + struct.pack("<I", _masked(_crc32c_bitwise(raw)))
        entries.append((key.encode(), e))
        data += raw
    def block(items, interval):
        out, restarts, prev = bytearray(), [], b""
        for i, (k, v) in enumerate(items):
            shared = 0
            if i % interval == 0:
                restarts.append(len(out))
            else:
                while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                    shared += 1
            out += _vi(shared) + _vi(len(k) - shared) + _vi(len(v)) + k[shared:] + v
            prev = k
        for r in restarts:
            out += struct.pack("<I", r)
        return bytes(out + struct.pack("<I", len(restarts)))
    f, handles = bytearray(), []
    def emit(b):
        off = len(f)
        f.extend(b + b"\x00" + struct.pack("<I", _masked(_crc32c_bitwise(b + b"\x00"))))
        return off, len(b)
    groups = [entries[0:2], entries[2:4], entries[4:6], entries[6:]]
    for gi, g in enumerate(groups):
        off, size = emit(block(g, 2))
        last = g[-1][0]
        nxt = groups[gi + 1][0][0] if gi + 1 < len(groups) else None
        sep = last + b"\x00" if nxt is None else last[:next(i for i in range(len(last) + 1) if i == len(last) or last[i] != nxt[i]) + 1]
        if not (last <= sep and (nxt is None or sep < nxt)):
            sep = last                                                                    # (a separator must sort in [last, next))
        handles.append((sep, _vi(off) + _vi(size)))
    mo, ms = emit(block([], 1))
    io_, is_ = emit(block(handles, 1))
    foot = _vi(mo) + _vi(ms) + _vi(io_) + _vi(is_)
    f += foot + b"\x00" * (40 - len(foot)) + struct.pack("<Q", 0xdb4775248b80fb57)
    prefix = str(tmp_path / "variables")
    open(prefix + ".index", "wb").write(bytes(f))
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
    back = tfckpt.read_bundle(prefix)
    floats = {k: v for k, v in tensors.items() if v is not None and v.dtype == np.dtype("<f4")}
    assert set(back) == set(floats)
    for k, v in floats.items():
        assert back[k].shape == v.shape and np.array_equal(back[k], v)
    # a flipped bit in ANY block (data, index) is caught
    for where in (10, io_ + 2):
        g = bytearray(f); g[where] ^= 0x10
        open(prefix + ".index", "wb").write(bytes(g))
        with pytest.raises(ValueError):
            tfckpt.read_bundle(prefix)


@pytest.mark.parametrize("block_size", [1024, 4096, 262144])
def test_bundle_laid_out_the_way_tensorflow_writes_it(tmp_path, block_size):
    """A bundle as `model.save_weights(prefix)` of an object-based Keras model leaves it (clair3_rna/call_variants.py:1472 loads exactly that):
    one shard, uncompressed table blocks flushed by size with restart points every 16 keys, keys sorted bytewise, the BundleHeaderProto under
    the empty key, and — besides the float variables — the serialized object graph (DT_STRING), `save_counter` and the optimizer's `iter`
    (DT_INT64), float optimizer slots.  The reader must skip what is not a model weight by dtype / name and still find every one of the
    2,072,216 parameters of the 18-channel model exactly once."""
    import struct
    w = synth.random_weights(18, seed=23)
    assert len(w) == 2072216
    named = _keras_like(w, 18, 0)
    named["optimizer/beta_1/.ATTRIBUTES/VARIABLE_VALUE"] = np.full((), 0.9, np.float32)
    named["optimizer/learning_rate/.ATTRIBUTES/VARIABLE_VALUE"] = np.full((), 1e-3, np.float32)
    named["L4/kernel/.OPTIMIZER_SLOT/optimizer/m/.ATTRIBUTES/VARIABLE_VALUE"] = np.zeros((10560, 128), np.float32)
    named["L4/kernel/.OPTIMIZER_SLOT/optimizer/v/.ATTRIBUTES/VARIABLE_VALUE"] = np.ones((10560, 128), np.float32)
    graph = b"\x0a\x2a\x0a\x05LSTM1" + bytes(range(64)) * 40               # (an opaque serialized TrackableObjectGraph: never parsed)
    extra = [("_CHECKPOINTABLE_OBJECT_GRAPH", 7, (), tfckpt._string_tensor_bytes([graph])),
             ("save_counter/.ATTRIBUTES/VARIABLE_VALUE", 9, (), struct.pack("<q", 3)),
             ("optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE", 9, (), struct.pack("<q", 123456))]
    prefix = str(tmp_path / "variables")
    tfckpt.write_bundle(prefix, named, block_size=block_size, extra=extra)
    idx = open(prefix + ".index", "rb").read()
    assert idx[-8:] == struct.pack("<Q", 0xdb4775248b80fb57)
    back = tfckpt.read_bundle(prefix)
    assert "_CHECKPOINTABLE_OBJECT_GRAPH" not in back and "save_counter/.ATTRIBUTES/VARIABLE_VALUE" not in back      # skipped by dtype
    assert set(back) == set(named) and all(np.array_equal(back[k], named[k]) for k in named)
    blob = tfckpt.weights_from_bundle(prefix, 18)
    assert blob.size == 2072216 and np.array_equal(blob, w)
    assert np.array_equal(io.load_weights(prefix, 18), w)
    # the 36 entries of this model fit one block at LevelDB's default 4 KB and at TensorFlow's 256 KB; at 1 KB the table is multi-block
    from clair3_rna_amd.tfckpt import _varint
    foot = idx[-48:]
    _mo, p = _varint(foot, 0); _ms, p = _varint(foot, p); io_, p = _varint(foot, p); is_, p = _varint(foot, p)
    n_blocks = len(tfckpt._read_block(idx, io_, is_))
    assert (n_blocks > 1) == (block_size == 1024)
    assert b"LSTM2/backward_layer/cell/recurrent_kernel" not in idx or idx.count(b"LSTM2/backward_layer/cell/") <= 2       # prefix compression at work
