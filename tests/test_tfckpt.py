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
