"""Numerical probe (CPU, float64 emulation): how far do the probabilities move when the LSTM / dense GEMMs are evaluated
with cheaper operand splits than the f16x3 scheme?  Every scheme is a list of (weight part, activation part) product terms,
each part quantised the way the matrix pipe would see it; the terms are summed in float64 (the fp32 accumulation noise is the
same for every scheme and is measured separately by the GPU tests).

  f16x3   w_hi*x_hi + w_hi*x_lo + w_lo*x_hi                     (product path, 3 f16 MFMAs)
  f16x2a  w_hi*x_hi + w_lo*x_hi                                 (activations rounded to f16)
  f16x2w  w_hi*x_hi + w_hi*x_lo                                 (weights rounded to f16)
  f16x1   w_hi*x_hi
  f16+2f8 w_hi*x_hi + q8(w)*q8(x_lo) + q8(w_lo)*q8(x)           (corrections on the MX fp8 pipe, 2x the f16 rate)
  f16+2f6 the same with e2m3 fp6 and power-of-two scales per 32-k block (4x the f16 rate)
  f16+2f8k the fp8 scheme exactly as a kernel would run it: e4m3 corrections with FIXED activation scales (h in (-1,1): x * 2^6,
          x_lo * 2^18), one power-of-two weight scale per (gate row, block of 32 k), layer 1's integer inputs on two f16 terms

usage: python tools/precision_probe.py [n_sites] [weight_gain]
"""
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clair3_rna_amd import synth  # noqa: E402

torch.set_num_threads(8)
D = torch.float64


def f16(v):
    return v.to(torch.float16).to(D)


def q8(v, scale):
    return (v * scale).clamp(-448, 448).to(torch.float8_e4m3fn).to(D) / scale


_E2M3 = torch.tensor(sorted(set([i * 0.125 for i in range(8)] + [1 + i * 0.125 for i in range(8)] + [2 + i * 0.25 for i in range(8)] +
                                [4 + i * 0.5 for i in range(8)])), dtype=D)


def q6_block(v, axis):
    """e2m3 with one power-of-two scale per block of 32 along `axis` (the MX block layout)."""
    v = v.movedim(axis, -1)
    K = v.shape[-1]
    pad = (-K) % 32
    if pad:
        v = torch.nn.functional.pad(v, (0, pad))
    b = v.reshape(*v.shape[:-1], -1, 32)
    m = b.abs().amax(-1, keepdim=True).clamp_min(1e-300)
    sc = torch.exp2(torch.ceil(torch.log2(m / 7.5)))
    a = (b / sc).abs().clamp(max=7.5)
    idx = torch.bucketize(a, _E2M3)
    idx = idx.clamp(1, len(_E2M3) - 1)
    lo, hi = _E2M3[idx - 1], _E2M3[idx]
    r = torch.where((a - lo) <= (hi - a), lo, hi)
    out = (torch.sign(b) * r * sc).reshape(*v.shape[:-1], -1)[..., :K]
    return out.movedim(-1, axis)


def pow2_scale(v, top):
    m = float(v.abs().max())
    return 2.0 ** np.floor(np.log2(top / max(m, 1e-30)))


class Scheme:
    def __init__(self, name):
        self.name = name

    def prep_w(self, W):          # W [K, N]
        wh = f16(W)
        wl = f16(W - wh)
        d = dict(wh=wh, wl=wl, w=W)
        if self.name == "f16+2f8":
            d["w8"] = q8(W, pow2_scale(W, 256.0))
            d["wl8"] = q8(W - wh, pow2_scale(W - wh, 256.0))
        if self.name == "f16+2f6":
            d["w8"] = q6_block(W, 0)
            d["wl8"] = q6_block(W - wh, 0)
        if self.name.startswith("f16+2f8k"):
            def q8_block(v):          # e4m3, one power-of-two scale per (column, block of 32 rows): block max lands in [128, 256)
                K = v.shape[0]; pad = (-K) % 32
                b = torch.nn.functional.pad(v, (0, 0, 0, pad)).reshape(-1, 32, v.shape[1])
                m = b.abs().amax(1, keepdim=True).clamp_min(1e-300)
                sc = torch.exp2(torch.floor(torch.log2(256.0 / m)))
                return ((b * sc).clamp(-448, 448).to(torch.float8_e4m3fn).to(D) / sc).reshape(-1, v.shape[1])[:K]
            d["w8"] = q8_block(W)
            d["wl8"] = q8_block(W - wh)
        if self.name.startswith("f16x3wl8"):
            # f16x3 with the weights' lo halves stored as 8-bit fixed point: one power-of-two scale per output column (= gate row of the
            # transposed GEMM) "r", or per (column, block of 16 k) "b"
            lo = W - wh
            if self.name.endswith("r"):
                m = lo.abs().amax(0, keepdim=True).clamp_min(1e-30)
                E = torch.ceil(torch.log2(m / 127.0))
                d["wl"] = torch.round(lo / torch.exp2(E)).clamp(-128, 127) * torch.exp2(E)
            else:
                K = lo.shape[0]; pad = (-K) % 16
                lp = torch.nn.functional.pad(lo, (0, 0, 0, pad)).reshape(-1, 16, lo.shape[1])
                m = lp.abs().amax(1, keepdim=True).clamp_min(1e-30)
                E = torch.ceil(torch.log2(m / 127.0))
                d["wl"] = (torch.round(lp / torch.exp2(E)).clamp(-128, 127) * torch.exp2(E)).reshape(-1, lo.shape[1])[:K]
        if self.name.startswith("f16+2i8"):
            # int8 corrections: w8 = rint(w * 127 / 2^E), wlo8 = rint(w_lo * 127 * 2^12 / 2^E); E per layer ("g"), per 32-row tile of
            # the permuted gate rows ("t": stand-in = per column block of 8 units) or per row ("r")
            mode = self.name[-1]
            if mode == "g":
                E = torch.ceil(torch.log2(W.abs().max())).expand(1, W.shape[1])
            elif mode == "r":
                E = torch.ceil(torch.log2(W.abs().amax(0, keepdim=True).clamp_min(1e-30)))
            else:
                H = W.shape[1] // 4 if W.shape[1] % 4 == 0 and W.shape[1] > 128 else None
                m = W.abs().amax(0)
                if H and H % 8 == 0:
                    mm = m.reshape(4, H // 8, 8).amax((0, 2))            # tile = 8 units x 4 gates
                    m = mm[None, :, None].expand(4, H // 8, 8).reshape(-1)
                else:
                    mm = m.reshape(-1, 32).amax(1)
                    m = mm[:, None].expand(-1, 32).reshape(-1)
                E = torch.ceil(torch.log2(m.clamp_min(1e-30)))[None, :]
            sw = 127.0 / torch.exp2(E)
            d["w8"] = torch.round(W * sw).clamp(-127, 127) / sw
            d["wl8"] = torch.round((W - wh) * sw * 4096.0).clamp(-127, 127) / (sw * 4096.0)
        return d

    def mm(self, x, Wd, nx=0, nf=None):    # x [B, K] -> [B, N]; the first nx inputs are integers (layer 1): exact in f16, never int8
        n = self.name
        if n == "f16+2f8k_l2":     # layer 1 on f16x3, layer 2 and L4 with fp8 corrections (what k_lstm2_mx alone does)
            if nx:
                xh = f16(x); xl = f16(x - xh)
                return xh @ Wd["wh"] + xl @ Wd["wh"] + xh @ Wd["wl"]
            xh = f16(x)
            return xh @ Wd["wh"] + q8(x - xh, 2.0 ** 18) @ Wd["w8"] + q8(x, 2.0 ** 6) @ Wd["wl8"]
        if n in ("f16x3_xa", "f16x3_xw"):
            # two product terms instead of three on layer 2's input projection only (nf columns: y1, which is no part of a recurrence):
            # "_xa" = layer 1's output stored as ONE f16 plane (drops w_hi * x_lo), "_xw" = its weights rounded to f16 (drops w_lo * x_hi)
            xh = f16(x); xl = f16(x - xh)
            if nf is None:
                return xh @ Wd["wh"] + xl @ Wd["wh"] + xh @ Wd["wl"]
            full = xh[:, nf:] @ Wd["wh"][nf:] + xl[:, nf:] @ Wd["wh"][nf:] + xh[:, nf:] @ Wd["wl"][nf:]
            if n == "f16x3_xa":
                return full + xh[:, :nf] @ Wd["wh"][:nf] + xh[:, :nf] @ Wd["wl"][:nf]
            return full + xh[:, :nf] @ Wd["wh"][:nf] + xl[:, :nf] @ Wd["wh"][:nf]
        if n in ("f16+2f8k_x", "f16+2f8k_x4"):
            # fp8 corrections only where the operand is NOT part of a recurrence: layer 2's input projection (nf columns), and with
            # "_x4" the L4 dense layer; everything else f16x3
            xh = f16(x); xl = f16(x - xh)
            if nf is None:
                return xh @ Wd["wh"] + xl @ Wd["wh"] + xh @ Wd["wl"]
            xf, xr = x[:, :nf], x[:, nf:]
            xfh = f16(xf)
            fast = xfh @ Wd["wh"][:nf] + q8(xf - xfh, 2.0 ** 18) @ Wd["w8"][:nf] + q8(xf, 2.0 ** 6) @ Wd["wl8"][:nf]
            if xr.shape[1] == 0:
                return fast
            xrh = f16(xr); xrl = f16(xr - xrh)
            return fast + xrh @ Wd["wh"][nf:] + xrl @ Wd["wh"][nf:] + xrh @ Wd["wl"][nf:]
        if nx and (n.startswith("f16+2i8") or n.startswith("f16+2f8k")):
            xi, Wi = x[:, :nx], {k: v[:nx] for k, v in Wd.items()}
            xr, Wr = x[:, nx:], {k: v[nx:] for k, v in Wd.items()}
            return f16(xi) @ Wi["wh"] + f16(xi) @ Wi["wl"] + self.mm(xr, Wr)
        if n == "exact":
            return x @ Wd["w"]
        xh = f16(x)
        xl = f16(x - xh)
        if n == "f16x3" or n.startswith("f16x3wl8"):
            return xh @ Wd["wh"] + xl @ Wd["wh"] + xh @ Wd["wl"]
        if n == "f16x2a":
            return xh @ Wd["wh"] + xh @ Wd["wl"]
        if n == "f16x2w":
            return xh @ Wd["wh"] + xl @ Wd["wh"]
        if n == "f16x1":
            return xh @ Wd["wh"]
        if n == "f16+2f8":
            xlr = x - xh
            x8 = q8(x, pow2_scale(x, 256.0))
            xl8 = q8(xlr, pow2_scale(xlr, 256.0)) if float(xlr.abs().max()) > 0 else xlr
            return xh @ Wd["wh"] + xl8 @ Wd["w8"] + x8 @ Wd["wl8"]
        if n == "f16+2f8k":
            xlr = x - xh
            return xh @ Wd["wh"] + q8(xlr, 2.0 ** 18) @ Wd["w8"] + q8(x, 2.0 ** 6) @ Wd["wl8"]
        if n == "f16+2f8k_a":      # only w_hi * x_lo on the fp8 pipe
            xlr = x - xh
            return xh @ Wd["wh"] + q8(xlr, 2.0 ** 18) @ Wd["w8"] + xh @ Wd["wl"]
        if n == "f16+2f8k_b":      # only w_lo * x_hi on the fp8 pipe
            xlr = x - xh
            return xh @ Wd["wh"] + xl @ Wd["wh"] + q8(x, 2.0 ** 6) @ Wd["wl8"]
        if n.startswith("f16+2i8"):
            xlr = x - xh
            x8 = torch.round(x * 127.0).clamp(-127, 127) / 127.0
            xl8 = torch.round(xlr * (127.0 * 4096.0)).clamp(-127, 127) / (127.0 * 4096.0)
            return xh @ Wd["wh"] + xl8 @ Wd["w8"] + x8 @ Wd["wl8"]
        if n == "f16+2f6":
            xlr = x - xh
            return xh @ Wd["wh"] + q6_block(xlr, 1) @ Wd["w8"] + q6_block(x, 1) @ Wd["wl8"]
        raise ValueError(n)


def unpack(blob, C):
    q = [0]

    def take(*shape):
        n = int(np.prod(shape))
        a = torch.tensor(blob[q[0]:q[0] + n].reshape(shape), dtype=D)
        q[0] += n
        return a
    L = []
    for cin, H in ((C, 128), (256, 160)):
        dirs = []
        for _ in range(2):
            dirs.append((take(cin, 4 * H), take(H, 4 * H), take(4 * H)))
        L.append(dirs)
    W4, b4 = take(33 * 320, 128), take(128)
    W51, b51, W52, b52 = take(128, 128), take(128), take(128, 128), take(128)
    Wg, bg, Wz, bz = take(128, 21), take(21), take(128, 3), take(3)
    return L, (W4, b4, W51, b51, W52, b52, Wg, bg, Wz, bz)


def forward(X, blob, C, sch):
    L, (W4, b4, W51, b51, W52, b52, Wg, bg, Wz, bz) = unpack(blob, C)
    x = torch.tensor(X, dtype=D)           # [B, 33, C]
    B = x.shape[0]
    for li, dirs in enumerate(L):
        outs = []
        for d, (Kw, R, b) in enumerate(dirs):
            H = R.shape[0]
            Wd = sch.prep_w(torch.cat([Kw, R], 0))
            h = torch.zeros(B, H, dtype=D)
            c = torch.zeros(B, H, dtype=D)
            ys = [None] * 33
            order = range(33) if d == 0 else range(32, -1, -1)
            for t in order:
                z = sch.mm(torch.cat([x[:, t, :], h], 1), Wd, nx=(Kw.shape[0] if li == 0 else 0), nf=(Kw.shape[0] if li == 1 else None)) + b
                i, f, g, o = z[:, :H], z[:, H:2 * H], z[:, 2 * H:3 * H], z[:, 3 * H:]
                c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
                h = torch.sigmoid(o) * torch.tanh(c)
                ys[t] = h
            outs.append(torch.stack(ys, 1))
        x = torch.cat(outs, 2)
    flat = x.reshape(B, -1)
    selu = torch.nn.functional.selu
    a4 = selu(sch.mm(flat, sch.prep_w(W4), nf=(flat.shape[1] if sch.name == "f16+2f8k_x4" else None)) + b4)
    a51 = selu(a4 @ W51 + b51)
    a52 = selu(a4 @ W52 + b52)
    p1 = torch.softmax(selu(a51 @ Wg + bg), 1)
    p2 = torch.softmax(selu(a52 @ Wz + bz), 1)
    return torch.cat([p1, p2], 1)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    gain = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    rng = np.random.RandomState(7)
    # windows shaped like the pileup tensor: counts 0..depth on a few channels, negative reference channel, some deep sites
    C = 18
    X = np.zeros((n, 33, C), np.int32)
    for s in range(n):
        depth = int(rng.choice([6, 12, 20, 40, 90, 216]))
        for t in range(33):
            k = rng.randint(0, 4)
            fwd = rng.binomial(depth, 0.5)
            X[s, t, k] = -fwd
            X[s, t, 9 + k] = -(depth - fwd)
            for _ in range(rng.randint(0, 3)):
                X[s, t, rng.randint(0, C)] += rng.randint(1, max(2, depth // 3))
    if os.environ.get("PROBE_HARSH"):      # the GPU precision test's inputs: every channel uniform in +-216 / +-20
        r2 = np.random.RandomState(11)
        X = np.concatenate([r2.randint(-216, 217, size=(40, 33, C)), r2.randint(-20, 21, size=(60, 33, C)), np.zeros((3, 33, C), int)]).astype(np.int32)
    blob = synth.random_weights(C).astype(np.float64)
    # gain > 1 scales every kernel (not the biases): a stand-in for trained weights with larger norms
    blob = blob.copy()
    if gain != 1.0:
        L = synth.random_weights(C)
        blob = (L * gain).astype(np.float64)
    ref = forward(X, blob, C, Scheme("exact"))
    print("sites %d, weight gain %.2f, max P spread %.3f" % (n, gain, float(ref.max())))
    names = ("f16x3", "f16x3wl8r", "f16x3wl8b", "f16+2i8g", "f16+2i8t", "f16+2i8r", "f16+2f8", "f16+2f8k", "f16+2f6", "f16x2a", "f16x2w", "f16x1")
    if len(sys.argv) > 3:
        names = tuple(sys.argv[3].split(","))
    for name in names:
        p = forward(X, blob, C, Scheme(name))
        d = (p - ref).abs()
        print("%-8s max|dP| %.3e   mean|dP| %.3e" % (name, float(d.max()), float(d.mean())))


if __name__ == "__main__":
    main()
