#!/usr/bin/env python3
"""CPU emulation of the f16x2 frequency-domain products exactly as the kernel variant would do them (go / no-go for building it):
v = v0 + v1, u = u0 + u1 (f16, round to nearest even), FOUR cross terms ([v0|v0] x [u0|u1] + [v1|v1] x [u0|u1] on
v_mfma_f32_16x16x32_f16: each MFMA sums 16 channels x 2 terms, f32 accumulate, modelled as exact inside an MFMA and one f32
rounding per MFMA - the behaviour measured in exp/split_mfma.hip), with the kernel's BLOCK-FLOATING scale: per 16-channel chunk the
scale 2^K, K = 9 - exponent(max |d| over the chunk), K_acc = min over the chunks so far (accumulators rescaled when it drops,
a chunk with smaller data runs on the running scale), weights scaled once per layer.  Errors against f64, relative to the output
scale, next to the f32 fma chain of today's kernel.

    python tools/wino_f16x2_emul.py
"""
import numpy as np
import torch
import torch.nn.functional as F
from fractions import Fraction as Fr
from wino_points import winograd


def f16(x):
    return x.astype(np.float32).astype(np.float16).astype(np.float32)


def split2(x):
    h0 = f16(x)
    return h0, f16((x.astype(np.float32) - h0).astype(np.float32))


def layer(x, w, mode):
    C, HW = x.shape[0], x.shape[1]
    K = w.shape[0]
    dt = np.float32
    AT, G, BT = winograd(4, 3, (0, Fr(3, 4), -Fr(3, 4), Fr(3, 2), -Fr(3, 2)))
    bt, at = BT.astype(dt), AT.astype(dt)
    xp = np.zeros((C, HW + 2, HW + 2), dtype=dt)
    xp[:, 1:-1, 1:-1] = x
    U = np.einsum("ij,kcjl,ml->kcim", G, w.astype(np.float64), G).astype(dt)
    tiles = [(ty, tx) for ty in range(0, HW, 4) for tx in range(0, HW, 4)]
    d = np.stack([xp[:, ty:ty + 6, tx:tx + 6] for ty, tx in tiles])                                # [T, C, 6, 6]
    V = np.einsum("tcil,ml->tcim", np.einsum("ij,tcjl->tcil", bt, d).astype(dt), bt).astype(dt)
    T = V.shape[0]
    if mode == "f32":
        M = np.zeros((T, K, 6, 6), dtype=dt)
        for c in range(C):
            M = (M.astype(np.float64) + V[:, None, c].astype(np.float64) * U[None, :, c].astype(np.float64)).astype(dt)
    else:
        su = 2.0 ** np.floor(np.log2(16384.0 / np.abs(U).max()))
        u0, u1 = split2(U * dt(su))
        M = np.zeros((T, K, 6, 6), dtype=np.float64)                     # holds f32 values; scaled by 2^Kacc * su
        kacc = None
        for c0 in range(0, C, 16):
            # one workgroup = 32 tiles; here: one scale per chunk over ALL tiles (a workgroup's patch is a subset: its max is <= this one,
            # i.e. the kernel's scale is at least as fine)
            m = np.abs(xp[c0:c0 + 16]).max()
            k = 0 if m == 0 else 9 - int(np.floor(np.log2(m)))
            if kacc is None:
                kacc = k
            elif k < kacc:
                M = (M * 2.0 ** (k - kacc)).astype(dt).astype(np.float64)
                kacc = k
            v0, v1 = split2(V[:, c0:c0 + 16] * dt(2.0 ** kacc))
            for a in (v0, v1):                                            # two MFMAs: [a | a] x [u0 | u1]
                part = np.einsum("tc...,kc...->tk...", a.astype(np.float64), (u0[:, c0:c0 + 16].astype(np.float64)))
                part += np.einsum("tc...,kc...->tk...", a.astype(np.float64), (u1[:, c0:c0 + 16].astype(np.float64)))
                M = (M + part).astype(dt).astype(np.float64)
        M = (M / (2.0 ** kacc * su)).astype(dt)
    Y = np.einsum("tkil,ml->tkim", np.einsum("ij,tkjl->tkil", at, M.astype(dt)).astype(dt), at).astype(dt)
    out = np.zeros((K, HW, HW), dtype=dt)
    for i, (ty, tx) in enumerate(tiles):
        out[:, ty:ty + 4, tx:tx + 4] = Y[i]
    return out


def case(name, x, w):
    ref = F.conv2d(torch.from_numpy(x.astype(np.float64))[None], torch.from_numpy(w.astype(np.float64)), padding=1)[0].numpy()
    sc = np.abs(ref).max()
    res = []
    for mode in ("f32", "f16x2"):
        o = layer(x, w, mode)
        res.append((np.abs(o - ref).max() / sc, np.sqrt(((o - ref) ** 2).mean()) / sc))
    print("%-64s f32 chain max %.2e rms %.2e | f16x2 max %.2e rms %.2e" % (name, *res[0], *res[1]), flush=True)


def main():
    np.random.seed(1)
    C, K, HW = 256, 16, 32
    x = (np.random.rand(C, HW, HW).astype(np.float32) * 2 - 0.5)
    w = ((np.random.rand(K, C, 3, 3).astype(np.float32) * 2 - 1) * np.float32(np.sqrt(3.0 / (9 * C)) * 1.4)).astype(np.float32)
    case("uniform activations O(1)", x, w)
    case("activations x 1e-4 (deep layers under the default init)", x * np.float32(1e-4), w)
    case("activations x 300", x * np.float32(300), w)
    xs = x.copy(); xs[:128] *= np.float32(1e-3)
    case("first half of the channels x 1e-3 (scale drops mid-way)", xs, w)
    xs = x.copy(); xs[128:] *= np.float32(1e-3)
    case("second half of the channels x 1e-3 (small data on a coarse scale)", xs, w)
    xo = x.copy(); xo[5, 7, 9] = 400.0
    case("one outlier of 400 among O(1) activations", xo, w)
    xr = np.maximum(x * np.float32(3.0) - np.float32(2.0), 0)                 # sparse, ReLU-like
    case("sparse non-negative activations", xr, w)
    wl = w.copy(); wl[:, :, 1, 1] *= np.float32(50)
    case("weights with a dominant centre tap (x50)", x, wl)


if __name__ == "__main__":
    main()
