#!/usr/bin/env python3
"""Which interpolation points for Winograd F(4x4,3x3) in float32?  (CPU, numpy; the numbers quoted in
dt4image_restoration_amd/csrc/winograd4_kernels.hip and DESIGN.md.)

Builds the Cook-Toom matrices A^T, G, B^T for points (0, +-a, +-b, inf) exactly (fractions), applies the algorithm to a random
256-channel layer in float32 (weights transformed in float64 and rounded once, as the library's host pack does; input
transform, products accumulated in channel order like an fma chain, output transform in float32) and reports the error
against the float64 direct convolution, relative to the output scale.

    python tools/wino_points.py            # ~2 minutes
"""
from fractions import Fraction as Fr

import numpy as np
import torch
import torch.nn.functional as F


def polymul(a, b):
    r = [Fr(0)] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            r[i + j] += x * y
    return r


def winograd(m, r, pts):
    """A^T [m x n], G [n x r], B^T [n x n] for n = m + r - 1 points: the finite `pts` plus infinity."""
    n = m + r - 1
    pts = [Fr(p) for p in pts]
    assert len(pts) == n - 1
    AT = [[p ** i for p in pts] + [Fr(1 if i == m - 1 else 0)] for i in range(m)]
    N = [np.prod([pts[j] - pts[l] for l in range(n - 1) if l != j]) for j in range(n - 1)]
    G = [[pts[j] ** k / N[j] for k in range(r)] for j in range(n - 1)] + [[Fr(0)] * (r - 1) + [Fr(1)]]
    BT = []
    for j in range(n - 1):
        poly = [Fr(1)]
        for l in range(n - 1):
            if l != j:
                poly = polymul(poly, [-pts[l], Fr(1)])
        BT.append(poly + [Fr(0)] * (n - len(poly)))
    poly = [Fr(1)]
    for l in range(n - 1):
        poly = polymul(poly, [-pts[l], Fr(1)])
    BT.append(poly)
    f = lambda M: np.array([[float(x) for x in row] for row in M])
    return f(AT), f(G), f(BT)


def conv2d_wino(x, w, AT, G, BT, dt, m):
    C, H, W = x.shape
    K = w.shape[0]
    n = BT.shape[0]
    xp = np.zeros((C, H + 2, W + 2), dtype=dt)
    xp[:, 1:-1, 1:-1] = x
    U = np.einsum("ij,kcjl,ml->kcim", G, w.astype(np.float64), G).astype(dt)
    out = np.zeros((K, H, W), dtype=dt)
    bt, at = BT.astype(dt), AT.astype(dt)
    for ty in range(0, H, m):
        for tx in range(0, W, m):
            d = xp[:, ty:ty + n, tx:tx + n]
            V = np.einsum("cil,ml->cim", np.einsum("ij,cjl->cil", bt, d).astype(dt), bt).astype(dt)
            M = np.zeros((K, n, n), dtype=dt)
            for c in range(C):
                M += U[:, c] * V[c][None]
            out[:, ty:ty + m, tx:tx + m] = np.einsum("kil,ml->kim", np.einsum("ij,kjl->kil", at, M).astype(dt), at).astype(dt)
    return out


def main():
    np.random.seed(1)
    C, K = 256, 16
    x = (np.random.rand(C, 32, 32).astype(np.float32) * 2 - 0.5)
    w = ((np.random.rand(K, C, 3, 3).astype(np.float32) * 2 - 1) * np.float32(np.sqrt(3.0 / (9 * C)) * 1.4)).astype(np.float32)
    ref = F.conv2d(torch.from_numpy(x.astype(np.float64))[None], torch.from_numpy(w.astype(np.float64)), padding=1)[0].numpy()
    d32 = F.conv2d(torch.from_numpy(x)[None], torch.from_numpy(w), padding=1)[0].numpy()
    sc = np.abs(ref).max()
    print("direct f32 sum          max %.2e rms %.2e" % (np.abs(d32 - ref).max() / sc, np.sqrt(((d32 - ref) ** 2).mean()) / sc))
    AT, G, BT = winograd(2, 3, (0, 1, -1))
    o = conv2d_wino(x, w, AT, G, BT, np.float32, 2)
    print("F(2x2) (0, +-1)         max %.2e rms %.2e" % (np.abs(o - ref).max() / sc, np.sqrt(((o - ref) ** 2).mean()) / sc))
    for a in (Fr(1, 2), Fr(5, 8), Fr(3, 4), Fr(7, 8), Fr(1)):
        for b in (Fr(1), Fr(5, 4), Fr(3, 2), Fr(7, 4), Fr(2)):
            if b <= a:
                continue
            AT, G, BT = winograd(4, 3, (0, a, -a, b, -b))
            o = conv2d_wino(x, w, AT, G, BT, np.float32, 4)
            print("F(4x4) (0, +-%s, +-%s)  max %.2e rms %.2e" % (a, b, np.abs(o - ref).max() / sc, np.sqrt(((o - ref) ** 2).mean()) / sc),
                  flush=True)


if __name__ == "__main__":
    main()
