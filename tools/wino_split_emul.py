#!/usr/bin/env python3
"""Gate experiment (CPU emulation): the F(4x4,3x3) frequency-domain products on the bf16 / f16 matrix pipe with SPLIT f32 operands.

A 256-channel layer (the configuration of tools/wino_points.py), points (0, +-3/4, +-3/2, inf), transforms in f32 exactly as
the kernel does; only the 36 channel-sum GEMMs  M[xi] = sum_c V[xi][c] U[xi][c]  change:

  f32 chain          one rounding per product-accumulate in channel order (v_mfma_f32_16x16x4_f32: bitwise an fmaf chain)
  bf16 x3, 6 terms   v = v0 + v1 + v2, u = u0 + u1 + u2 (bf16, round-to-nearest-even residual splits: exact), the six cross
                     terms of order <= 2 (v0u0, v0u1, v1u0, v1u1, v0u2, v2u0), f32 accumulate
  bf16 x3, 3 terms   v0u0, v0u1, v1u0 only (what dropping the order-2 terms costs)
  f16 x2, 3 terms    v = v0 + v1, u = u0 + u1 (f16), v0u0 + v0u1 + v1u0, operands scaled by a power of two per tensor so
                     that max|.| sits just under the f16 range (without scaling small values fall into f16 subnormals)

The matrix pipe's internal accumulation is not documented; two models bracket it: `blk` = the 16 products of one MFMA
(k = 16 channels x 1 term) summed exactly, one f32 rounding per MFMA; `seq` = one f32 rounding per product (like the f32 form).
exp/split_mfma.hip measures the real instruction on the GPU.

    python tools/wino_split_emul.py        # ~1 minute
"""
import numpy as np
import torch
import torch.nn.functional as F

from wino_points import winograd
from fractions import Fraction as Fr


def bf16_rne(x):
    """float32 -> nearest bfloat16 (as float32)"""
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split_bf16(x, n):
    out, r = [], x.astype(np.float32)
    for _ in range(n):
        h = bf16_rne(r)
        out.append(h)
        r = (r - h).astype(np.float32)          # exact
    return out, r


def split_f16(x, n):
    out, r = [], x.astype(np.float32)
    for _ in range(n):
        h = r.astype(np.float16).astype(np.float32)
        out.append(h)
        r = (r - h).astype(np.float32)
    return out, r


def gemm_chain(V, U):
    """M[t,k] = sum_c V[t,c] U[k,c], one f32 rounding per product-accumulate (fma: product exact, then one rounding)."""
    T, C = V.shape[0], V.shape[1]
    M = np.zeros((T, U.shape[0]) + V.shape[2:], dtype=np.float32)
    for c in range(C):
        M = (M.astype(np.float64) + V[:, None, c].astype(np.float64) * U[None, :, c].astype(np.float64)).astype(np.float32)
    return M


def gemm_terms(terms, model, blk=16):
    """terms: list of (Vpiece, Upiece); accumulate chunk by chunk (blk channels), term by term inside a chunk."""
    V0, U0 = terms[0]
    T, C = V0.shape[0], V0.shape[1]
    M = np.zeros((T, U0.shape[0]) + V0.shape[2:], dtype=np.float32)
    for c0 in range(0, C, blk):
        for (Vp, Up) in terms:
            if model == "blk":
                part = np.einsum("tc...,kc...->tk...", Vp[:, c0:c0 + blk].astype(np.float64), Up[:, c0:c0 + blk].astype(np.float64))
                M = (M.astype(np.float64) + part).astype(np.float32)
            else:
                for c in range(c0, c0 + blk):
                    M = (M.astype(np.float64) + Vp[:, None, c].astype(np.float64) * Up[None, :, c].astype(np.float64)).astype(np.float32)
    return M


def main():
    np.random.seed(1)
    C, K, HW = 256, 16, 32
    x = (np.random.rand(C, HW, HW).astype(np.float32) * 2 - 0.5)
    w = ((np.random.rand(K, C, 3, 3).astype(np.float32) * 2 - 1) * np.float32(np.sqrt(3.0 / (9 * C)) * 1.4)).astype(np.float32)
    ref = F.conv2d(torch.from_numpy(x.astype(np.float64))[None], torch.from_numpy(w.astype(np.float64)), padding=1)[0].numpy()
    sc = np.abs(ref).max()
    AT, G, BT = winograd(4, 3, (0, Fr(3, 4), -Fr(3, 4), Fr(3, 2), -Fr(3, 2)))
    dt = np.float32
    bt, at = BT.astype(dt), AT.astype(dt)
    xp = np.zeros((C, HW + 2, HW + 2), dtype=dt)
    xp[:, 1:-1, 1:-1] = x
    U = np.einsum("ij,kcjl,ml->kcim", G, w.astype(np.float64), G).astype(dt)                       # [K, C, 6, 6]
    tiles = [(ty, tx) for ty in range(0, HW, 4) for tx in range(0, HW, 4)]
    d = np.stack([xp[:, ty:ty + 6, tx:tx + 6] for ty, tx in tiles])                                # [T, C, 6, 6]
    V = np.einsum("tcil,ml->tcim", np.einsum("ij,tcjl->tcil", bt, d).astype(dt), bt).astype(dt)    # f32 transforms (two roundings)

    def finish(M):
        Y = np.einsum("tkil,ml->tkim", np.einsum("ij,tkjl->tkil", at, M).astype(dt), at).astype(dt)
        out = np.zeros((K, HW, HW), dtype=dt)
        for i, (ty, tx) in enumerate(tiles):
            out[:, ty:ty + 4, tx:tx + 4] = Y[i]
        return out

    def report(name, M):
        o = finish(M)
        print("%-46s max %.2e  rms %.2e" % (name, np.abs(o - ref).max() / sc, np.sqrt(((o - ref) ** 2).mean()) / sc), flush=True)

    Mex = np.einsum("tc...,kc...->tk...", V.astype(np.float64), U.astype(np.float64)).astype(dt)
    report("exact products + sums (transform error only)", Mex)
    report("f32 chain (today's kernel)", gemm_chain(V, U))
    (v0, v1, v2), rv = split_bf16(V, 3)
    (u0, u1, u2), ru = split_bf16(U, 3)
    print("  bf16 x3 split residuals: V %.1e  U %.1e (must be 0)" % (np.abs(rv).max(), np.abs(ru).max()))
    six = [(v1, u1), (v0, u2), (v2, u0), (v0, u1), (v1, u0), (v0, u0)]                                 # small terms first
    for model in ("blk", "seq"):
        report("bf16 x3, 6 terms, %s" % model, gemm_terms(six, model))
    report("bf16 x3, 3 terms, blk", gemm_terms([(v0, u1), (v1, u0), (v0, u0)], "blk"))
    # f16 x2 with per-tensor power-of-two scaling
    for scaled in (True, False):
        sv = 2.0 ** np.floor(np.log2(32768.0 / np.abs(V).max())) if scaled else 1.0
        su = 2.0 ** np.floor(np.log2(32768.0 / np.abs(U).max())) if scaled else 1.0
        (a0, a1), _ = split_f16(V * np.float32(sv), 2)
        (b0, b1), _ = split_f16(U * np.float32(su), 2)
        for model in ("blk", "seq"):
            M = gemm_terms([(a0, b1), (a1, b0), (a0, b0)], model)
            report("f16 x2, 3 terms, %s, %s" % ("scaled" if scaled else "unscaled", model), (M.astype(np.float64) / (sv * su)).astype(dt))
    # small activations (PyTorch-default init attenuates ~3x per layer): the same layer with inputs x 1e-4
    print("inputs scaled by 1e-4 (deep layers under the default init):")
    Vs = (V * np.float32(1e-4)).astype(dt)
    Ms = np.einsum("tc...,kc...->tk...", Vs.astype(np.float64), U.astype(np.float64))
    def rel(M):
        return np.abs(M - Ms).max() / np.abs(Ms).max()
    print("  f32 chain                %.2e" % rel(gemm_chain(Vs, U)))
    (v0, v1, v2), _ = split_bf16(Vs, 3)
    print("  bf16 x3, 6 terms, blk    %.2e" % rel(gemm_terms([(v1, u1), (v0, u2), (v2, u0), (v0, u1), (v1, u0), (v0, u0)], "blk")))
    (a0, a1), _ = split_f16(Vs, 2)
    (b0, b1), _ = split_f16(U, 2)
    print("  f16 x2 unscaled, blk     %.2e" % rel(gemm_terms([(a0, b1), (a1, b0), (a0, b0)], "blk")))


if __name__ == "__main__":
    main()
