"""Seeded synthetic CS-MRI problems in the reference's evaluation-data layout.

The reference evaluates on `.mat` files holding `x0, y0, ATy0` (float32 [...,H,W,2],
real/imag last), `mask` [H,W] and `gt` [1,H,W] (dataset/datasets.py:153-160,191-199); the
files are an external download.  This module builds the same dict from an analytic
phantom so every box (and the golden-vector generator) sees identical inputs:

* gt      : sum of random ellipses in [0,1], per-slice parameters from (seed, slice index)
* mask    : radial lines through the k-space centre, sampled fraction >= 1/accel
* y0      : mask * (fft_c(gt) + complex Gaussian noise, sigma_n per component)
* x0=ATy0 : ifft_c(y0); x0 real part clipped at 0 like datasets.py:160,199
            (np.clip on the stacked array clips the imaginary plane too - reproduced)
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np

from .weights import hash_uniform


def fft2c_np(a: np.ndarray) -> np.ndarray:
    """Centred orthonormal 2-D DFT over the last two axes (transformations.py:6-12)."""
    return np.fft.fftshift(np.fft.fft2(np.fft.ifftshift(a, axes=(-2, -1)), norm="ortho"), axes=(-2, -1))


def ifft2c_np(a: np.ndarray) -> np.ndarray:
    """Centred orthonormal inverse 2-D DFT (transformations.py:14-19)."""
    return np.fft.fftshift(np.fft.ifft2(np.fft.ifftshift(a, axes=(-2, -1)), norm="ortho"), axes=(-2, -1))


def radial_mask(h: int, w: int, accel: float) -> np.ndarray:
    """Bool [h,w]: L lines through the centre at angles k*pi/L, nearest-pixel rasterised,
    L = smallest count whose sampled fraction is >= 1/accel."""
    cy, cx = h // 2, w // 2
    r = math.hypot(h, w) / 2.0
    t = np.arange(-r, r + 0.25, 0.25)
    target = 1.0 / accel

    def raster(nlines: int) -> np.ndarray:
        m = np.zeros((h, w), dtype=bool)
        for k in range(nlines):
            th = math.pi * k / nlines
            ys = np.rint(cy + t * math.sin(th)).astype(np.int64)
            xs = np.rint(cx + t * math.cos(th)).astype(np.int64)
            ok = (ys >= 0) & (ys < h) & (xs >= 0) & (xs < w)
            m[ys[ok], xs[ok]] = True
        return m

    lo, hi = 1, 4 * max(h, w)
    while lo < hi:                       # sampled fraction is monotone enough in L; bisect
        mid = (lo + hi) // 2
        if raster(mid).mean() >= target:
            hi = mid
        else:
            lo = mid + 1
    return raster(lo)


def phantom(h: int, w: int, seed: int) -> np.ndarray:
    """float64 [h,w] in [0,1]: 6-10 soft-edged ellipses."""
    u = (hash_uniform(seed, 7001, 64).astype(np.float64) + 1.0) * 0.5   # [0,1)
    n_ell = 6 + int(u[0] * 5)
    yy, xx = np.meshgrid((np.arange(h) + 0.5) / h * 2 - 1, (np.arange(w) + 0.5) / w * 2 - 1, indexing="ij")
    img = np.zeros((h, w))
    # body ellipse
    body = ((xx / 0.85) ** 2 + (yy / 0.9) ** 2)
    img += 0.35 * np.clip((1.0 - body) * 12.0, 0.0, 1.0)
    for e in range(n_ell):
        p = u[1 + 6 * e: 7 + 6 * e]
        cx, cy = (p[0] - 0.5) * 1.1, (p[1] - 0.5) * 1.1
        ax, ay = 0.08 + 0.35 * p[2], 0.08 + 0.35 * p[3]
        th = math.pi * p[4]
        amp = 0.15 + 0.5 * p[5]
        xr = (xx - cx) * math.cos(th) + (yy - cy) * math.sin(th)
        yr = -(xx - cx) * math.sin(th) + (yy - cy) * math.cos(th)
        d = (xr / ax) ** 2 + (yr / ay) ** 2
        img += amp * np.clip((1.0 - d) * 10.0, 0.0, 1.0) * np.clip((1.0 - body) * 12.0, 0.0, 1.0)
    return np.clip(img / max(img.max(), 1e-9), 0.0, 1.0)


def _gauss(seed: int, stream: int, count: int) -> np.ndarray:
    u1 = (hash_uniform(seed, stream, count).astype(np.float64) + 1.0) * 0.5
    u2 = (hash_uniform(seed, stream + 1, count).astype(np.float64) + 1.0) * 0.5
    u1 = np.maximum(u1, 2.0 ** -25)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2)


def make_problem(n: int, h: int, w: int, accel: float = 4.0, sigma_n: float = 10.0 / 255.0,
                 seed: int = 1234, first_slice: int = 0) -> Dict[str, np.ndarray]:
    """Collated-batch dict with the `.mat` keys the reference's `PnPEnv.reset` reads
    (env.py:57-71): x0, y0, ATy0 float32 [n,1,h,w,2]; mask bool [h,w]; gt float32 [n,1,h,w]; plus x0_raw [n,1,h,w].
    Slice i depends only on (seed, first_slice + i), so shards of one job agree with the
    unsharded job."""
    mask = radial_mask(h, w, accel)
    gt = np.empty((n, 1, h, w), dtype=np.float32)
    x0 = np.empty((n, 1, h, w, 2), dtype=np.float32)
    y0 = np.empty((n, 1, h, w, 2), dtype=np.float32)
    aty0 = np.empty((n, 1, h, w, 2), dtype=np.float32)
    for i in range(n):
        s = seed + first_slice + i
        g = phantom(h, w, s)
        noise = (_gauss(s, 9001, h * w) + 1j * _gauss(s, 9003, h * w)).reshape(h, w) * sigma_n
        y = mask * (fft2c_np(g) + noise)
        a = ifft2c_np(y)
        gt[i, 0] = g
        y0[i, 0, ..., 0], y0[i, 0, ..., 1] = y.real, y.imag
        aty0[i, 0, ..., 0], aty0[i, 0, ..., 1] = a.real, a.imag
        x0[i, 0] = np.clip(aty0[i, 0], 0.0, None)           # datasets.py:160
    # x0_raw: the UNclipped real part of the zero-filled reconstruction - the policy's first state token in the reference
    # (datasets.py:162,201 read mat['x0'][..., 0], which the clip at :160,199 does not touch)
    return {"x0": x0, "y0": y0, "ATy0": aty0, "mask": mask, "gt": gt, "x0_raw": aty0[..., 0].copy()}


def param_table(n: int, iters: int, seed: int = 77):
    """Seeded per-slice (mu_t, sigma_t) schedule standing in for DT-chosen parameters at
    sizes the 128x128-only policy cannot see: mu in (0.05,0.6), sigma_d
    decaying 50/255 -> 5/255 with per-slice jitter.  float32 [n,iters] each."""
    u = (hash_uniform(seed, 31, n * iters).astype(np.float64).reshape(n, iters) + 1.0) * 0.5
    v = (hash_uniform(seed, 32, n * iters).astype(np.float64).reshape(n, iters) + 1.0) * 0.5
    mu = 0.05 + 0.55 * u
    t = np.arange(iters) / max(iters - 1, 1)
    sig = (50.0 * (5.0 / 50.0) ** t)[None, :] * (0.9 + 0.2 * v) / 255.0
    return mu.astype(np.float32), sig.astype(np.float32)
