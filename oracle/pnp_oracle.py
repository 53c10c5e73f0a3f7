"""CPU ORACLE for the PnP-ADMM CS-MRI hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, on the CPU, the algorithm of the reference's per-iteration loop so
the HIP path can be checked against it.  Only `tests/`, `__graft_entry__.smoke()` and
the `cpu_baseline` leg of `bench.py` may import it; nothing under
`dt4image_restoration_amd/` does, and the product path raises when the HIP library is
missing rather than falling back to this code.

Parity status: PINNED.  `tests/golden/*.npz` were produced by importing the reference's
own functions in the build container (`tests/golden/gen_golden.py`), and
`tests/test_oracle_golden.py` checks every function below against them.  The reference
ships no tests or golden vectors of its own, so those fixtures are the pin.

Third-party arithmetic: conv2d / max_pool2d / bilinear upsample / fftn live in PyTorch
ATen (the reference pins no version; fixtures made with torch 2.10.0).  The torch calls
here are the same ATen entry points the reference reaches; `oracle/pnp_ref.c` restates
those operators once more in plain C (double accumulation) as an ATen-independent check.

Each function cites the reference lines it follows (paths relative to /root/reference).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Mapping, Optional

import numpy as np
import torch
import torch.nn.functional as F

LEAKY = 0.2


def _t(a, dtype):
    if isinstance(a, torch.Tensor):
        return a.to(dtype)
    return torch.from_numpy(np.asarray(a)).to(dtype)


def _bf16(t: torch.Tensor) -> torch.Tensor:
    """Round to bfloat16 (nearest even) and back: the operand rounding of the engine's PNP_FLAG_BF16_CONVS mode."""
    return t.to(torch.bfloat16).to(t.dtype)


# first conv index (execution order 0..27) of each ConvBlock, for per-layer operand plans
_STAGE_BASE = {"inc.conv": 0, "down1.mpconv.1": 3, "down2.mpconv.1": 6, "down3.mpconv.1": 9, "down4.mpconv.1": 12,
               "up1.conv": 15, "up2.conv": 18, "up3.conv": 21, "up4.conv": 24}


class Bf16Plan:
    """Operand rounding of the engine's PNP_FLAG_BF16_CONVS mode, per conv layer (index 0..27 in execution order; only the
    conv3x3 layers with Cin >= 32 take part).

    acts:          conv inputs are rounded to bfloat16 (nearest even)
    weight_terms:  2 (the engine's default) - a weight is carried as TWO bfloat16 terms hi = bf16(w), lo = bf16(w - hi): the
                   matrix pipe multiplies the rounded activations by hi and by lo and accumulates both products in f32, which
                   is the convolution with the 16-bit-mantissa weight hi + lo (the sum is exact in f32);
                   1 - one bfloat16 term (PNP_BF16_W1: the round-3 arithmetic, 0.015 dB off the f32 reference after 50
                   iterations at 512 x 512); 0 - f32 weights (experiments only)
    layer_terms:   {layer index: terms} overrides (tools/bf16_drift.py)
    `True` where a plan is expected means Bf16Plan()."""

    def __init__(self, acts: bool = True, weight_terms: int = 2, layer_terms=None, lo_format: str = "bf16"):
        self.acts, self.weight_terms, self.layer_terms = acts, weight_terms, dict(layer_terms or {})
        # lo_format (round-5 gate experiment, tools/bf16_drift.py `lo-e4m3`): "e4m3" evaluates the SECOND weight term on 8-bit operands -
        # lo * 2^s rounded to OCP float8 e4m3 (s per layer: the largest |lo| lands just under 448) times the bf16-rounded activations
        # rounded once more to e4m3, products exact, f32 accumulate - as the block-scaled fp8 matrix instruction would
        self.lo_format = lo_format

    def conv(self, li: int, x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        """conv3x3 + bias of layer li under this plan"""
        terms = self.layer_terms.get(li, self.weight_terms)
        if self.lo_format == "bf16" or terms != 2:
            x, w = self.operands(li, x, w)
            return F.conv2d(x, w, b, stride=1, padding=1)
        xb = _bf16(x) if self.acts else x
        hi = _bf16(w)
        lo = w - hi
        s = torch.floor(torch.log2(448.0 / lo.abs().max().clamp_min(1e-30)))
        lo8 = (lo * 2.0 ** s).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(w.dtype) * 2.0 ** (-s)
        x8 = xb.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(x.dtype)
        return F.conv2d(xb, hi, b, stride=1, padding=1) + F.conv2d(x8, lo8, None, stride=1, padding=1)

    def operands(self, li: int, x: torch.Tensor, w: torch.Tensor):
        if self.acts:
            x = _bf16(x)
        terms = self.layer_terms.get(li, self.weight_terms)
        if terms == 2:
            hi = _bf16(w)
            w = hi + _bf16(w - hi)
        elif terms == 1:
            w = _bf16(w)
        return x, w


def _plan(q):
    return None if not q else (q if isinstance(q, Bf16Plan) else Bf16Plan())


def _stage(sd, prefix: str, x: torch.Tensor, bf16_operands=False) -> torch.Tensor:
    """ConvBlock: 3 x [conv3x3 s1 p1 + bias, LeakyReLU(0.2)]  (evaluation/noise.py:88-98, 75-85).

    bf16_operands (BASELINE configs[4], not the reference's arithmetic; True or a Bf16Plan): every conv with Cin >= 32 sees
    its input tensor rounded to bfloat16 and its weights as one or two bfloat16 terms, as the plan says; products and sums
    stay in the working precision, bias is not rounded."""
    plan = _plan(bf16_operands)
    for j in range(3):
        w = sd[f"{prefix}.conv-{j}.conv2d.weight"]
        b = sd[f"{prefix}.conv-{j}.conv2d.bias"]
        if plan is not None and w.shape[1] >= 32:
            x = F.leaky_relu(plan.conv(_STAGE_BASE[prefix] + j, x, w, b), LEAKY)
        else:
            x = F.leaky_relu(F.conv2d(x, w, b, stride=1, padding=1), LEAKY)
    return x


def _up(sd, prefix: str, x1: torch.Tensor, x2: torch.Tensor, bf16_operands: bool = False) -> torch.Tensor:
    """up.forward: bilinear x2 (align_corners=True), pad to the skip's size, cat([skip, up])
    then ConvBlock  (evaluation/noise.py:44-61)."""
    x1 = F.interpolate(x1, scale_factor=2, mode="bilinear", align_corners=True)
    dy = x2.shape[2] - x1.shape[2]
    dx = x2.shape[3] - x1.shape[3]
    x1 = F.pad(x1, (dx // 2, dx - dx // 2, dy // 2, dy - dy // 2))
    return _stage(sd, prefix, torch.cat([x2, x1], dim=1), bf16_operands)


def unet_forward(sd: Mapping[str, torch.Tensor], x: torch.Tensor, return_stages: bool = False,
                 bf16_operands: bool = False):
    """UNet(2,1).forward on [N,2,H,W]  (evaluation/noise.py:119-133)."""
    q = bf16_operands
    x1 = _stage(sd, "inc.conv", x, q)
    x2 = _stage(sd, "down1.mpconv.1", F.max_pool2d(x1, 2), q)
    x3 = _stage(sd, "down2.mpconv.1", F.max_pool2d(x2, 2), q)
    x4 = _stage(sd, "down3.mpconv.1", F.max_pool2d(x3, 2), q)
    x5 = _stage(sd, "down4.mpconv.1", F.max_pool2d(x4, 2), q)
    y1 = _up(sd, "up1.conv", x5, x4, q)
    y2 = _up(sd, "up2.conv", y1, x3, q)
    y3 = _up(sd, "up3.conv", y2, x2, q)
    y4 = _up(sd, "up4.conv", y3, x1, q)
    residual = F.conv2d(y4, sd["outc.conv.weight"], sd["outc.conv.bias"])
    out = x[:, :1] + residual
    if return_stages:
        return out, OrderedDict(x1=x1, x2=x2, x3=x3, x4=x4, x5=x5, y1=y1, y2=y2, y3=y3, y4=y4)
    return out


def denoise(sd: Mapping[str, torch.Tensor], x: torch.Tensor, sigma: torch.Tensor, bf16_operands: bool = False) -> torch.Tensor:
    """UNetDenoiser2D.forward(x[N,1,H,W], sigma[N]) -> [N,1,H,W] in [0,1]
    (evaluation/noise.py:155-164)."""
    n, _, h, w = x.shape
    noise_map = torch.ones(n, 1, h, w, dtype=x.dtype) * sigma.reshape(n, 1, 1, 1).to(x.dtype)
    return torch.clamp(unet_forward(sd, torch.cat([x, noise_map], dim=1), bf16_operands=bf16_operands), 0, 1)


def fft2c(img: torch.Tensor) -> torch.Tensor:
    """Centred orthonormal 2-D DFT  (evaluation/utils/transformations.py:6-12)."""
    t = torch.fft.ifftshift(img, dim=(-2, -1))
    t = torch.fft.fftn(t, dim=(-2, -1), norm="ortho")
    return torch.fft.fftshift(t, dim=(-2, -1))


def ifft2c(img: torch.Tensor) -> torch.Tensor:
    """Centred orthonormal inverse 2-D DFT  (evaluation/utils/transformations.py:14-19)."""
    t = torch.fft.ifftshift(img, dim=(-2, -1))
    t = torch.fft.ifftn(t, dim=(-2, -1), norm="ortho")
    return torch.fft.fftshift(t, dim=(-2, -1))


def psnr(output: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """torch_psnr: clamp(Re x,0,1), per-slice MSE, 10 log10(1/mse) -> [N,1]  (evaluation/env.py:120-125)."""
    n = output.shape[0]
    out = torch.clamp(output.real if output.is_complex() else output, 0, 1)
    mse = torch.mean((out.reshape(n, -1) - gt.reshape(n, -1)) ** 2, dim=1)
    return (10 * torch.log10(1.0 / mse)).unsqueeze(1)


def reset(data: Mapping[str, np.ndarray], dtype=torch.float32) -> "OrderedDict[str, torch.Tensor]":
    """PnPEnv.reset  (evaluation/env.py:57-71), generalised from (1,1,128,128) to [N,1,H,W]:
    x = complex(x0); z = x; u = 0; mask bool broadcast over the batch; y0 complex."""
    cdtype = torch.complex64 if dtype == torch.float32 else torch.complex128
    x0 = _t(data["x0"], dtype)
    y0 = _t(data["y0"], dtype)
    x = torch.view_as_complex(x0.contiguous())
    n, _, h, w = x.shape
    mask = torch.from_numpy(np.asarray(data["mask"])).reshape(1, 1, h, w).to(torch.bool)
    return OrderedDict(x=x.clone(), y0=torch.view_as_complex(y0.contiguous()).to(cdtype), z=x.clone(),
                       u=torch.zeros_like(x), mask=mask, gt=_t(data["gt"], dtype),
                       T=torch.zeros(n, dtype=dtype))


def admm_step(sd: Mapping[str, torch.Tensor], st: "OrderedDict[str, torch.Tensor]",
              mu: torch.Tensor, sigma_d: torch.Tensor, T: Optional[torch.Tensor] = None, bf16_operands: bool = False):
    """PnPEnv.step  (evaluation/env.py:74-100), batched over N independent slices.

    The reference is hard-wired to N=1 with scalar mu; for N>1 the oracle is DEFINED as N
    independent single-slice reference steps, which is what broadcasting
    mu[n], sigma_d[n] over slice n computes.  Slices whose T[n] > 0.5 are left untouched
    and flagged done (env.py:79-81).  Returns (state, done[N] bool)."""
    x, y0, z, u, mask = st["x"], st["y0"], st["z"], st["u"], st["mask"]
    n = z.shape[0]
    rdtype = z.real.dtype
    mu = mu.reshape(n, 1, 1, 1).to(rdtype)
    sigma_d = sigma_d.reshape(n).to(rdtype)
    done = (T.reshape(n) > 0.5) if T is not None else torch.zeros(n, dtype=torch.bool)
    act = ~done
    if act.any():
        ia = act.nonzero().flatten()
        za, ua, y0a, mua = z[ia], u[ia], y0[ia], mu[ia]
        xa = denoise(sd, (za - ua).real, sigma_d[ia], bf16_operands)  # env.py:85-86
        zf = fft2c(xa + ua)                                            # env.py:87
        temp = (mua * zf + y0a) / (1 + mua)                            # env.py:88-89
        zf = torch.where(mask, temp, zf)                               # env.py:90  z[mask] = temp[mask]
        zn = ifft2c(zf)                                                # env.py:91
        un = ua + xa - zn                                              # env.py:93
        xs = st["x"]
        if xs.is_complex():                                            # first call: x was complex x0
            xs = xs.real.clone()
        xs[ia] = xa
        z = z.clone(); u = u.clone()
        z[ia] = zn
        u[ia] = un
        st["x"], st["z"], st["u"] = xs, z, u
        st["T"] = st["T"] + act.to(rdtype) / 30                        # env.py:98
    return st, done


def run_episode(sd, data, mu_tab: np.ndarray, sig_tab: np.ndarray, iters: int, dtype=torch.float32,
                record_psnr: bool = True, bf16_operands=False):
    """reset + `iters` steps with per-slice parameter tables [N,iters]; returns (state, psnr[N,iters])."""
    st = reset(data, dtype)
    n = st["z"].shape[0]
    mu_t = torch.from_numpy(np.asarray(mu_tab)).to(dtype).reshape(n, -1)
    sg_t = torch.from_numpy(np.asarray(sig_tab)).to(dtype).reshape(n, -1)
    hist = []
    for t in range(iters):
        st, _ = admm_step(sd, st, mu_t[:, t], sg_t[:, t], None, bf16_operands)
        if record_psnr:
            hist.append(psnr(st["x"], st["gt"])[:, 0])
    return st, (torch.stack(hist, dim=1) if hist else None)


def torch_weights(sd_np: Mapping[str, np.ndarray], dtype=torch.float32) -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(np.array(v, copy=True)).to(dtype) for k, v in sd_np.items()}
