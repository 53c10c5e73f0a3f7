"""`fft` / `ifft`: drop-ins for the reference's centred orthonormal FFT pair
(/root/reference/evaluation/utils/transformations.py:6-19), running the LDS Stockham kernels of
libpnpadmm.so.  Power-of-two sizes >= 16 (the reference only ever passes 128 x 128)."""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from .engine import PnPEngine

_engines: Dict[Tuple[int, int, int, int], PnPEngine] = {}


def _engine(batch: int, h: int, w: int, dev: int) -> PnPEngine:
    key = (batch, h, w, dev)
    e = _engines.get(key)
    if e is None:
        e = _engines[key] = PnPEngine(batch, h, w, device=dev, denoiser=False)
    return e


def _run(img: torch.Tensor, inverse: bool) -> torch.Tensor:
    if not img.is_cuda:
        raise RuntimeError("fft/ifft: the HIP path needs a GPU tensor; there is no CPU path")
    if not img.is_complex():
        img = img.to(torch.complex64)
    h, w = img.shape[-2:]
    c = img.to(torch.complex64).contiguous()
    batch = c.numel() // (h * w)
    return _engine(batch, h, w, c.device.index).fft2c(c, inverse=inverse).reshape(img.shape)


def fft(img: torch.Tensor) -> torch.Tensor:
    return _run(img, False)


def ifft(img: torch.Tensor) -> torch.Tensor:
    return _run(img, True)
