"""`UNetDenoiser2D`: drop-in for the reference's plug-in regulariser object
(/root/reference/evaluation/noise.py:139-164) - same constructor argument, same call
signature `denoiser(x[N,1,H,W] f32, sigma[N]) -> [N,1,H,W] in [0,1]` - but the forward is
the HIP U-Net of libpnpadmm.so.  There is no torch.nn model inside."""
from __future__ import annotations

from typing import Dict, Mapping, Optional, Tuple

import numpy as np
import torch

from .engine import PnPEngine
from .weights import check_state_dict, generate_unet_weights


class UNetDenoiser2D:
    def __init__(self, ckpt_path: Optional[str] = None, state_dict: Optional[Mapping[str, object]] = None,
                 bf16_convs: bool = False):
        """ckpt_path: a `torch.save`d state_dict with the reference's 56 keys (noise.py:146-148).
        state_dict: the same mapping given directly (tensors or ndarrays).
        With neither, the reference raises ValueError (noise.py:143-145); so does this.
        bf16_convs: BASELINE configs[4] - conv operands rounded to bfloat16, f32 accumulate (PNP_FLAG_BF16_CONVS);
        NOT the reference's arithmetic, off by default."""
        if state_dict is None:
            if ckpt_path is None:
                raise ValueError("Default ckpt not found, you have to provide a ckpt path")
            state_dict = torch.load(ckpt_path, map_location="cpu")
        self.weights: Dict[str, np.ndarray] = check_state_dict(state_dict)
        self._engines: Dict[Tuple[int, int, int, int, int], PnPEngine] = {}
        self.bf16_convs = bool(bf16_convs)

    @classmethod
    def seeded(cls, seed: int = 0, init: str = "unit_gain", bf16_convs: bool = False) -> "UNetDenoiser2D":
        """Deterministic stand-in weights (the trained checkpoint is an external download)."""
        return cls(state_dict=generate_unet_weights(seed, init), bf16_convs=bf16_convs)

    # nn.Module-shaped no-ops the reference's callers use (env.py:33 `.to(device_type)`)
    def to(self, *_a, **_k):
        return self

    def eval(self):
        return self

    def engine_for(self, n: int, h: int, w: int, device_index: int, replica: int = 0) -> PnPEngine:
        """The handle for [n, h, w] on that device.  `replica` > 0: another handle of the same shape with its own workspace
        and k-space constants - what two sub-batches stepped concurrently on two streams need (drivers/greedy.run_pipelined)."""
        key = (n, h, w, device_index, replica)
        eng = self._engines.get(key)
        if eng is None:
            eng = PnPEngine(n, h, w, device=device_index, bf16_convs=self.bf16_convs)
            eng.load_weights(self.weights)
            self._engines[key] = eng
        return eng

    def forward(self, x: torch.Tensor, sigma: torch.Tensor) -> torch.Tensor:
        if x.dim() != 4 or x.shape[1] != 1:
            raise ValueError(f"denoiser expects x of shape [N,1,H,W], got {tuple(x.shape)}")
        n, _, h, w = x.shape
        if sigma.numel() != n:          # reference: sigma.view(N,1,1,1) raises here (noise.py:159)
            raise RuntimeError(f"shape '[{n}, 1, 1, 1]' is invalid for input of size {sigma.numel()}")
        if not x.is_cuda:
            raise RuntimeError("the HIP denoiser needs a tensor on the GPU; there is no CPU path")
        eng = self.engine_for(n, h, w, x.device.index)
        return eng.denoise(x.contiguous().float(), sigma.reshape(n).to(x.device, torch.float32).contiguous())

    __call__ = forward
