"""Denoiser weights: deterministic generator and state_dict ingest.

Trained weights (`evaluation/pretrained/unet-nm.pt`, reference README.md:24) are an
external download and are not available offline, so parity and benchmarks run on
seeded weights produced HERE, identically on every machine, from a counter-based hash
(no dependence on torch's or numpy's RNG streams).  The result is a plain
``{key: float32 ndarray}`` dict under the 56 key names the reference's
``UNetDenoiser2D`` loads (noise.py:146-148), OIHW like ``nn.Conv2d``.

Two initialisations:

* ``"torch_default"`` - U(-1/sqrt(fan_in), +1/sqrt(fan_in)) for weights and biases, the
  bound of ``nn.Conv2d.reset_parameters``.  Activations shrink ~3x per layer, so the
  network output is dominated by the last biases; a weak parity probe.
* ``"unit_gain"`` (default) - variance-preserving for LeakyReLU(0.2) so every layer's
  activations stay O(1) and an error in ANY layer reaches the output; the last 1x1
  conv is scaled so the residual is a few percent of the image range and the ADMM
  iteration stays well-conditioned.
"""
from __future__ import annotations

import math
from typing import Dict, Mapping

import numpy as np

from .unet_spec import LEAKY_SLOPE, UNET_LAYERS, STATE_DICT_KEYS

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser on uint64 (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def hash_uniform(seed: int, stream: int, count: int) -> np.ndarray:
    """`count` float32 in [-1, 1): element i depends only on (seed, stream, i)."""
    idx = np.arange(count, dtype=np.uint64)
    with np.errstate(over="ignore"):
        base = _splitmix64(np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream))
        bits = _splitmix64(idx ^ base)
    u24 = (bits >> np.uint64(40)).astype(np.float64)          # 24 random bits
    return (u24 * (2.0 / 16777216.0) - 1.0).astype(np.float32)


def generate_unet_weights(seed: int = 0, init: str = "unit_gain", out_scale: float = 0.05) -> Dict[str, np.ndarray]:
    sd: Dict[str, np.ndarray] = {}
    for l in UNET_LAYERS:
        fan_in = l.cin * l.ksize * l.ksize
        if init == "torch_default":
            wb = 1.0 / math.sqrt(fan_in)
            bb = 1.0 / math.sqrt(fan_in)
        elif init == "unit_gain":
            if l.ksize == 3:
                wb = math.sqrt(6.0 / ((1.0 + LEAKY_SLOPE ** 2) * fan_in))
                bb = 0.05
            else:  # outc: small linear read-out, zero bias
                wb = out_scale * math.sqrt(3.0 / fan_in)
                bb = 0.0
        else:
            raise ValueError(f"unknown init {init!r}")
        nw = l.cout * fan_in
        w = hash_uniform(seed, 2 * l.index, nw) * np.float32(wb)
        b = hash_uniform(seed, 2 * l.index + 1, l.cout) * np.float32(bb)
        sd[l.weight_key] = w.reshape(l.cout, l.cin, l.ksize, l.ksize)
        sd[l.bias_key] = b.reshape(l.cout)
    return sd


def check_state_dict(sd: Mapping[str, object]) -> Dict[str, np.ndarray]:
    """Validate keys/shapes of a reference-style state_dict; return float32 ndarrays.

    Accepts torch tensors or ndarrays.  Raises KeyError / ValueError with the offending
    key, where the reference would raise from ``load_state_dict`` (noise.py:148).
    """
    out: Dict[str, np.ndarray] = {}
    missing = [k for k in STATE_DICT_KEYS if k not in sd]
    if missing:
        raise KeyError(f"denoiser state_dict is missing keys: {missing[:4]}{'...' if len(missing) > 4 else ''}")
    unexpected = [k for k in sd.keys() if k not in STATE_DICT_KEYS]
    if unexpected:
        raise KeyError(f"denoiser state_dict has unexpected keys: {unexpected[:4]}")
    for l in UNET_LAYERS:
        for key, shape in ((l.weight_key, (l.cout, l.cin, l.ksize, l.ksize)), (l.bias_key, (l.cout,))):
            v = sd[key]
            if hasattr(v, "detach"):
                v = v.detach().cpu().numpy()
            a = np.ascontiguousarray(np.asarray(v, dtype=np.float32))
            if a.shape != shape:
                raise ValueError(f"{key}: expected shape {shape}, got {a.shape}")
            out[key] = a
    return out


def flatten_state_dict(sd: Mapping[str, np.ndarray]) -> np.ndarray:
    """Concatenate the 56 tensors (OIHW, state_dict order) into the blob
    `pnp_load_unet_weights` takes (include/pnpadmm.h)."""
    sd = check_state_dict(sd)
    return np.concatenate([sd[k].reshape(-1) for k in STATE_DICT_KEYS]).astype(np.float32)


def to_torch_state_dict(sd: Mapping[str, np.ndarray]):
    import torch
    return {k: torch.from_numpy(np.array(v, dtype=np.float32, copy=True)) for k, v in sd.items()}


def generate_policy_weights(model, seed: int = 0, t_bias: float = 0.0, head_gain: float = 1.0):
    """Deterministic stand-in for the decision-transformer checkpoints (external downloads): N(0, 0.02)-like
    weights from the counter hash for every Linear/Embedding/Conv, LayerNorm = (1, 0), biases 0 - the reference's
    `_init_weights` (decision_transformer.py:157-164) statistics - written into `model.state_dict()` order so the
    SAME tensors load into the reference's DecisionTransformer.  `t_bias` shifts the stop logit, `head_gain` scales
    the action head (random heads sit at sigmoid(0) = 0.5, exactly on the stop threshold)."""
    import torch
    sd = model.state_dict()
    out = {}
    for i, (k, v) in enumerate(sd.items()):
        if k.endswith("masking"):
            out[k] = v.clone()
            continue
        n = v.numel()
        if ".ln" in k or k.startswith("layer_n"):
            arr = np.ones(n, np.float32) if k.endswith("weight") else np.zeros(n, np.float32)
        elif k.endswith("bias"):
            arr = np.zeros(n, np.float32)
        else:
            u = hash_uniform(seed, 1000 + 2 * i, n).astype(np.float64) + hash_uniform(seed, 1001 + 2 * i, n) + \
                hash_uniform(seed, 5000 + 2 * i, n)                     # sum of 3 uniforms: var = 1
            arr = (0.02 * u).astype(np.float32)
            if k.startswith("state_encoder.0") or k.startswith("state_encoder.2") or k.startswith("state_encoder.4"):
                arr = (arr * 5.0).astype(np.float32)                    # conv stem: keep the image signal alive
        out[k] = torch.from_numpy(arr.reshape(tuple(v.shape)).copy())
    out["predict_action.0.weight"] = out["predict_action.0.weight"] * head_gain
    order = list(model.action_range.keys())
    b = out["predict_action.0.bias"].clone()
    b[order.index("T")] = t_bias
    out["predict_action.0.bias"] = b
    return out
