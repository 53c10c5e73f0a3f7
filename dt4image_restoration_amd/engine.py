"""PnPEngine: one handle of libpnpadmm.so bound to torch-ROCm tensors.

torch is used only for device memory and streams; every computation on the hot path is a
HIP kernel behind the C ABI (include/pnpadmm.h)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Mapping, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .weights import flatten_state_dict


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


_episode_counter = 0


def _next_episode() -> int:
    """Process-wide id of one `reset` (one set of k-space constants): states carry it, engines remember the live one."""
    global _episode_counter
    _episode_counter += 1
    return _episode_counter


class PnPEngine:
    """Owns the workspace for N slices of H x W on one GPU.  Not thread-safe; one per process/GPU."""

    def __init__(self, n: int, h: int, w: int, device: Optional[int] = None, profile: bool = False,
                 denoiser: bool = True, keep_stages: bool = False, bf16_convs: bool = False, profile_layers: bool = False):
        if not torch.cuda.is_available():
            raise _lib.PnPError("PnPEngine needs a ROCm GPU (torch.cuda.is_available() is False); no CPU fallback")
        self.lib = _lib.load()
        self.n, self.h, self.w = int(n), int(h), int(w)
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)
        flags = ((_lib.PNP_FLAG_PROFILE if profile else 0) | (0 if denoiser else _lib.PNP_FLAG_NO_DENOISER)
                 | (_lib.PNP_FLAG_KEEP_STAGES if keep_stages else 0) | (_lib.PNP_FLAG_BF16_CONVS if bf16_convs else 0)
                 | (_lib.PNP_FLAG_PROFILE_LAYERS if profile_layers else 0))
        cfg = _lib.pnp_config(self.n, self.h, self.w, self.device_index, flags)
        hnd = C.c_void_p()
        _lib.check(self.lib.pnp_create(C.byref(cfg), C.byref(hnd)), "pnp_create")
        self._h = hnd
        self.profile = profile or profile_layers
        self.bf16_convs = bool(bf16_convs)
        self.live_episode = 0        # id of the reset whose y0 / mask the engine currently holds (0: none)

    def _stream(self) -> int:
        """The caller's current stream ON THE ENGINE'S DEVICE (not on torch's current device)."""
        return torch.cuda.current_stream(self.device).cuda_stream

    def close(self):
        if getattr(self, "_h", None):
            self.lib.pnp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers ---------------------------------------------------------------------------
    def _chk(self, t: torch.Tensor, dtype, numel: int, name: str) -> torch.Tensor:
        if t.device != self.device:
            raise ValueError(f"{name}: expected a tensor on {self.device}, got {t.device}")
        if t.dtype != dtype:
            raise ValueError(f"{name}: expected dtype {dtype}, got {t.dtype}")
        if t.numel() != numel:
            raise ValueError(f"{name}: expected {numel} elements, got {tuple(t.shape)}")
        if not t.is_contiguous():
            raise ValueError(f"{name}: tensor must be contiguous")
        return t

    @property
    def workspace_bytes(self) -> int:
        return int(self.lib.pnp_workspace_bytes(self._h))

    # -- weights ---------------------------------------------------------------------------
    def load_weights(self, state_dict: Mapping[str, object]) -> None:
        """state_dict with the reference's 56 keys (evaluation/noise.py:146-148); tensors or ndarrays."""
        blob = np.ascontiguousarray(flatten_state_dict(state_dict))
        _lib.check(self.lib.pnp_load_unet_weights(self._h, blob.ctypes.data, blob.size), "pnp_load_unet_weights")

    # -- hot path --------------------------------------------------------------------------
    def reset(self, x0: torch.Tensor, y0: torch.Tensor, mask: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """x0, y0 complex64 [N,1,H,W]; mask bool/uint8 [H,W] (or [N,H,W]).  Returns fresh (x f32, z c64, u c64)."""
        nhw = self.n * self.h * self.w
        x0 = self._chk(x0, torch.complex64, nhw, "x0")
        y0 = self._chk(y0, torch.complex64, nhw, "y0")
        m = mask.to(torch.uint8).contiguous()
        if m.numel() == self.h * self.w:
            mask_n = 1
        elif m.numel() == nhw:
            mask_n = self.n
        else:
            raise ValueError(f"mask: expected {self.h * self.w} or {nhw} elements, got {tuple(mask.shape)}")
        self._chk(m, torch.uint8, m.numel(), "mask")
        x = torch.empty((self.n, 1, self.h, self.w), dtype=torch.float32, device=self.device)
        z = torch.empty((self.n, 1, self.h, self.w), dtype=torch.complex64, device=self.device)
        u = torch.empty_like(z)
        _lib.check(self.lib.pnp_reset(self._h, x0.data_ptr(), y0.data_ptr(), m.data_ptr(), mask_n, x.data_ptr(),
                                      z.data_ptr(), u.data_ptr(), self._stream()), "pnp_reset")
        self.live_episode = _next_episode()
        return x, z, u

    def set_kspace(self, y0: torch.Tensor, mask: torch.Tensor, episode: int = 0) -> None:
        """Re-install the k-space constants (y0, mask) of another episode without touching any iterate
        (pnp_set_kspace); `episode` = the id that episode's reset returned (0: a fresh id)."""
        nhw = self.n * self.h * self.w
        y0 = self._chk(y0, torch.complex64, nhw, "y0")
        m = mask.to(torch.uint8).contiguous()
        if m.numel() not in (self.h * self.w, nhw):
            raise ValueError(f"mask: expected {self.h * self.w} or {nhw} elements, got {tuple(mask.shape)}")
        self._chk(m, torch.uint8, m.numel(), "mask")
        _lib.check(self.lib.pnp_set_kspace(self._h, y0.data_ptr(), m.data_ptr(), 1 if m.numel() == self.h * self.w else self.n,
                                           self._stream()), "pnp_set_kspace")
        self.live_episode = episode or _next_episode()

    def step(self, x: torch.Tensor, z: torch.Tensor, u: torch.Tensor, mu: torch.Tensor, sigma_d: torch.Tensor,
             t_action: Optional[torch.Tensor] = None, t_state: Optional[torch.Tensor] = None,
             done: Optional[torch.Tensor] = None) -> None:
        """One ADMM iteration in place on (x, z, u).  mu, sigma_d, t_action, t_state: float32 [N]; done: uint8 [N]."""
        nhw = self.n * self.h * self.w
        self._chk(x, torch.float32, nhw, "x"); self._chk(z, torch.complex64, nhw, "z"); self._chk(u, torch.complex64, nhw, "u")
        self._chk(mu, torch.float32, self.n, "mu"); self._chk(sigma_d, torch.float32, self.n, "sigma_d")
        if t_action is not None: self._chk(t_action, torch.float32, self.n, "t_action")
        if t_state is not None: self._chk(t_state, torch.float32, self.n, "t_state")
        if done is not None: self._chk(done, torch.uint8, self.n, "done")
        _lib.check(self.lib.pnp_step(self._h, mu.data_ptr(), sigma_d.data_ptr(), _ptr(t_action), x.data_ptr(),
                                     z.data_ptr(), u.data_ptr(), _ptr(t_state), _ptr(done), self._stream()), "pnp_step")

    def denoise(self, x: torch.Tensor, sigma: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        nhw = self.n * self.h * self.w
        self._chk(x, torch.float32, nhw, "x"); self._chk(sigma, torch.float32, self.n, "sigma")
        if out is None:
            out = torch.empty_like(x)
        self._chk(out, torch.float32, nhw, "out")
        _lib.check(self.lib.pnp_denoise(self._h, x.data_ptr(), sigma.data_ptr(), out.data_ptr(), self._stream()), "pnp_denoise")
        return out

    def fft2c(self, img: torch.Tensor, inverse: bool = False) -> torch.Tensor:
        if img.dtype != torch.complex64 or img.shape[-2:] != (self.h, self.w):
            raise ValueError(f"fft2c: expected complex64 [...,{self.h},{self.w}], got {img.dtype} {tuple(img.shape)}")
        batch = img.numel() // (self.h * self.w)
        img = self._chk(img, torch.complex64, batch * self.h * self.w, "img")
        out = torch.empty_like(img)
        _lib.check(self.lib.pnp_fft2c(self._h, img.data_ptr(), out.data_ptr(), batch, self.h, self.w, int(inverse),
                                      self._stream()), "pnp_fft2c")
        return out

    def prox_dual(self, x, z, u, mu, t_action=None) -> None:
        nhw = self.n * self.h * self.w
        self._chk(x, torch.float32, nhw, "x"); self._chk(z, torch.complex64, nhw, "z"); self._chk(u, torch.complex64, nhw, "u")
        self._chk(mu, torch.float32, self.n, "mu")
        _lib.check(self.lib.pnp_prox_dual(self._h, mu.data_ptr(), _ptr(t_action), x.data_ptr(), z.data_ptr(),
                                          u.data_ptr(), self._stream()), "pnp_prox_dual")

    def psnr(self, x: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
        nhw = self.n * self.h * self.w
        self._chk(x, torch.float32, nhw, "x"); self._chk(gt, torch.float32, nhw, "gt")
        out = torch.empty(self.n, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.pnp_psnr(self._h, x.data_ptr(), gt.data_ptr(), out.data_ptr(), self._stream()), "pnp_psnr")
        return out

    def snapshot(self, x: torch.Tensor, z: torch.Tensor, u: torch.Tensor, t_state: Optional[torch.Tensor] = None,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One packed device buffer [x | z | u | T] (pnp_snapshot): a tree-search node's copy of the iterate."""
        nhw = self.n * self.h * self.w
        self._chk(x, torch.float32, nhw, "x"); self._chk(z, torch.complex64, nhw, "z"); self._chk(u, torch.complex64, nhw, "u")
        if t_state is not None:
            self._chk(t_state, torch.float32, self.n, "t_state")
        nbytes = self.lib.pnp_snapshot_bytes(self._h)
        if out is None:
            out = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        elif out.numel() * out.element_size() != nbytes or not out.is_contiguous() or out.device != self.device:
            raise ValueError(f"snapshot buffer must be {nbytes} contiguous bytes on {self.device}")
        _lib.check(self.lib.pnp_snapshot(self._h, x.data_ptr(), z.data_ptr(), u.data_ptr(), _ptr(t_state), out.data_ptr(),
                                         self._stream()), "pnp_snapshot")
        return out

    def restore(self, snap: torch.Tensor, x: torch.Tensor, z: torch.Tensor, u: torch.Tensor,
                t_state: Optional[torch.Tensor] = None) -> None:
        nhw = self.n * self.h * self.w
        self._chk(x, torch.float32, nhw, "x"); self._chk(z, torch.complex64, nhw, "z"); self._chk(u, torch.complex64, nhw, "u")
        if t_state is not None:
            self._chk(t_state, torch.float32, self.n, "t_state")
        if snap.numel() * snap.element_size() != self.lib.pnp_snapshot_bytes(self._h) or not snap.is_contiguous():
            raise ValueError("not a snapshot of this engine")
        _lib.check(self.lib.pnp_restore(self._h, snap.data_ptr(), x.data_ptr(), z.data_ptr(), u.data_ptr(), _ptr(t_state),
                                        self._stream()), "pnp_restore")

    def read_stage(self, which: int) -> torch.Tensor:
        c, hh, ww = C.c_int(), C.c_int(), C.c_int()
        _lib.check(self.lib.pnp_unet_read_stage(self._h, which, None, C.byref(c), C.byref(hh), C.byref(ww), self._stream()),
                   "pnp_unet_read_stage")
        out = torch.empty((self.n, c.value, hh.value, ww.value), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.pnp_unet_read_stage(self._h, which, out.data_ptr(), None, None, None, self._stream()),
                   "pnp_unet_read_stage")
        return out

    def conv_algorithms(self):
        """Per conv layer: 0 direct MFMA, 1 Winograd F(2x2), 4 Winograd F(4x4), 2 first layer (VALU), 3 last layer,
        5 bf16 producer/consumer kernel (bf16 mode, chip-filling problems)."""
        out = (C.c_int32 * _lib.N_LAYERS)()
        _lib.check(self.lib.pnp_conv_algorithms(self._h, out), "pnp_conv_algorithms")
        return list(out)

    def bf16_weight_terms(self) -> int:
        """bf16 terms per conv weight: 0 on an f32 handle, 2 in bf16 mode (hi + lo), 1 with PNP_BF16_W1 (ablation)."""
        return int(self.lib.pnp_bf16_weight_terms(self._h))

    # -- kernel timing ---------------------------------------------------------------------
    def profile_reset(self) -> None:
        _lib.check(self.lib.pnp_profile_reset(self._h), "pnp_profile_reset")

    def profile_collect(self) -> Dict[str, Dict[str, float]]:
        """Call after synchronising the stream.  {class: {ms, launches}} plus per-layer lists."""
        ms = (C.c_double * _lib.PROFILE_CLASSES)()
        cnt = (C.c_int64 * _lib.PROFILE_CLASSES)()
        _lib.check(self.lib.pnp_profile_collect(self._h, ms, cnt), "pnp_profile_collect")
        lms = (C.c_double * _lib.N_LAYERS)()
        lcnt = (C.c_int64 * _lib.N_LAYERS)()
        _lib.check(self.lib.pnp_profile_layers(self._h, lms, lcnt), "pnp_profile_layers")
        out = {name: {"ms": ms[i], "launches": int(cnt[i])} for i, name in enumerate(_lib.PROFILE_CLASS_NAMES)}
        out["layers"] = {"ms": list(lms), "launches": [int(v) for v in lcnt]}
        return out
