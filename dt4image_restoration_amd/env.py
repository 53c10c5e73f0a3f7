"""`PnPEnv`: the reference's PnP-ADMM environment interface (/root/reference/evaluation/env.py:30-116)
served by the HIP engine.  Same method names, arity and return shapes, so a driver written against the
reference (`Evaluator.run_greedy` eval.py:189-220, `run_mcts` mcts.py:212-258) calls it unchanged:

    reset(data, device) -> OrderedDict          step(states, action_dict) -> (states, done)
    get_policy_ob(states)                       compute_reward(x, gt)       run_no_ref_reward(states)

Differences, all documented in DESIGN.md:
  * any batch N and any power-of-two H, W (the reference is hard-wired to 1 x 128 x 128, env.py:44,64,115)
  * per-slice mu / sigma_d / T (1-element tensors broadcast, as the reference's drivers pass)
  * states['x'|'z'|'u'] are persistent device tensors updated IN PLACE by `step` (the reference rebinds
    freshly allocated tensors, env.py:95-97); use `snapshot`/`restore` to keep an old state (MCTS)
  * states['x'] is float32 from reset on (the reference holds complex x0 until the first step;
    every caller reads `.real`, which is a no-op on a real tensor)
  * the ARNIQA scorer (a torch.hub network fetch, env.py:36-40) is replaced by an injectable callable
  * `step` stays a function of the `states` it is handed (as in the reference, env.py:75,88-90): the engine holds ONE set
    of pre-shifted k-space constants, and states carry the id of their episode in the private key `_episode`; when
    states of another episode (another env on the same denoiser, an older reset) are stepped, the shim re-installs that
    episode's y0 / mask from the dict first (pnp_set_kspace)
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Callable, Dict, Optional, Tuple

import torch

from .denoiser import UNetDenoiser2D
from .engine import PnPEngine, _next_episode


def _as_complex(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_complex() else torch.view_as_complex(t.contiguous().float())


class PnPEnv:
    def __init__(self, max_episode_step: int, denoiser: UNetDenoiser2D, device_type,
                 no_ref_scorer: Optional[Callable[[torch.Tensor], float]] = None, replica: int = 0) -> None:
        self.max_episode_step = max_episode_step
        self.denoiser = denoiser.to(device_type)
        self.no_ref_model = no_ref_scorer
        self._engine: Optional[PnPEngine] = None
        self._device_type = device_type
        self._replica = int(replica)

    def fork(self, replica: int) -> "PnPEnv":
        """Another env on the same denoiser weights whose engines are replicas of their own (own workspace, own k-space
        constants): two sub-batches of one job can then be stepped concurrently on two streams."""
        return PnPEnv(self.max_episode_step, self.denoiser, self._device_type, self.no_ref_model, replica=replica)

    # ---- engine management ------------------------------------------------------------------
    def _engine_for(self, n: int, h: int, w: int, device: torch.device) -> PnPEngine:
        idx = device.index if device.index is not None else torch.cuda.current_device()
        self._engine = self.denoiser.engine_for(n, h, w, idx, self._replica)
        return self._engine

    # ---- reference interface ----------------------------------------------------------------
    def reset(self, data: Dict[str, torch.Tensor], device_type) -> "OrderedDict[str, torch.Tensor]":
        """env.py:57-71.  data: collated `.mat` dict - x0, y0, ATy0 float [..,H,W,2]; mask [..,H,W]; gt."""
        device = torch.device(device_type)
        if device.type != "cuda":
            raise RuntimeError("PnPEnv runs on the GPU only (device_type='cuda'); there is no CPU path")
        x0 = _as_complex(torch.as_tensor(data["x0"]))
        h, w = x0.shape[-2:]
        n = x0.numel() // (h * w)
        x0 = x0.reshape(n, 1, h, w).to(device).contiguous()
        y0 = _as_complex(torch.as_tensor(data["y0"])).reshape(n, 1, h, w).to(device).contiguous()
        mask = torch.as_tensor(data["mask"])
        mask = mask.reshape(-1, h, w) if mask.numel() != h * w else mask.reshape(h, w)
        if mask.dim() == 3 and mask.shape[0] not in (1, n):
            raise ValueError(f"mask batch {mask.shape[0]} does not match {n} slices")
        mask = mask.to(device).to(torch.bool).contiguous()
        gt = torch.as_tensor(data["gt"]).to(device).float()
        eng = self._engine_for(n, h, w, device)
        x, z, u = eng.reset(x0, y0, mask)
        aty0 = torch.as_tensor(data["ATy0"])[..., 0] if "ATy0" in data else None
        return OrderedDict({"x": x, "y0": y0, "z": z, "u": u, "mask": mask, "gt": gt, "ATy0": aty0,
                            "T": torch.zeros(n, dtype=torch.float32, device=device),
                            "complex_y0": data["y0"], "_episode": eng.live_episode})

    def _bind_episode(self, eng: PnPEngine, states) -> None:
        """Make the engine's k-space constants those of `states` (env.py:88-90 reads y0 / mask from the dict)."""
        ep = states.get("_episode", 0)
        if ep and ep == eng.live_episode:
            return
        if not ep:                              # a state dict built by hand the reference's way
            ep = states["_episode"] = _next_episode()
        n, h, w = eng.n, eng.h, eng.w
        y0 = _as_complex(states["y0"]).reshape(n, 1, h, w).to(eng.device).contiguous()
        mask = torch.as_tensor(states["mask"]).to(eng.device)
        mask = mask.reshape(h, w) if mask.numel() == h * w else mask.reshape(n, h, w)
        eng.set_kspace(y0, mask, episode=ep)

    def _param(self, v, n: int, device) -> torch.Tensor:
        t = torch.as_tensor(v, dtype=torch.float32, device=device).reshape(-1)
        if t.numel() == 1 and n > 1:
            t = t.expand(n)
        if t.numel() != n:
            raise ValueError(f"action has {t.numel()} values for {n} slices")
        return t.contiguous()

    def step(self, states: "OrderedDict[str, torch.Tensor]", action_dict) -> Tuple["OrderedDict", object]:
        """env.py:74-100.  Slices with T > 0.5 are done and untouched.  Returns (same dict, done): a Python
        bool for N == 1 like the reference, else a bool tensor [N] (no host sync)."""
        x, z, u = states["x"], states["z"], states["u"]
        n, _, h, w = z.shape
        eng = self._engine if self._engine is not None and (self._engine.n, self._engine.h, self._engine.w) == (n, h, w) \
            else self._engine_for(n, h, w, z.device)
        dev = z.device
        T = self._param(action_dict["T"], n, dev)
        mu = self._param(action_dict["mu"], n, dev)
        sigma_d = self._param(action_dict["sigma_d"], n, dev)
        if x.is_complex():                      # a state built by hand the reference's way
            x = x.real.contiguous()
        done = torch.empty(n, dtype=torch.uint8, device=dev)
        self._bind_episode(eng, states)
        eng.step(x, z, u, mu, sigma_d, t_action=T, t_state=states["T"], done=done)
        states["x"] = x
        if n == 1:
            return states, bool(done.item())
        return states, done.bool()

    @staticmethod
    def get_policy_ob(state) -> torch.Tensor:
        """env.py:102-109: Re(x) flattened, one row per slice."""
        x = state["x"].real
        return x.reshape(x.shape[0], -1)

    def compute_reward(self, x: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
        """env.py:112-125: PSNR of clamp(Re x, 0, 1) against gt, [N,1] on the CPU like the reference."""
        gt = gt.to(x.device).float()
        h, w = gt.shape[-2:]
        n = gt.numel() // (h * w)
        xr = (x.real if x.is_complex() else x).float().reshape(n, 1, h, w).contiguous()
        eng = self._engine if self._engine is not None and (self._engine.n, self._engine.h, self._engine.w) == (n, h, w) \
            else self._engine_for(n, h, w, x.device)
        return eng.psnr(xr, gt.reshape(n, 1, h, w).contiguous()).reshape(n, 1).cpu()

    def run_no_ref_reward(self, state) -> float:
        """env.py:42-54 scored with ARNIQA fetched from the network; here an injected callable
        scorer(x[N,1,H,W]) -> float."""
        if self.no_ref_model is None:
            raise RuntimeError("no-reference scorer not configured: pass no_ref_scorer= to PnPEnv "
                               "(the reference fetches ARNIQA with torch.hub, unavailable offline)")
        return float(self.no_ref_model(state["x"]))

    # ---- explicit state copies (the reference rebinds tensors; this engine updates in place) ----
    def snapshot(self, states) -> Dict[str, torch.Tensor]:
        """A node's copy of the iterate.  Whole-batch states of the live engine go through pnp_snapshot into one packed
        device buffer; anything else (row slices, states of another size) is cloned tensor by tensor."""
        e = self._engine
        if (e is not None and states["x"].numel() == e.n * e.h * e.w and states["T"].numel() == e.n
                and states["T"].dtype == torch.float32 and states["T"].is_contiguous()):
            return {"packed": e.snapshot(states["x"], states["z"], states["u"], states["T"])}
        return {k: states[k].clone() for k in ("x", "z", "u", "T")}

    def restore(self, states, snap) -> None:
        if "packed" in snap:
            self._engine.restore(snap["packed"], states["x"], states["z"], states["u"], states["T"])
            return
        for k in ("x", "z", "u", "T"):
            states[k].copy_(snap[k])
