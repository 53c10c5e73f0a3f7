"""Greedy decision-transformer rollout: the caller of `PnPEnv.step`.

Counterpart of `Evaluator` in /root/reference/evaluation/eval.py (`get_initial_policy_setup` :62-100,
`predict_action_and_rtg` :146-186, `run_greedy` :189-220), batched: ALL N slices of a batch advance in one
`env.step` and one policy forward per step, each slice with its own (T, sigma_d, mu), context window and stop time;
a slice that has stopped (T > 0.5) keeps its action, so the engine leaves it untouched (env.py:79-81).

The reference's indexing behaviours are reproduced exactly (they decide which token the action is read from), so a
checkpoint trained for the reference produces the same action sequence here:
  * first action: model(actions=None) over the first `ctx` steps, read at position 0            (eval.py:80-86)
  * first rtg: the reference passes `eval_rtg[:, ctx]` / `eval_actions[:, ctx]` (an INDEX, not a slice), i.e.
    all-zero rtg and action tokens at every position; prediction read at position 0              (eval.py:90-98,59)
  * time < ctx: window = steps [0, ctx); action read at position `time`, rtg at position `time`  (eval.py:150-166)
  * time >= ctx: window = steps [time-ctx, time) - it EXCLUDES the step just observed; action read at the last
    position, rtg at the second-to-last                                                           (eval.py:168-184,53-60)
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Optional

import torch

from ..policy import policy_observation


@dataclass
class GreedyResult:
    reward: torch.Tensor        # [N,1] final PSNR (CPU), env.compute_reward at each slice's stop
    initial_reward: torch.Tensor  # [N,1] PSNR of x0
    stop_time: torch.Tensor     # [N] step at which each slice stopped (1..max_timesteps)
    actions: torch.Tensor       # [N, max_timesteps, 3] actions handed to the env (model order)
    x: torch.Tensor             # [N,1,H,W] final images (device)


class GreedyEvaluator:
    def __init__(self, model, env, action_dim: int = 3, max_timesteps: int = 30, block_size: int = 18,
                 device_type="cuda"):
        self.model = model.to(device_type).eval()
        self.env = env
        self.action_dim = action_dim
        self.max_timesteps = max_timesteps
        self.context_length = block_size // 3
        self.device = torch.device(device_type)

    # ---- policy calls --------------------------------------------------------------------------------------
    def _pick(self, action_dict, pred_actions, pos: int):
        picked = OrderedDict((k, v[:, pos, 0].contiguous()) for k, v in action_dict.items())     # each [N]
        return picked, pred_actions[:, pos]

    @torch.no_grad()
    def _initial(self, es, ea, er, et, ek):
        ctx = self.context_length
        pred_actions, action_dict = self.model(er[:, :ctx], es[:, :ctx], et[:, :ctx], ek[:, :ctx], actions=None)
        action, pa = self._pick(action_dict, pred_actions, 0)
        ea[:, 0] = pa
        # eval.py:90-95 hands the model the rtg/action at INDEX ctx - still all zeros at this point - which the model
        # broadcasts over every position of the window
        w = min(ctx, es.shape[1])
        zeros_r = torch.zeros_like(er[:, :w])
        zeros_a = torch.zeros_like(ea[:, :w])
        pred_rtg = self.model(zeros_r, es[:, :ctx], et[:, :ctx], ek[:, :ctx], zeros_a, eval_rtg=True)
        return action, pred_rtg[:, 0]

    @torch.no_grad()
    def _predict(self, es, ea, er, et, ek, time: int):
        ctx = self.context_length
        lo, hi = (0, ctx) if time < ctx else (time - ctx, time)
        pa_pos = time if time < ctx else -1
        rtg_pos = time if time + 1 <= ctx else -2
        pred_actions, action_dict = self.model(er[:, lo:hi], es[:, lo:hi], et[:, lo:hi], ek[:, lo:hi], ea[:, lo:hi],
                                               eval_actions=True)
        action, pa = self._pick(action_dict, pred_actions, pa_pos)
        ea[:, time] = pa
        pred_rtg = self.model(er[:, lo:hi], es[:, lo:hi], et[:, lo:hi], ek[:, lo:hi], ea[:, lo:hi], eval_rtg=True)
        return action, pred_rtg[:, rtg_pos]

    # ---- rollout ---------------------------------------------------------------------------------------------
    def buffers(self, n: int, task: torch.Tensor):
        """Zeroed context buffers (eval.py:65-70): states [n,T,16384], actions [n,T,3], rtg [n,T,1], timesteps, task."""
        dev, T = self.device, self.max_timesteps
        es = torch.zeros((n, T, 128 * 128), device=dev)
        ea = torch.zeros((n, T, self.action_dim), device=dev)
        er = torch.zeros((n, T, 1), device=dev)
        et = torch.arange(T, device=dev).reshape(1, T, 1).expand(n, -1, -1).contiguous()
        ek = task.reshape(n, 1).to(dev).expand(-1, T).contiguous()
        return es, ea, er, et, ek

    def rollout(self, states, action, pred_rtg, start_time: int, es, ea, er, et, ek, scorer=None):
        """eval.py:189-220 from `start_time`: step, observe, re-plan, until every slice stopped or max_timesteps.
        Returns (reward [N,1] CPU, stop_time [N]).  `scorer(states) -> [N]` replaces PSNR (no-reference rollouts)."""
        dev, T = self.device, self.max_timesteps
        n = states["z"].shape[0]
        stopped = torch.zeros(n, dtype=torch.bool, device=dev)
        stop_time = torch.full((n,), T, dtype=torch.int64, device=dev)
        for time in range(start_time, T + 1):
            states, done = self.env.step(states, action)
            done = torch.as_tensor(done, device=dev).reshape(-1)
            stop_time[done & ~stopped] = time
            stopped |= done
            if time == T or bool(stopped.all()):
                break
            live = ~stopped
            ob = policy_observation(states["x"])
            es[live, time] = ob[live]
            er[live, time] = pred_rtg[live]
            new_action, new_rtg = self._predict(es, ea, er, et, ek, time)
            for k in action:                                   # stopped slices keep the action that stopped them
                action[k] = torch.where(live, new_action[k], action[k])
            pred_rtg = torch.where(live.reshape(n, 1), new_rtg, pred_rtg)
        if scorer is not None:
            reward = torch.as_tensor(scorer(states)).reshape(n, 1).float().cpu()
        else:
            reward = self.env.compute_reward(states["x"], states["gt"])
        return reward, stop_time.cpu()

    def run(self, mat: Dict[str, torch.Tensor], rtg: torch.Tensor, task: torch.Tensor,
            first_state: Optional[torch.Tensor] = None) -> GreedyResult:
        """mat: collated `.mat` dict (x0, y0, ATy0, mask, gt); rtg [N] normalised return-to-go target;
        task [N] int task token; first_state [N, H*W]: the policy's first state token - default `mat['x0_raw']`, the
        UNclipped Re x0 the reference's datasets hand over (datasets.py:162,201), else the env's (clipped) Re x0."""
        dev = self.device
        states = self.env.reset(mat, dev)
        n = states["z"].shape[0]
        if first_state is None:
            first_state = torch.as_tensor(mat["x0_raw"]) if "x0_raw" in mat else states["x"]
        if first_state.is_complex():
            first_state = first_state.real
        es, ea, er, et, ek = self.buffers(n, task)
        es[:, 0] = policy_observation(first_state.to(dev).float().reshape(n, 1, *states["z"].shape[-2:]))
        er[:, 0, 0] = rtg.reshape(n).to(dev).float()
        initial_reward = self.env.compute_reward(states["x"], states["gt"])
        action, pred_rtg = self._initial(es, ea, er, et, ek)
        reward, stop_time = self.rollout(states, action, pred_rtg, 1, es, ea, er, et, ek)
        return GreedyResult(reward=reward, initial_reward=initial_reward, stop_time=stop_time,
                            actions=ea.cpu(), x=states["x"])
