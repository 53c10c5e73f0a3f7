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

Policy-side cost (the reference re-encodes the whole 6-image context window in BOTH forwards of every step): the state
encoder is a pure per-image function, so each observation is encoded ONCE when it is written into the context
(`PolicyContext.ee`) and both forwards read the cached embeddings - 1 encoder image per slice and step instead of 12.
`sync_every` spaces out the only host synchronisation of the loop (the all-stopped check).

`use_graphs` (default on a GPU): the policy side of a steady-state step - the observation's state encoder, and the ONE transformer
forward with both heads over the 6-step window (from step `ctx` on the reference's two forwards read identical tokens, see
`_predict`) - is ~110 kernels of a few microseconds each; the encoder call and that forward are captured ONCE per batch size in
two hipGraphs (`torch.cuda.CUDAGraph`) over static window buffers and replayed (same kernels, same arithmetic; the first `ctx`
steps of an episode, whose read positions move and which still need two forwards, stay eager).

`rollout_rows` is the same loop with a per-row clock (rows of one batch at different episode times), which is what a
batched tree search needs: the nodes selected in different images sit at different depths.
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


@dataclass
class PolicyContext:
    """Context buffers of the policy (eval.py:65-70) for n rows."""
    es: torch.Tensor            # [n, T, 16384] observations
    ea: torch.Tensor            # [n, T, 3] actions (model order)
    er: torch.Tensor            # [n, T, 1] return-to-go tokens
    et: torch.Tensor            # [n, T, 1] time steps
    ek: torch.Tensor            # [n, T] task token
    ee: Optional[torch.Tensor]  # [n, T, E] cached state-encoder outputs of `es` (None: re-encode like the reference)
    tag: int = 0                # which concurrently running sub-batch this context belongs to (its own captured graphs / static buffers)

    def window(self, lo: int, hi: int):
        return (self.er[:, lo:hi], self.es[:, lo:hi], self.et[:, lo:hi], self.ek[:, lo:hi], self.ea[:, lo:hi],
                None if self.ee is None else self.ee[:, lo:hi])


class GreedyEvaluator:
    def __init__(self, model, env, action_dim: int = 3, max_timesteps: int = 30, block_size: int = 18,
                 device_type="cuda", cache_state_embeddings: bool = True, sync_every: int = 1,
                 use_graphs: Optional[bool] = None):
        self.model = model.to(device_type).eval()
        self.env = env
        self.action_dim = action_dim
        horizon = getattr(getattr(model, "time_embed", None), "num_embeddings", None)
        if horizon is not None and max_timesteps > horizon:
            # (on the GPU an out-of-range time step would be a device-side assert in the embedding lookup: the process aborts)
            raise ValueError(f"max_timesteps={max_timesteps} exceeds the policy's time embedding ({horizon} steps, "
                             f"decision_transformer.py:283 max_timestep)")
        self.max_timesteps = max_timesteps
        self.context_length = block_size // 3
        self.device = torch.device(device_type)
        self.cache_state_embeddings = cache_state_embeddings
        self.sync_every = max(1, int(sync_every))
        # hipGraph replay of the policy calls: GPU only, and only with cached state embeddings (the window is then five small tensors)
        self.use_graphs = (self.device.type == "cuda" and cache_state_embeddings) if use_graphs is None else \
            (bool(use_graphs) and self.device.type == "cuda" and cache_state_embeddings)
        self._graphs = {}

    # ---- context ---------------------------------------------------------------------------------------------
    @torch.no_grad()
    def buffers(self, n: int, task: torch.Tensor) -> PolicyContext:
        """Zeroed context buffers (eval.py:65-70): states [n,T,16384], actions [n,T,3], rtg [n,T,1], timesteps, task."""
        dev, T = self.device, self.max_timesteps
        es = torch.zeros((n, T, 128 * 128), device=dev)
        ea = torch.zeros((n, T, self.action_dim), device=dev)
        er = torch.zeros((n, T, 1), device=dev)
        et = torch.arange(T, device=dev).reshape(1, T, 1).expand(n, -1, -1).contiguous()
        ek = task.reshape(n, 1).to(dev).expand(-1, T).contiguous()
        ee = None
        if self.cache_state_embeddings:
            z = self.model.encode_states(torch.zeros((1, 128 * 128), device=dev))      # a not-yet-observed (all-zero) state
            ee = z.reshape(1, 1, -1).expand(n, T, -1).contiguous()
        return PolicyContext(es, ea, er, et, ek, ee)

    @torch.no_grad()
    def observe(self, ctx: PolicyContext, time, ob: torch.Tensor, rows: Optional[torch.Tensor] = None,
                rtg: Optional[torch.Tensor] = None, emb: Optional[torch.Tensor] = None) -> None:
        """Write observation `ob` [n,16384] (and, if given, the return-to-go token `rtg` [n,1]) at step `time` (int, or a
        per-row int64 tensor) for `rows` (bool mask or None = all).  Masked writes are selects, not boolean-mask indexing:
        no `nonzero`, so no host synchronisation in the rollout loop."""
        if emb is None:
            emb = self.model.encode_states(ob) if ctx.ee is not None else None
        n = ob.shape[0]
        if isinstance(time, int):
            sel = None if rows is None else rows.reshape(n, 1)
            ctx.es[:, time] = ob if sel is None else torch.where(sel, ob, ctx.es[:, time])
            if emb is not None:
                ctx.ee[:, time] = emb if sel is None else torch.where(sel, emb, ctx.ee[:, time])
            if rtg is not None:
                ctx.er[:, time] = rtg if sel is None else torch.where(sel, rtg, ctx.er[:, time])
            return
        idx = torch.arange(n, device=ob.device)
        sel = None if rows is None else rows.reshape(n, 1)
        ctx.es[idx, time] = ob if sel is None else torch.where(sel, ob, ctx.es[idx, time])
        if emb is not None:
            ctx.ee[idx, time] = emb if sel is None else torch.where(sel, emb, ctx.ee[idx, time])
        if rtg is not None:
            ctx.er[idx, time] = rtg if sel is None else torch.where(sel, rtg, ctx.er[idx, time])

    # ---- hipGraph capture of the policy side ---------------------------------------------------------------
    def _capture(self, fn):
        """Warm `fn` up on a side stream (library handles, autotuning, allocator), then capture one call of it.  The warm-up
        calls run OUTSIDE any handler and are synchronised: a genuine HIP error (device-side assert, out of memory, launch
        failure) propagates from here as what it is.  Only the capture step itself is guarded: a capture that fails (a PyTorch
        build or op that cannot be captured) switches the evaluator back to eager policy calls - the same kernels launched one
        by one - with a warning, after a synchronize that re-raises if the context did not survive; the HIP env is not
        involved either way."""
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(3):
                fn()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)                     # warm-up errors surface here, not as "capture failed"
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = fn()
            return g, out
        except RuntimeError as exc:                             # capture refused: run eager
            import warnings
            warnings.warn(f"hipGraph capture of the policy failed ({type(exc).__name__}: {exc}); continuing with eager policy calls")
            self.use_graphs = False
            torch.cuda.synchronize(self.device)                 # raises if the failure left the context unusable
            return None, None

    @torch.no_grad()
    def _graphed_encode(self, ob_src: torch.Tensor, tag: int = 0):
        """`encode_states(policy_observation(x))` through a captured graph; x [n,1,H,W] float32 on the device."""
        key = ("enc", tuple(ob_src.shape), tag)
        ent = self._graphs.get(key)
        if ent is None:
            xs = torch.zeros_like(ob_src)
            g, out = self._capture(lambda: (lambda ob: (ob, self.model.encode_states(ob)))(policy_observation(xs)))
            if g is None:
                ob = policy_observation(ob_src)
                return ob, self.model.encode_states(ob)
            ent = self._graphs[key] = (g, xs, out)
        g, xs, out = ent
        xs.copy_(ob_src)
        g.replay()
        return out                                          # (ob [n,16384], emb [n,E]): static tensors, consumed before the next replay

    @torch.no_grad()
    def _graphed_predict(self, ctx: PolicyContext, time: int):
        """`_predict` for time >= ctx (window [time-ctx, time), action read at the last position, rtg at the second-to-last)."""
        c = self.context_length
        n = ctx.ea.shape[0]
        key = ("predict", n, ctx.tag)
        ent = self._graphs.get(key)
        if ent is None:
            er, _, et, ek, ea, ee = (t.clone() if t is not None else None for t in ctx.window(0, c))

            def body():
                both, action_dict = self.model(er, ee, et, ek, ea, state_emb=ee)          # one forward, both heads (see _predict)
                action = OrderedDict((k, v[:, -1, 0].contiguous()) for k, v in action_dict.items())
                return action, both[:, -1, :self.action_dim].contiguous(), both[:, -2, self.action_dim:].contiguous()
            g, out = self._capture(body)
            if g is None:
                return self._predict(ctx, time)                # use_graphs is off now: the eager path
            ent = self._graphs[key] = (g, (er, et, ek, ea, ee), out)
        g, (er, et, ek, ea, ee), (action, pa, rtg) = ent
        lo, hi = time - c, time
        er.copy_(ctx.er[:, lo:hi]); et.copy_(ctx.et[:, lo:hi]); ek.copy_(ctx.ek[:, lo:hi])
        ea.copy_(ctx.ea[:, lo:hi]); ee.copy_(ctx.ee[:, lo:hi])
        g.replay()
        ctx.ea[:, time] = pa                                # outside the window (time >= ctx): the second forward did not need it
        return OrderedDict((k, v.clone()) for k, v in action.items()), rtg.clone()

    # ---- policy calls --------------------------------------------------------------------------------------
    def _pick(self, action_dict, pred_actions, pos: int):
        picked = OrderedDict((k, v[:, pos, 0].contiguous()) for k, v in action_dict.items())     # each [N]
        return picked, pred_actions[:, pos]

    @torch.no_grad()
    def _initial(self, ctx: PolicyContext):
        c = self.context_length
        er, es, et, ek, ea, ee = ctx.window(0, c)
        pred_actions, action_dict = self.model(er, es, et, ek, actions=None, state_emb=ee)
        action, pa = self._pick(action_dict, pred_actions, 0)
        ctx.ea[:, 0] = pa
        # eval.py:90-95 hands the model the rtg/action at INDEX ctx - still all zeros at this point - which the model
        # broadcasts over every position of the window
        pred_rtg = self.model(torch.zeros_like(er), es, et, ek, torch.zeros_like(ea), eval_rtg=True, state_emb=ee)
        return action, pred_rtg[:, 0]

    @torch.no_grad()
    def _predict(self, ctx: PolicyContext, time: int):
        c = self.context_length
        if self.use_graphs and time >= c and ctx.ee is not None:
            return self._graphed_predict(ctx, time)
        lo, hi = (0, c) if time < c else (time - c, time)
        pa_pos = time if time < c else -1
        rtg_pos = time if time + 1 <= c else -2
        er, es, et, ek, ea, ee = ctx.window(lo, hi)
        if time >= c:
            # The reference's second forward (eval.py:176-184) sees the action it has just written at step `time` - which lies
            # OUTSIDE the window [time-ctx, time): both forwards read identical tokens, so one forward with both heads gives
            # the same action and return-to-go (the same tensors through the same kernels, bit for bit).
            both, action_dict = self.model(er, es, et, ek, ea, state_emb=ee)
            action, pa = self._pick(action_dict, both[..., :self.action_dim], pa_pos)
            ctx.ea[:, time] = pa
            return action, both[:, rtg_pos, self.action_dim:]
        pred_actions, action_dict = self.model(er, es, et, ek, ea, eval_actions=True, state_emb=ee)
        action, pa = self._pick(action_dict, pred_actions, pa_pos)
        ctx.ea[:, time] = pa                                   # inside the window while time < ctx (a view: `ea` sees it)
        pred_rtg = self.model(er, es, et, ek, ea, eval_rtg=True, state_emb=ee)
        return action, pred_rtg[:, rtg_pos]

    @torch.no_grad()
    def _predict_rows(self, ctx: PolicyContext, tvec: torch.Tensor, single_forward: bool = False):
        """`_predict` with a per-row time `tvec` [n] (int64, device): every row gets the window, read positions and write
        position its own clock implies.  Rows whose clock has run past the last step are clamped (their result is unused).
        single_forward: the caller knows every row that matters sits at time >= ctx, where the reference's two forwards read
        identical tokens (see `_predict`): one forward, both heads."""
        c, T = self.context_length, self.max_timesteps
        n = tvec.shape[0]
        t = tvec.clamp(max=T - 1)
        early = t < c
        lo = torch.where(early, torch.zeros_like(t), t - c)
        # [n, c] window steps; an episode shorter than the context window (T < c) has the T-step window the scalar path's
        # slice `[0:c]` of a T-step buffer yields (every row is `early` then)
        idx = lo[:, None] + torch.arange(min(c, T), device=t.device)[None, :]
        pos_a = torch.where(early, t, torch.full_like(t, c - 1))
        pos_r = torch.where(early, t, torch.full_like(t, c - 2))
        rows = torch.arange(n, device=t.device)

        def win(buf):
            if buf.dim() == 2:
                return buf.gather(1, idx)
            return buf.gather(1, idx[:, :, None].expand(-1, -1, buf.shape[2]))

        er, et, ek = win(ctx.er), win(ctx.et), win(ctx.ek)
        ee = win(ctx.ee) if ctx.ee is not None else None
        es = win(ctx.es) if ctx.ee is None else ee              # with cached embeddings `states` is only read for its shape
        ea = win(ctx.ea)
        if single_forward:
            both, action_dict = self.model(er, es, et, ek, ea, state_emb=ee)
            action = OrderedDict((k, v[rows, pos_a, 0].contiguous()) for k, v in action_dict.items())
            ctx.ea[rows, t] = both[rows, pos_a, :self.action_dim]
            return action, both[rows, pos_r, self.action_dim:]
        pred_actions, action_dict = self.model(er, es, et, ek, ea, eval_actions=True, state_emb=ee)
        action = OrderedDict((k, v[rows, pos_a, 0].contiguous()) for k, v in action_dict.items())
        pa = pred_actions[rows, pos_a]
        ctx.ea[rows, t] = pa
        ea = win(ctx.ea)                                        # the write is inside the window while t < ctx
        pred_rtg = self.model(er, es, et, ek, ea, eval_rtg=True, state_emb=ee)
        return action, pred_rtg[rows, pos_r]

    # ---- rollout ---------------------------------------------------------------------------------------------
    def rollout(self, states, action, pred_rtg, start_time: int, ctx: PolicyContext, scorer=None):
        """eval.py:189-220 from `start_time`: step, observe, re-plan, until every slice stopped or max_timesteps.
        Returns (reward [N,1] CPU, stop_time [N]).  `scorer(states) -> [N]` replaces PSNR (no-reference rollouts)."""
        out: Dict[str, torch.Tensor] = {}
        for _ in self._rollout_phases(states, action, pred_rtg, start_time, ctx, scorer, self.env, out):
            pass
        return out["reward"], out["stop_time"]

    def _rollout_phases(self, states, action, pred_rtg, start_time: int, ctx: PolicyContext, scorer, env, out):
        """`rollout` as a generator that yields after each phase it has ENQUEUED - "step" (the env step) and "policy" (observe
        + re-plan) - so that a caller can interleave several sub-batches on several streams (`run_pipelined`).  Results go
        into `out` (reward, stop_time)."""
        dev, T = self.device, self.max_timesteps
        n = states["z"].shape[0]
        stopped = torch.zeros(n, dtype=torch.bool, device=dev)
        stop_time = torch.full((n,), T, dtype=torch.int64, device=dev)
        for time in range(start_time, T + 1):
            states, done = env.step(states, action)
            done = torch.as_tensor(done, device=dev).reshape(-1)
            stop_time = torch.where(done & ~stopped, torch.full_like(stop_time, time), stop_time)
            stopped = stopped | done
            yield "step"
            if time == T:
                break
            if (time - start_time) % self.sync_every == 0 and bool(stopped.all()):      # the loop's only host sync
                break
            live = ~stopped
            x = states["x"]
            if self.use_graphs and ctx.ee is not None and x.dim() == 4 and x.dtype == torch.float32:
                ob, emb = self._graphed_encode(x, ctx.tag)
                self.observe(ctx, time, ob, live, rtg=pred_rtg, emb=emb)
            else:
                self.observe(ctx, time, policy_observation(x), live, rtg=pred_rtg)
            new_action, new_rtg = self._predict(ctx, time)
            for k in action:                                   # stopped slices keep the action that stopped them
                action[k] = torch.where(live, new_action[k], action[k])
            pred_rtg = torch.where(live.reshape(n, 1), new_rtg, pred_rtg)
            yield "policy"
        if scorer is not None:
            out["reward"] = torch.as_tensor(scorer(states)).reshape(n, 1).float().cpu()
        else:
            out["reward"] = env.compute_reward(states["x"], states["gt"])
        out["stop_time"] = stop_time.cpu()

    def rollout_rows(self, states, action, pred_rtg, start_times: torch.Tensor, ctx: PolicyContext, scorer=None,
                     active: Optional[torch.Tensor] = None):
        """`rollout` with a per-row clock: row i takes its first step at time start_times[i] (>= 1) and runs until it stops
        or its clock passes max_timesteps.  `active` [N] bool: rows that take part at all (others are never stepped).
        Returns (reward [N,1] CPU, stop_time [N] CPU)."""
        dev, T = self.device, self.max_timesteps
        n = states["z"].shape[0]
        clock = start_times.to(dev).to(torch.int64).clone()
        stopped = torch.zeros(n, dtype=torch.bool, device=dev) if active is None else ~active.to(dev)
        stopped = stopped | (clock > T)
        stop_time = torch.full((n,), T, dtype=torch.int64, device=dev)
        one = torch.ones((), dtype=torch.float32, device=dev)
        steps = int(T + 1 - int(clock.min())) if n else 0
        # earliest clock among the rows that take part (one host read, beside the one above): live rows advance by one per
        # iteration, so from iteration ctx - first on every row that matters is past the warm-up window
        first = int(torch.where(stopped, torch.full_like(clock, T + 1), clock).min()) if n else T + 1
        for it in range(max(steps, 0)):
            # rows that are out of the game are handed a stop action: the engine leaves them untouched (env.py:79-81)
            act = OrderedDict((k, (torch.where(stopped, one, v) if k == "T" else v)) for k, v in action.items())
            states, done = self.env.step(states, act)
            done = torch.as_tensor(done, device=dev).reshape(-1)
            newly = done & ~stopped
            stop_time = torch.where(newly, clock, stop_time)
            stopped = stopped | done | (clock >= T)             # a row whose clock shows T has taken its last step
            if it % self.sync_every == 0 and bool(stopped.all()):
                break
            live = ~stopped
            self.observe(ctx, clock.clamp(max=T - 1), policy_observation(states["x"]), live, rtg=pred_rtg)
            new_action, new_rtg = self._predict_rows(ctx, clock, single_forward=first + it >= self.context_length)
            for k in action:
                action[k] = torch.where(live, new_action[k], action[k])
            pred_rtg = torch.where(live.reshape(n, 1), new_rtg, pred_rtg)
            clock = clock + live.to(torch.int64)
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
        out: Dict[str, object] = {}
        for _ in self._run_phases(mat, rtg, task, first_state, self.env, 0, out):
            pass
        return out["result"]

    def _run_phases(self, mat, rtg, task, first_state, env, tag: int, out):
        dev = self.device
        states = env.reset(mat, dev)
        n = states["z"].shape[0]
        if first_state is None:
            first_state = torch.as_tensor(mat["x0_raw"]) if "x0_raw" in mat else states["x"]
        if first_state.is_complex():
            first_state = first_state.real
        ctx = self.buffers(n, task)
        ctx.tag = tag
        self.observe(ctx, 0, policy_observation(first_state.to(dev).float().reshape(n, 1, *states["z"].shape[-2:])))
        ctx.er[:, 0, 0] = rtg.reshape(n).to(dev).float()
        initial_reward = env.compute_reward(states["x"], states["gt"])
        action, pred_rtg = self._initial(ctx)
        yield "policy"
        ro: Dict[str, torch.Tensor] = {}
        yield from self._rollout_phases(states, action, pred_rtg, 1, ctx, None, env, ro)
        out["result"] = GreedyResult(reward=ro["reward"], initial_reward=initial_reward, stop_time=ro["stop_time"],
                                     actions=ctx.ea.cpu(), x=states["x"])

    def run_pipelined(self, mat: Dict[str, torch.Tensor], rtg: torch.Tensor, task: torch.Tensor, parts: int = 2) -> GreedyResult:
        """`run` with the batch cut into `parts` contiguous sub-batches that advance on their own HIP streams, half a period
        apart: a step of the episode is a chain  env.step -> policy -> env.step ...  in which the policy's ~110 kernels of a few
        microseconds each leave the GPU almost idle (1.2-1.4 of 9.2 ms per step at 64 x 256 x 256); with two sub-batches one's
        policy call runs UNDER the other's env step.  Each sub-batch has its own engine replica (`PnPEnv.fork`: own workspace and
        k-space constants), policy context and captured graphs; slices are independent (env.py:74-100), so the result is the
        unpipelined one up to the f32 summation order of another tile plan (a sub-batch of 32 is planned unlike a batch of 64).
        MEASURED (MI355X, 64 x 256 x 256, 30 steps, bench.py --mode greedy --pipeline 2): 9.64 ms per step against 9.39 on one
        stream - the policy's dependent chain of small kernels stretches when each of them has to be dispatched between the other
        sub-batch's conv workgroups (high stream priority did not change that), and two 32-slice steps cost more than one
        64-slice step.  It is therefore OFF by default everywhere; it pays only where the policy side is a larger share."""
        n = int(torch.as_tensor(mat["gt"]).shape[0])
        parts = max(1, min(int(parts), n))
        if parts == 1 or self.device.type != "cuda":
            return self.run(mat, rtg, task)
        dev = self.device
        if not hasattr(self, "_forks"):
            self._forks, self._streams = {}, {}
        bounds = [(k * n) // parts for k in range(parts + 1)]
        cur = torch.cuda.current_stream(dev)
        gens, outs, streams = [], [], []
        hw = int(torch.as_tensor(mat["gt"]).shape[-2]) * int(torch.as_tensor(mat["gt"]).shape[-1])
        for k in range(parts):
            a, b = bounds[k], bounds[k + 1]
            if k not in self._forks:
                self._forks[k] = self.env if k == 0 else self.env.fork(k)
                # ONE stream per sub-batch, for its env steps and its policy calls alike: every tensor of the sub-batch is allocated,
                # written and read on that stream, so the caching allocator can never hand a block back while another stream still
                # reads it (a separate high-priority stream for the policy calls measured no faster and needed record_stream on
                # every tensor crossing over)
                self._streams[k] = torch.cuda.Stream(device=dev)
            def cut(key, v):
                v = torch.as_tensor(v)
                if key == "mask":
                    return v if v.numel() == hw else v.reshape(-1, *v.shape[-2:])[a:b]
                return v[a:b] if v.dim() > 0 and v.shape[0] == n else v
            sub = {key: cut(key, v) for key, v in mat.items()}
            st = self._streams[k]
            st.wait_stream(cur)
            out: Dict[str, object] = {}
            gens.append(self._run_phases(sub, rtg[a:b], task[a:b], None, self._forks[k], k, out))
            outs.append(out)
            streams.append(st)
        # round-robin over the sub-batches, one phase (a policy call or an env step) each; sub-batch k starts k phases late so that
        # steps and policy calls of different sub-batches face each other.  Stream order carries the chain step -> policy -> step.
        live = [True] * parts
        rnd = 0
        while any(live):
            for k in range(parts):
                if not live[k] or rnd < k:
                    continue
                with torch.cuda.stream(streams[k]):
                    try:
                        next(gens[k])
                    except StopIteration:
                        live[k] = False
            rnd += 1
        for k in range(parts):
            cur.wait_stream(streams[k])
        res = [o["result"] for o in outs]
        for r in res:                                       # allocated on a sub-batch's stream, read from here on on the caller's
            for t in (r.reward, r.initial_reward, r.stop_time, r.actions, r.x):
                if t.is_cuda:
                    t.record_stream(cur)
        return GreedyResult(reward=torch.cat([r.reward for r in res]), initial_reward=torch.cat([r.initial_reward for r in res]),
                            stop_time=torch.cat([r.stop_time for r in res]), actions=torch.cat([r.actions for r in res]),
                            x=torch.cat([r.x for r in res]))
