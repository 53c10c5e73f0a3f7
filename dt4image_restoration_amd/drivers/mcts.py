"""Tree search over (sigma_d, mu) on top of the greedy policy, batched over images.

Counterpart of /root/reference/evaluation/mcts.py: per round select by p-UCB (`select_p_ucb` :74-88), expand the
selected node into `n_children` children whose (sigma_d, mu) are sampled around the policy's prediction
(`sample_action_dict` :64-70, std 0.2 / 0.001, `expand_tree` :103-143), score the node with a no-reference rollout to
termination (`run_beam_search` :198-207), back up the maximum (`Node.backprop` :34-38), finally report the PSNR of
the best program (`get_best_program` :165-192).

BASELINE configs[3] scale (`MCTS.run_batch`): B images are searched AT ONCE, one tree per image.  Per round
  * selection walks each image's tree on the host (a few dozen float comparisons per image);
  * expansion is ONE policy call on the B selected nodes and ONE `env.step` on B x k rows (the k children of every image,
    engine of B*k replicas);
  * simulation is ONE no-reference greedy rollout of B rows with a per-row clock (`GreedyEvaluator.rollout_rows`: the
    selected nodes of different images sit at different depths; engine of B replicas);
so a round costs 1 + (rollout length) engine steps however many images are searched.  Node states live in one
preallocated device pool (x f32, z c64, u c64, T per node; a node is an integer id), filled by index copies.

Deliberate differences (the reference cannot be matched bit for bit here):
  * `env.step` in the reference mutates and returns one shared dict, so a node, its five children and the policy
    state all alias the same tensors (mcts.py:118-128 vs env.py:95-100).  Here every node owns its (x, z, u, T) and
    children start from their parent's state - the search the code evidently intends.
  * ARNIQA (a network fetch) is replaced by an injectable `scorer(states) -> [N]`; sampling uses seeded generators, one
    per image (seed, image index), so an image's search does not depend on which other images share the batch.
  * rewards are backed up to the ancestors (mcts.py:249,255 assign `node.reward = reward` before `backprop(reward)`,
    whose `reward > self.reward` test then never fires).
"""
from __future__ import annotations

import math
import time as _time
from collections import OrderedDict
from typing import Callable, Dict, List, Optional

import torch

from ..policy import policy_observation
from .greedy import GreedyEvaluator, PolicyContext


class NodePool:
    """Device storage of node states: row `id` of x [cap, H*W] f32, z / u [cap, H*W] c64, T [cap] f32 (20 B per pixel and
    node).  The pool GROWS on demand (geometrically, rows are copied once per doubling) instead of reserving every node a
    search could ever create: 64 images x 64 rounds x 5 children at 256 x 256 would be 27 GB up front, most of it for rounds
    whose selected node is terminal and expands nothing."""

    def __init__(self, capacity: int, h: int, w: int, device, max_capacity: Optional[int] = None):
        self.h, self.w, self.device = h, w, device
        self.max_capacity = max_capacity
        self.x = self.z = self.u = self.T = None
        self.used = 0
        self._resize(max(1, capacity))

    def _resize(self, capacity: int) -> None:
        px = self.h * self.w
        try:
            x = torch.empty((capacity, px), dtype=torch.float32, device=self.device)
            z = torch.empty((capacity, px), dtype=torch.complex64, device=self.device)
            u = torch.empty((capacity, px), dtype=torch.complex64, device=self.device)
            T = torch.empty((capacity,), dtype=torch.float32, device=self.device)
        except RuntimeError as exc:                             # out of device memory: say what the search needs
            raise RuntimeError(f"tree-search node pool: {capacity} node states of {self.h}x{self.w} need "
                               f"{capacity * (px * 20 + 4)} bytes of device memory ({exc})") from exc
        if self.used:
            x[:self.used] = self.x[:self.used]; z[:self.used] = self.z[:self.used]
            u[:self.used] = self.u[:self.used]; T[:self.used] = self.T[:self.used]
        self.x, self.z, self.u, self.T = x, z, u, T

    def alloc(self, count: int) -> torch.Tensor:
        need = self.used + count
        if need > self.x.shape[0]:
            cap = max(need, 2 * self.x.shape[0])
            if self.max_capacity is not None:
                if need > self.max_capacity:
                    raise RuntimeError("node pool exhausted")
                cap = min(cap, self.max_capacity)
            self._resize(cap)
        ids = torch.arange(self.used, self.used + count, device=self.x.device)
        self.used += count
        return ids

    def store(self, ids: torch.Tensor, states, rows=slice(None)) -> None:
        n = ids.shape[0]
        self.x[ids] = states["x"][rows].reshape(n, -1)
        self.z[ids] = states["z"][rows].reshape(n, -1)
        self.u[ids] = states["u"][rows].reshape(n, -1)
        self.T[ids] = states["T"][rows]

    def load(self, ids: torch.Tensor, states, repeat: int = 1) -> None:
        """states rows <- pool rows `ids`, each repeated `repeat` times (the replicas of one image are adjacent)."""
        if repeat > 1:
            ids = ids.repeat_interleave(repeat)
        shape = states["x"].shape
        states["x"].copy_(self.x[ids].reshape(shape))
        states["z"].copy_(self.z[ids].reshape(shape))
        states["u"].copy_(self.u[ids].reshape(shape))
        states["T"].copy_(self.T[ids])


class Node:
    def __init__(self, pool: NodePool, nid: int, time, prob, parent, edge, action, emb, ob, rtg, index):
        self.pool = pool
        self.nid = nid                # row of this node's state in the pool
        self.time = time
        self.prob = float(prob)
        self.parent = parent
        self.edge = edge
        self.action = action          # model-order action vector [3] that led here (None for the root)
        self.emb = emb                # cached state-encoder output of this node's observation [E] (or None)
        self.ob = ob                  # policy observation [16384] (kept only when embeddings are not cached)
        self.rtg = rtg                # return-to-go token written at this node's time step [1]
        self.index = index
        self.children: List["Node"] = []
        self.reward = 0.0
        self.visits = 0
        self.rollout_reward: Optional[float] = None
        self.final_x: Optional[torch.Tensor] = None

    @property
    def snap(self) -> Dict[str, torch.Tensor]:
        """Views of this node's state in the pool (x [1,1,H,W] f32, z, u c64, T [1])."""
        p, i = self.pool, self.nid
        return {"x": p.x[i].view(1, 1, p.h, p.w), "z": p.z[i].view(1, 1, p.h, p.w), "u": p.u[i].view(1, 1, p.h, p.w),
                "T": p.T[i:i + 1]}

    def __repr__(self):
        return f"Node(time = {self.time}, edge = {self.edge})_{self.index}"

    def backprop(self, reward: float):                        # mcts.py:34-38: keep the maximum along the path
        if reward > self.reward:
            self.reward = reward
            if self.parent is not None:
                self.parent.backprop(reward)

    def chain(self) -> List["Node"]:
        out, n = [], self
        while n is not None:
            out.append(n)
            n = n.parent
        return out[::-1]


def select_p_ucb(parent: Node, children: List[Node]) -> Node:
    """mcts.py:74-88 (the unused beta term of the reference is dropped)."""
    best, best_v = parent, -1000.0
    for c in children:
        v = (c.reward - parent.reward) + c.prob * math.sqrt(max(math.log(max(parent.visits, 1)), 0.0)) / (1 + c.visits)
        if v > best_v:
            best, best_v = c, v
    return best


def sample_around(value: float, std: float, count: int, gen: torch.Generator):
    """mcts.py:64-70: |N(value, std)| samples sorted by density, with their densities."""
    x = (value + std * torch.randn(count, generator=gen)).abs()
    p = torch.exp(-0.5 * ((x - value) / std) ** 2) / (std * math.sqrt(2 * math.pi))
    p, idx = torch.sort(p, descending=True)
    return x[idx], p


class MCTS:
    def __init__(self, evaluator: GreedyEvaluator, scorer: Callable[[Dict[str, torch.Tensor]], torch.Tensor],
                 n_children: int = 5, rounds: int = 30, seed: int = 0):
        self.ev = evaluator
        self.env = evaluator.env
        self.scorer = scorer
        self.k = n_children
        self.rounds = rounds
        self.seed = seed
        self.last_stats: Dict[str, float] = {}

    # -- helpers -------------------------------------------------------------------------------------------------
    def _fill_context(self, ctx: PolicyContext, row: int, node: Node) -> None:
        """Rebuild row `row` of the policy context from `node`'s ancestor chain (mcts.py:40-59 build_eval/build_action)."""
        for nd in node.chain():
            if ctx.ee is not None:
                ctx.ee[row, nd.time] = nd.emb
            if nd.ob is not None:
                ctx.es[row, nd.time] = nd.ob
            ctx.er[row, nd.time] = nd.rtg
            if nd.action is not None and nd.time >= 1:
                ctx.ea[row, nd.time - 1] = nd.action

    def _context(self, nodes: List[Node], task: torch.Tensor) -> PolicyContext:
        ctx = self.ev.buffers(len(nodes), task)
        for b, nd in enumerate(nodes):
            self._fill_context(ctx, b, nd)
        return ctx

    def _plan(self, nodes: List[Node], ctx: PolicyContext):
        """Policy proposal at each row's node: `_initial` for roots, `_predict` at the node's own time otherwise."""
        times = [nd.time for nd in nodes]
        if all(t == 0 for t in times):
            return self.ev._initial(ctx)
        if len(set(times)) == 1:
            return self.ev._predict(ctx, times[0])
        dev = self.ev.device
        tv = torch.tensor(times, dtype=torch.int64, device=dev)
        roots = tv == 0
        if not bool(roots.any()):
            return self.ev._predict_rows(ctx, tv)
        # mixed: roots use the first-step call (eval.py:80-98), the others their own time.  Both calls write the action they
        # picked into the context for EVERY row (ea[:, 0] / ea[row, t]); keep each row's write from its own call only.
        ea0, ea1 = ctx.ea[:, 0].clone(), ctx.ea[:, 1].clone()
        a0, r0 = self.ev._initial(ctx)
        ctx.ea[~roots, 0] = ea0[~roots]
        action, prtg = self.ev._predict_rows(ctx, tv.clamp(min=1))
        ctx.ea[roots, 1] = ea1[roots]
        for k in action:
            action[k] = torch.where(roots, a0[k], action[k])
        return action, torch.where(roots.reshape(-1, 1), r0, prtg)

    # -- search --------------------------------------------------------------------------------------------------
    def run_batch(self, mat: Dict[str, torch.Tensor], rtg: torch.Tensor, task: torch.Tensor,
                  first_state: Optional[torch.Tensor] = None, first_image: int = 0):
        """B images at once (one tree each).  Returns (best PSNR [B,1] CPU, list of B roots).
        first_image: index of this batch's first image in the whole job - image i samples its children from the stream seeded
        with (seed, first_image + i), so a shard of a job searches exactly the trees the unsharded job does."""
        k, dev, ev = self.k, self.ev.device, self.ev
        mat = {key: torch.as_tensor(v) for key, v in mat.items()}
        B = mat["gt"].shape[0]
        t_start = _time.perf_counter()
        # engine of B rows for the rollouts, engine of B*k rows for the expansions (the k children of an image are adjacent)
        # (a per-image mask [B,H,W] - the reference reads one per .mat file - is repeated like the images; one shared [H,W] mask is not)
        hw = mat["gt"].shape[-2] * mat["gt"].shape[-1]
        def replicate(key, v):
            if key == "mask":
                return v.reshape(-1, *v.shape[-2:]).repeat_interleave(k, dim=0) if v.numel() != hw else v
            return v.repeat_interleave(k, dim=0) if v.dim() > 2 else v
        rep = {key: replicate(key, v) for key, v in mat.items()}
        st_roll = self.env.reset(mat, dev)
        st_exp = self.env.reset(rep, dev)
        h, w = st_roll["z"].shape[-2:]
        # roots + the first rounds' children; grows when the trees do (NodePool); the bound is every round expanding every image
        pool = NodePool(B * (1 + k * min(self.rounds, 8)), h, w, dev, max_capacity=B * (1 + k * self.rounds))
        cache_emb = ev.cache_state_embeddings

        def observation(x):                                      # -> (emb [n,E] or None, ob [n,16384] or None)
            ob = policy_observation(x)
            return (ev.model.encode_states(ob), None) if cache_emb else (None, ob)

        if first_state is None:
            first_state = mat["x0_raw"] if "x0_raw" in mat else st_roll["x"]
        if first_state.is_complex():
            first_state = first_state.real
        with torch.no_grad():
            emb0, ob0 = observation(first_state.to(dev).float().reshape(B, 1, h, w))
        root_ids = pool.alloc(B)
        pool.store(root_ids, st_roll)
        rtg0 = rtg.reshape(B, 1).to(dev).float()
        roots = [Node(pool, int(root_ids[b]), 0, 1.0, None, 0, None, None if emb0 is None else emb0[b],
                      None if ob0 is None else ob0[b], rtg0[b], 0) for b in range(B)]
        gens = [torch.Generator().manual_seed(self.seed * 1000003 + first_image + b) for b in range(B)]
        order = list(ev.model.action_range.keys())
        n_rollouts = 0
        for rnd in range(self.rounds):
            # ---- selection (host) ----
            sel: List[Node] = []
            for b in range(B):
                node = roots[b]
                node.visits += 1
                while node.children:
                    node = select_p_ucb(node, node.children)
                    node.visits += 1
                sel.append(node)
            expand = [nd.time < ev.max_timesteps - 1 for nd in sel]
            for nd, ex in zip(sel, expand):
                if not ex:
                    nd.backprop(nd.reward)
            if not any(expand):
                continue
            # ---- expansion: one policy call on the B selected nodes, one env.step on their B*k children ----
            ctx = self._context(sel, task)
            action, pred_rtg = self._plan(sel, ctx)
            a_cpu = {key: action[key].detach().float().cpu() for key in ("sigma_d", "mu")}      # one sync per round
            sig = torch.empty(B, k); mu = torch.empty(B, k); probs = torch.empty(B, k)
            for b in range(B):
                sig[b], probs[b] = sample_around(float(a_cpu["sigma_d"][b]), 0.2, k, gens[b])
                mu[b], _ = sample_around(float(a_cpu["mu"][b]), 0.001, k, gens[b])
            child_action = OrderedDict((key, v.repeat_interleave(k)) for key, v in action.items())
            child_action["sigma_d"] = sig.reshape(-1).to(dev)
            child_action["mu"] = mu.reshape(-1).to(dev)
            if not all(expand):                                   # images whose selected node is terminal sit this round out
                ex_rows = torch.tensor(expand, device=dev).repeat_interleave(k)
                child_action["T"] = torch.where(ex_rows, child_action["T"], torch.ones_like(child_action["T"]))
            sel_ids = torch.tensor([nd.nid for nd in sel], device=dev)
            pool.load(sel_ids, st_exp, repeat=k)
            st_exp, _ = self.env.step(st_exp, child_action)
            with torch.no_grad():
                emb_c, ob_c = observation(st_exp["x"])
            child_ids = pool.alloc(B * k)
            pool.store(child_ids, st_exp)
            cid = child_ids.tolist()
            vecs = torch.stack([child_action[key] for key in order], dim=1)                      # [B*k, 3] model order
            for b in range(B):
                if not expand[b]:
                    continue
                nd = sel[b]
                for i in range(k):
                    r = b * k + i
                    nd.children.append(Node(pool, cid[r], nd.time + 1, probs[b, i], nd, i, vecs[r],
                                            None if emb_c is None else emb_c[r], None if ob_c is None else ob_c[r],
                                            pred_rtg[b], rnd))
            # ---- simulation: no-reference greedy rollout from the selected node itself (mcts.py:242-252), B rows, per-row clock
            need = [expand[b] and sel[b].rollout_reward is None for b in range(B)]
            if any(need):
                pool.load(sel_ids, st_roll)
                ctx = self._context(sel, task)
                act, prtg = self._plan(sel, ctx)
                starts = torch.tensor([nd.time + 1 for nd in sel], dtype=torch.int64)
                reward, _ = ev.rollout_rows(st_roll, act, prtg, starts, ctx, scorer=self.scorer,
                                            active=torch.tensor(need))
                for b in range(B):
                    if need[b]:
                        sel[b].rollout_reward = float(reward[b])
                        sel[b].final_x = st_roll["x"][b:b + 1].clone()      # this row only (a view would pin the whole batch)
                        n_rollouts += 1
            for b in range(B):
                if expand[b]:
                    # (the reference assigns node.reward first, which turns its own backprop into a no-op)
                    sel[b].backprop(sel[b].rollout_reward)
        # ---- best program per image: the rollout with the highest score; its PSNR (mcts.py:165-192) ----
        best_x = []
        for b in range(B):
            best, stack = None, [roots[b]]
            while stack:
                nd = stack.pop()
                stack += nd.children
                if nd.rollout_reward is not None and (best is None or nd.rollout_reward > best.rollout_reward):
                    best = nd
            best_x.append(best.final_x if best is not None else roots[b].snap["x"])
        psnr = self.env.compute_reward(torch.cat(best_x, dim=0).contiguous(), st_roll["gt"])
        torch.cuda.synchronize() if dev.type == "cuda" else None
        dt = _time.perf_counter() - t_start
        self.last_stats = {"images": B, "rounds": self.rounds, "rollouts": n_rollouts, "seconds": dt,
                           "rollouts_per_s": n_rollouts / dt if dt > 0 else 0.0, "nodes": pool.used}
        return psnr, roots

    def run(self, mat: Dict[str, torch.Tensor], rtg: torch.Tensor, task: torch.Tensor,
            first_state: Optional[torch.Tensor] = None):
        """One image (batch 1 in `mat`), the reference's call shape (mcts.py:212).  Returns (best PSNR [1,1], root)."""
        psnr, roots = self.run_batch(mat, rtg, task, first_state)
        return psnr[:1], roots[0]
