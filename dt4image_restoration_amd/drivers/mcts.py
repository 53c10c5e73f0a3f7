"""Tree search over (sigma_d, mu) on top of the greedy policy.

Counterpart of /root/reference/evaluation/mcts.py: per round select by p-UCB (`select_p_ucb` :74-88), expand the
selected node into `n_children` children whose (sigma_d, mu) are sampled around the policy's prediction
(`sample_action_dict` :64-70, std 0.2 / 0.001, `expand_tree` :103-143), score the node with a no-reference rollout to
termination (`run_beam_search` :198-207), back up the maximum (`Node.backprop` :34-38), finally report the PSNR of
the best program (`get_best_program` :165-192).

Deliberate differences (the reference cannot be matched bit for bit here):
  * `env.step` in the reference mutates and returns one shared dict, so a node, its five children and the policy
    state all alias the same tensors (mcts.py:118-128 vs env.py:95-100).  Here every node owns a snapshot of its
    (x, z, u, T) and children start from their parent's snapshot - the search the code evidently intends.
  * ARNIQA (a network fetch) is replaced by an injectable `scorer(states) -> [N]`; sampling uses a seeded generator.
  * rewards are backed up to the ancestors (mcts.py:249,255 assign `node.reward = reward` before `backprop(reward)`,
    whose `reward > self.reward` test then never fires).
  * the children of one expansion are stepped as ONE batch (the engine holds n_children replicas of the image).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Callable, Dict, List, Optional

import torch

from ..policy import policy_observation
from .greedy import GreedyEvaluator


class Node:
    def __init__(self, snap, time, prob, parent, edge, action, ob, rtg, index):
        self.snap = snap              # {'x','z','u','T'} tensors [1,...]: this node's environment state
        self.time = time
        self.prob = float(prob)
        self.parent = parent
        self.edge = edge
        self.action = action          # model-order action vector [3] that led here (None for the root)
        self.ob = ob                  # policy observation of this node's state [16384]
        self.rtg = rtg                # return-to-go token written at this node's time step [1]
        self.index = index
        self.children: List["Node"] = []
        self.reward = 0.0
        self.visits = 0

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
        self.gen = torch.Generator().manual_seed(seed)

    # -- helpers -------------------------------------------------------------------------------------------------
    def _context(self, node: Node, task: torch.Tensor, n: int):
        """Rebuild the policy context of `node` from its ancestor chain (mcts.py:40-59 build_eval/build_action),
        replicated n times."""
        es, ea, er, et, ek = self.ev.buffers(n, task.expand(n))
        for nd in node.chain():
            es[:, nd.time] = nd.ob
            er[:, nd.time] = nd.rtg
            if nd.action is not None and nd.time >= 1:
                ea[:, nd.time - 1] = nd.action
        return es, ea, er, et, ek

    def _load(self, states, snap, rows):
        for key in ("x", "z", "u", "T"):
            states[key][rows] = snap[key]

    def _plan(self, node: Node, ctx):
        es, ea, er, et, ek = ctx
        if node.time == 0:
            return self.ev._initial(es, ea, er, et, ek)
        return self.ev._predict(es, ea, er, et, ek, node.time)

    # -- search --------------------------------------------------------------------------------------------------
    def run(self, mat: Dict[str, torch.Tensor], rtg: torch.Tensor, task: torch.Tensor):
        """One image (batch 1 in `mat`).  Returns (best PSNR [1,1], root)."""
        k, dev = self.k, self.ev.device
        rep = {key: (torch.as_tensor(v).expand(k, *torch.as_tensor(v).shape[1:]).contiguous() if key != "mask" else v)
               for key, v in mat.items()}
        states = self.env.reset(rep, dev)                       # k replicas: children are stepped as one batch
        root_snap = {key: states[key][:1].clone() for key in ("x", "z", "u", "T")}
        root = Node(root_snap, 0, 1.0, None, 0, None, policy_observation(states["x"][:1])[0],
                    rtg.reshape(1).to(dev).float(), 0)
        cache: Dict[str, float] = {}
        finals: Dict[str, torch.Tensor] = {}
        for rnd in range(self.rounds):
            node = root
            node.visits += 1
            while node.children:                                # selection
                node = select_p_ucb(node, node.children)
                node.visits += 1
            if node.time >= self.ev.max_timesteps - 1:
                node.backprop(node.reward)
                continue
            # expansion: policy proposal at the node, k perturbed children stepped in one batch
            ctx = self._context(node, task, k)
            action, pred_rtg = self._plan(node, ctx)
            sig, probs = sample_around(float(action["sigma_d"][0]), 0.2, k, self.gen)
            mu, _ = sample_around(float(action["mu"][0]), 0.001, k, self.gen)
            child_action = OrderedDict(action)
            child_action["sigma_d"] = sig.to(dev)
            child_action["mu"] = mu.to(dev)
            self._load(states, node.snap, slice(None))
            states, _ = self.env.step(states, child_action)
            obs = policy_observation(states["x"])
            order = list(self.ev.model.action_range.keys())
            for i in range(k):
                vec = torch.stack([child_action[key][i] for key in order])
                snap = {key: states[key][i:i + 1].clone() for key in ("x", "z", "u", "T")}
                node.children.append(Node(snap, node.time + 1, probs[i], node, i, vec, obs[i], pred_rtg[0], rnd))
            # simulation: no-reference greedy rollout from the node itself (mcts.py:242-252)
            key = repr(node)
            if key not in cache:
                self._load(states, node.snap, slice(None))
                ctx = self._context(node, task, k)
                act, prtg = self._plan(node, ctx)
                reward, _ = self.ev.rollout(states, act, prtg, node.time + 1, *ctx, scorer=self.scorer)
                cache[key] = float(reward[0])
                finals[key] = states["x"][:1].clone()
            node.backprop(cache[key])       # (the reference assigns node.reward first, which turns its own backprop into a no-op)
        best = max(cache, key=cache.get)
        psnr = self.env.compute_reward(finals[best].expand(k, -1, -1, -1).contiguous(), states["gt"])[:1]
        return psnr, root
