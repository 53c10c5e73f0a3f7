#!/usr/bin/env python3
"""BASELINE configs[3] at scale on one GPU: B images of 256x256 searched at once, 64 rounds (rollouts) per image, the reference's
5 children per expansion, stub no-reference scorer (ARNIQA is a network fetch), fixed seed.  Prints one JSON line.

    python tools/mcts_scale.py --images 64 --rounds 64
"""
import argparse
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dt4image_restoration_amd import data as D, synthetic, weights  # noqa: E402
from dt4image_restoration_amd.denoiser import UNetDenoiser2D  # noqa: E402
from dt4image_restoration_amd.drivers.greedy import GreedyEvaluator  # noqa: E402
from dt4image_restoration_amd.drivers.mcts import MCTS  # noqa: E402
from dt4image_restoration_amd.env import PnPEnv  # noqa: E402
from dt4image_restoration_amd.policy import DecisionTransformer, DecisionTransformerConfig  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=64)
    ap.add_argument("--size", type=int, default=256)
    args = ap.parse_args()
    den = UNetDenoiser2D.seeded(0, "unit_gain")
    m = DecisionTransformer(DecisionTransformerConfig(block_size=18, n_embeds=9, mode="norm"))
    m.load_state_dict(weights.generate_policy_weights(m, 7, t_bias=-1.0, head_gain=12.0))

    def scorer(states):
        x = states["x"]
        return 1.0 / (1e-3 + ((x - F.avg_pool2d(x, 3, 1, 1)) ** 2).mean(dim=(1, 2, 3)))

    B = args.images
    p = synthetic.make_problem(B, args.size, args.size, accel=4.0, seed=9)
    mat = {k: torch.from_numpy(np.asarray(v)) for k, v in p.items()}
    ev = GreedyEvaluator(m, PnPEnv(30, den, "cuda"), max_timesteps=30, device_type="cuda", sync_every=4)
    tree = MCTS(ev, scorer, n_children=5, rounds=args.rounds, seed=11)
    psnr, roots = tree.run_batch(mat, torch.full((B,), D.normalised_rtg(10.0)), torch.full((B,), 4))
    st = tree.last_stats
    x0 = ev.env.compute_reward(torch.from_numpy(p["x0"][..., 0]).cuda().float().contiguous(), torch.from_numpy(p["gt"]).cuda())
    print(json.dumps({"workload": f"configs[3] geometry: {B} images of {args.size}x{args.size} searched at once, {args.rounds} rounds per image, "
                                  "5 children per expansion, stub scorer", "images": B, "rounds": args.rounds,
                      "rollouts": st["rollouts"], "seconds": round(st["seconds"], 3), "rollouts_per_s": round(st["rollouts_per_s"], 2),
                      "tree_nodes_per_image": sum(1 for _ in _walk(roots[0])), "psnr_mean_db": round(float(psnr.mean()), 4),
                      "psnr_x0_mean_db": round(float(x0.mean()), 4), "device_memory_gb": round(torch.cuda.max_memory_allocated() / 2**30, 2)}))


def _walk(node):
    stack = [node]
    while stack:
        nd = stack.pop()
        yield nd
        stack += nd.children


if __name__ == "__main__":
    main()
