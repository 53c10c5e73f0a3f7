"""Command line with the reference's sub-commands (/root/reference/main.py:133-240, scripts.sh):

    python -m dt4image_restoration_amd.cli --block_size 18 --n_embeds 9 eval --rtg 10 --max_timesteps 30
    python -m dt4image_restoration_amd.cli --block_size 18 --n_embeds 9 mcts --rtg 5  --max_timesteps 30
    python -m dt4image_restoration_amd.cli --block_size 18 --n_embeds 6 flex --max_timesteps 30

Differences: `train` is out of scope; checkpoint and data locations are options instead of
hard-coded paths (main.py:175,178,181-183); without `--data` the run uses the seeded synthetic problems and without
`--denoiser-ckpt` / `--policy-ckpt` the seeded stand-in weights (the real ones are external downloads); images of one
directory are evaluated as ONE batch instead of `DataLoader(batch_size=1)` (main.py:232, eval.py:232).
"""
from __future__ import annotations

import argparse
import json
import sys

import numpy as np
import torch


def _build(args, mode):
    from . import weights
    from .denoiser import UNetDenoiser2D
    from .env import PnPEnv
    from .policy import DecisionTransformer, DecisionTransformerConfig
    model = DecisionTransformer(DecisionTransformerConfig(block_size=args.block_size, n_embeds=args.n_embeds, mode=mode))
    if args.policy_ckpt:
        model.load_state_dict(torch.load(args.policy_ckpt, map_location="cpu"))
    else:
        model.load_state_dict(weights.generate_policy_weights(model, args.seed, t_bias=-1.0, head_gain=8.0))
    den = UNetDenoiser2D(ckpt_path=args.denoiser_ckpt) if args.denoiser_ckpt else UNetDenoiser2D.seeded(args.seed)
    scorer = (lambda st: 1.0 / (1e-3 + (st["x"] - torch.nn.functional.avg_pool2d(st["x"], 3, 1, 1)).pow(2).mean(dim=(1, 2, 3))))
    return model, PnPEnv(max_episode_step=30, denoiser=den, device_type="cuda", no_ref_scorer=None), scorer


def _batches(args, flex_target=None):
    from . import data as D, synthetic
    if args.data:
        for d in args.data:
            batch, tasks = D.load_dir(d, limit=args.limit)
            yield d, batch, D.task_tokens(tasks, flex_target)
    else:
        for accel, sig in ((4, 10), (8, 10)):
            p = synthetic.make_problem(args.limit or 7, args.size, args.size, accel=accel, sigma_n=sig / 255.0, seed=args.seed + accel)
            tasks = [f"{accel}x_{sig}"] * p["gt"].shape[0]
            yield f"synthetic {accel}x_{sig}", p, D.task_tokens(tasks, flex_target)


def main(argv=None):
    ap = argparse.ArgumentParser(description="PnP-ADMM CS-MRI restoration with a decision-transformer policy (MI355X)")
    ap.add_argument("--block_size", type=int, required=True)
    ap.add_argument("--n_embeds", type=int, required=True)
    ap.add_argument("--denoiser-ckpt", default=None)
    ap.add_argument("--policy-ckpt", default=None)
    ap.add_argument("--data", nargs="*", default=None, help="directories of .mat files (one batch each)")
    ap.add_argument("--limit", type=int, default=7, help="images per directory (the reference averages the first 7)")
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--seed", type=int, default=0)
    sub = ap.add_subparsers(dest="mode", required=True)
    for name in ("eval", "mcts"):
        sp = sub.add_parser(name)
        sp.add_argument("--rtg", type=float, default=10.0)
        sp.add_argument("--max_timesteps", type=int, default=30)
        if name == "mcts":
            sp.add_argument("--rollouts", type=int, default=30)
    sub.add_parser("flex").add_argument("--max_timesteps", type=int, default=30)
    args = ap.parse_args(argv)

    from . import data as D
    from .drivers.greedy import GreedyEvaluator
    from .drivers.mcts import MCTS
    out = []
    if args.mode in ("eval", "mcts"):
        model, env, scorer = _build(args, "norm")
        ev = GreedyEvaluator(model, env, max_timesteps=args.max_timesteps, block_size=args.block_size)
        for name, batch, tokens in _batches(args):
            mat = {k: torch.from_numpy(np.asarray(v)) for k, v in batch.items()}
            n = mat["gt"].shape[0]
            rtg = torch.full((n,), D.normalised_rtg(args.rtg))
            first = mat.get("x0_raw")               # datasets.py:162: the policy's first token is the UNclipped Re x0
            if args.mode == "eval":
                r = ev.run(mat, rtg, torch.from_numpy(tokens), first_state=first)
                out.append({"set": name, "n": n, "psnr": float(r.reward.mean()), "psnr_increment": float((r.reward - r.initial_reward).mean()),
                            "mean_stop_iteration": float(r.stop_time.float().mean())})
            else:
                tree = MCTS(ev, scorer, rounds=args.rollouts, seed=args.seed)
                rewards = [float(tree.run({k: (v[i:i + 1] if k != "mask" else v) for k, v in mat.items()}, rtg[i:i + 1],
                                          torch.from_numpy(tokens[i:i + 1]))[0]) for i in range(n)]
                out.append({"set": name, "n": n, "mcts_psnr": float(np.mean(rewards))})
            print(json.dumps(out[-1]), flush=True)
    else:
        model, env, _ = _build(args, "flex")
        ev = GreedyEvaluator(model, env, max_timesteps=args.max_timesteps, block_size=args.block_size)
        for target in (1.5, 3, 3.5, 4, 4.5):                       # main.py:198
            incs = []
            for name, batch, tokens in _batches(args, flex_target=target):
                mat = {k: torch.from_numpy(np.asarray(v)) for k, v in batch.items()}
                n = mat["gt"].shape[0]
                r = ev.run(mat, torch.full((n,), D.normalised_rtg(target, flex=True)), torch.from_numpy(tokens),
                           first_state=mat.get("x0_raw"))
                incs.append(float((r.reward - r.initial_reward).mean()))
            out.append({"rtg_target": target, "average_increment": float(np.mean(incs))})
            print(json.dumps(out[-1]), flush=True)
    return out


if __name__ == "__main__":
    main(sys.argv[1:])
