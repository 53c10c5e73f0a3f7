"""Command line with the reference's sub-commands (/root/reference/main.py:133-240, scripts.sh):

    python -m dt4image_restoration_amd.cli --block_size 18 --n_embeds 9 eval --rtg 10 --max_timesteps 30
    python -m dt4image_restoration_amd.cli --block_size 18 --n_embeds 9 mcts --rtg 5  --max_timesteps 30
    python -m dt4image_restoration_amd.cli --block_size 18 --n_embeds 6 flex --max_timesteps 30

Multi-GPU (BASELINE configs[2]): launch the same command under `python -m torch.distributed.run --nproc-per-node N
--master-addr 127.0.0.1 -m dt4image_restoration_amd.cli ... eval|mcts|flex ...`: every rank takes a contiguous shard of each
set's images (drivers/sharded.py), the per-image PSNR / stop iteration are gathered over RCCL, rank 0 prints.

Differences: `train` is out of scope; checkpoint and data locations are options instead of
hard-coded paths (main.py:175,178,181-183); without `--data` the run uses the seeded synthetic problems and without
`--denoiser-ckpt` / `--policy-ckpt` the seeded stand-in weights (the real ones are external downloads); images of one
directory are evaluated as ONE batch instead of `DataLoader(batch_size=1)` (main.py:232, eval.py:232).
"""
from __future__ import annotations

import argparse
import json
import os
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


def _sets(args, flex_target=None):
    """(name, number of images, load(start, stop) -> (batch dict, task tokens)) per evaluation set."""
    from . import data as D, synthetic
    if args.data:
        for d in args.data:
            def load(a, b, d=d):
                batch, tasks = D.load_dir(d, limit=args.limit, start=a, stop=b)
                return batch, D.task_tokens(tasks, flex_target)
            yield d, len(D.list_dir(d, args.limit)), load
    else:
        for accel, sig in ((4, 10), (8, 10)):
            def load(a, b, accel=accel, sig=sig):
                p = synthetic.make_problem(b - a, args.size, args.size, accel=accel, sigma_n=sig / 255.0, seed=args.seed + accel,
                                           first_slice=a)
                return p, D.task_tokens([f"{accel}x_{sig}"] * (b - a), flex_target)
            yield f"synthetic {accel}x_{sig}", args.limit or 7, load


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
    from .drivers.sharded import run_sharded_greedy, run_sharded_mcts
    out = []
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    dist = None
    if world > 1:                                            # one process per GPU under torch.distributed.run
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
    if args.mode == "eval":
        model, env, _ = _build(args, "norm")
        ev = GreedyEvaluator(model, env, max_timesteps=args.max_timesteps, block_size=args.block_size,
                             device_type=torch.device("cuda", torch.cuda.current_device()), sync_every=5)
        for name, total, load in _sets(args):
            def load_shard(a, b, load=load):
                batch, tokens = load(a, b)
                mat = {k: torch.from_numpy(np.asarray(v)) for k, v in batch.items()}
                return mat, torch.full((b - a,), D.normalised_rtg(args.rtg)), torch.from_numpy(tokens)
            r = run_sharded_greedy(ev, total, load_shard, sync=torch.cuda.synchronize)
            out.append({"set": name, "n": total, "psnr": float(r.reward.mean()),
                        "psnr_increment": float((r.reward - r.initial_reward).mean()),
                        "mean_stop_iteration": float(r.stop_time.float().mean()), "ranks": world})
            if rank == 0:
                print(json.dumps(out[-1]), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return out
    if args.mode == "mcts":
        model, env, scorer = _build(args, "norm")
        ev = GreedyEvaluator(model, env, max_timesteps=args.max_timesteps, block_size=args.block_size,
                             device_type=torch.device("cuda", torch.cuda.current_device()), sync_every=4)
        for name, total, load in _sets(args):
            def load_shard(a, b, load=load):
                batch, tokens = load(a, b)
                mat = {k: torch.from_numpy(np.asarray(v)) for k, v in batch.items()}
                return mat, torch.full((b - a,), D.normalised_rtg(args.rtg)), torch.from_numpy(tokens)
            # all images of a rank's shard are searched at once: one tree per image, children and rollouts batched over images;
            # the policy's first token is the UNclipped Re x0 (datasets.py:162; `x0_raw` of the batch)
            tree = MCTS(ev, scorer, rounds=args.rollouts, seed=args.seed)
            psnr, rollouts, secs = run_sharded_mcts(tree, total, load_shard)
            out.append({"set": name, "n": total, "mcts_psnr": float(psnr.mean()),
                        "rollouts_per_s": round(rollouts / secs, 2) if secs > 0 else 0.0, "ranks": world})
            if rank == 0:
                print(json.dumps(out[-1]), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return out
    if True:
        # main.py:187-209: the greedy evaluation once per return-to-go target, PSNR increment averaged over the sets; sharded over
        # the ranks like `eval` (every rank a contiguous shard of each set, one gather per set)
        model, env, _ = _build(args, "flex")
        ev = GreedyEvaluator(model, env, max_timesteps=args.max_timesteps, block_size=args.block_size,
                             device_type=torch.device("cuda", torch.cuda.current_device()))
        for target in (1.5, 3, 3.5, 4, 4.5):                       # main.py:198
            incs = []
            for name, total, load in _sets(args, flex_target=target):
                def load_shard(a, b, load=load):
                    batch, tokens = load(a, b)
                    mat = {k: torch.from_numpy(np.asarray(v)) for k, v in batch.items()}
                    return mat, torch.full((b - a,), D.normalised_rtg(target, flex=True)), torch.from_numpy(tokens)
                r = run_sharded_greedy(ev, total, load_shard, sync=torch.cuda.synchronize)
                incs.append(float((r.reward - r.initial_reward).mean()))
            out.append({"rtg_target": target, "average_increment": float(np.mean(incs)), "ranks": world})
            if rank == 0:
                print(json.dumps(out[-1]), flush=True)
        if dist is not None:
            dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main(sys.argv[1:])
