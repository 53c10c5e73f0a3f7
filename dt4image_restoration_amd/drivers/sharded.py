"""Sharded DT-driven episode (BASELINE configs[2]: 512 slices over 8 GPUs, 30 iterations).

The loop the reference runs one image at a time on one GPU (`Evaluator._generate` / `run_greedy`,
/root/reference/evaluation/eval.py:105-143,189-220) is cut by slices: rank r of W owns the contiguous shard
`sharding.shard_range(total, r, W)`, runs the batched greedy rollout on it with NO data-path collective (slices are
independent units, env.py:74-100), and the per-slice results - final PSNR, initial PSNR, stop iteration - are assembled on
every rank by ONE padded all_gather each (`sharding.gather_per_slice`: RCCL over xGMI on GPUs via backend "nccl", gloo on
CPU in the tests).  Weights are replicated.  One process per GPU; launched by `torch.distributed.run` (bench.py --mode
greedy, `python -m torch.distributed.run ... -m dt4image_restoration_amd.cli ... eval`) or stand-alone (world size 1).
"""
from __future__ import annotations

import time
from dataclasses import dataclass
from typing import Callable, Dict, Optional, Tuple

import torch

from .. import sharding
from .greedy import GreedyEvaluator


@dataclass
class ShardedResult:
    reward: torch.Tensor          # [total, 1] final PSNR of every slice of the job (all ranks hold the full tensor)
    initial_reward: torch.Tensor  # [total, 1]
    stop_time: torch.Tensor       # [total] iteration at which each slice stopped
    local_range: Tuple[int, int]  # this rank's [start, stop)
    seconds: float                # this rank's rollout wall time (reset + DT-driven steps), max over ranks
    steps: int                    # env steps of the longest episode in the job (max stop_time)


def world_info(group=None) -> Tuple[int, int]:
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def run_sharded_greedy(evaluator: GreedyEvaluator, total: int,
                       load_shard: Callable[[int, int], Tuple[Dict[str, torch.Tensor], torch.Tensor, torch.Tensor]],
                       group=None, sync: Optional[Callable[[], None]] = None, pipeline: int = 1) -> ShardedResult:
    """`load_shard(start, stop)` -> (mat dict of the slices [start, stop), rtg [n], task [n]) - only this rank's shard is
    ever materialised.  `sync()` (e.g. torch.cuda.synchronize) brackets the timed rollout.  pipeline > 1: the shard advances
    as that many sub-batches on their own streams, one's policy call under another's env step (`GreedyEvaluator.run_pipelined`)."""
    import torch.distributed as dist
    rank, world = world_info(group)
    a, b = sharding.shard_range(total, rank, world)
    mat, rtg, task = load_shard(a, b)
    dev = evaluator.device
    if sync is not None:
        sync()
    if world > 1:
        dist.barrier(group)
    t0 = time.perf_counter()
    if b > a:
        res = evaluator.run_pipelined(mat, rtg, task, pipeline) if pipeline > 1 else evaluator.run(mat, rtg, task)
        local = (res.reward.to(dev).float(), res.initial_reward.to(dev).float(), res.stop_time.to(dev))
    else:                                                 # more ranks than slices: an empty shard still joins the gather
        local = (torch.zeros((0, 1), device=dev), torch.zeros((0, 1), device=dev), torch.zeros((0,), dtype=torch.int64, device=dev))
    if sync is not None:
        sync()
    if world > 1:
        dist.barrier(group)
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX, group=group)
    # the path's only exchange: per-slice results, once per episode
    reward = sharding.gather_per_slice(local[0], total, group)
    initial = sharding.gather_per_slice(local[1], total, group)
    stop = sharding.gather_per_slice(local[2], total, group)
    return ShardedResult(reward=reward.cpu(), initial_reward=initial.cpu(), stop_time=stop.cpu(), local_range=(a, b),
                         seconds=float(dt.item()), steps=int(stop.max()) if total else 0)


def run_sharded_mcts(tree, total: int, load_shard: Callable[[int, int], Tuple[Dict[str, torch.Tensor], torch.Tensor, torch.Tensor]],
                     group=None):
    """BASELINE configs[3] over several GPUs: the images of a set are cut into contiguous shards like the slices of the greedy
    episode, every rank searches its own images at once (`MCTS.run_batch`: one tree per image) and the per-image PSNR of the
    best program is gathered - again the only exchange.  Returns (psnr [total, 1] CPU, rollouts of the whole job, seconds)."""
    import torch.distributed as dist
    rank, world = world_info(group)
    a, b = sharding.shard_range(total, rank, world)
    dev = tree.ev.device
    t0 = time.perf_counter()
    if b > a:
        mat, rtg, task = load_shard(a, b)
        psnr, _ = tree.run_batch(mat, rtg, task, first_image=a)      # per-image sampling streams follow the GLOBAL image index
        local = psnr.to(dev).float()
        rollouts = float(tree.last_stats["rollouts"])
    else:
        local, rollouts = torch.zeros((0, 1), device=dev), 0.0
    stat = torch.tensor([time.perf_counter() - t0, rollouts], dtype=torch.float64, device=dev)
    if world > 1:
        secs = stat[:1].clone()
        dist.all_reduce(secs, op=dist.ReduceOp.MAX, group=group)
        dist.all_reduce(stat, op=dist.ReduceOp.SUM, group=group)
        stat[0] = secs[0]
    return sharding.gather_per_slice(local, total, group).cpu(), float(stat[1]), float(stat[0])
