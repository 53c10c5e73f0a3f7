"""N>1 path on CPU: two gloo ranks each own a contiguous shard of the slices, step them independently (no
data-path collective) and all_gather the per-slice PSNR; the result equals the single-process run.  The CPU
oracle stands in for the per-rank engine here (tests may use it; the product path on GPUs uses the HIP engine
with the same sharding code, bench.py)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dt4image_restoration_amd import sharding, synthetic, weights
from oracle import pnp_oracle as O

TOTAL, H, W, ITERS = 3, 32, 32, 2        # ragged on purpose: shards of 2 and 1 slices


def _episode(first, count):
    sd = O.torch_weights(weights.generate_unet_weights(0, "unit_gain"))
    data = synthetic.make_problem(count, H, W, seed=42, first_slice=first)
    mu, sg = synthetic.param_table(TOTAL, ITERS, seed=5)
    st, _ = O.run_episode(sd, data, mu[first:first + count], sg[first:first + count], ITERS, record_psnr=False)
    return O.psnr(st["x"], st["gt"])[:, 0]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b = sharding.shard_range(TOTAL, rank, world)
    local = _episode(a, b - a)
    full = sharding.gather_per_slice(local, TOTAL)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), full.numpy())
    dist.destroy_process_group()


def test_two_rank_sharded_episode_equals_single_process(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    want = _episode(0, TOTAL).numpy()
    for r in range(2):
        got = np.load(tmp_path / f"r{r}.npy")
        assert got.shape == (TOTAL,)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-4)


def test_gather_is_identity_without_a_process_group():
    t = torch.arange(5.0)
    assert torch.equal(sharding.gather_per_slice(t, 5), t)


# ---- the product entry point for configs[2] (drivers/sharded.run_sharded_greedy) under 2 gloo ranks ----------------------
GT, GH, GSTEPS = 3, 128, 4


def _greedy_parts():
    from dt4image_restoration_amd import data as D
    from dt4image_restoration_amd.drivers.greedy import GreedyEvaluator
    from dt4image_restoration_amd.policy import DecisionTransformer, DecisionTransformerConfig

    class OracleEnv:                                        # PnPEnv-shaped wrapper over the CPU oracle (tests only)
        def __init__(self):
            self.sd = O.torch_weights(weights.generate_unet_weights(0, "unit_gain"))

        def reset(self, mat, device):
            return O.reset({k: (v.numpy() if hasattr(v, "numpy") else v) for k, v in mat.items()})

        def step(self, st, action):
            with torch.no_grad():
                return O.admm_step(self.sd, st, action["mu"], action["sigma_d"], action["T"])

        def compute_reward(self, x, gt):
            return O.psnr(x, gt)

    m = DecisionTransformer(DecisionTransformerConfig(block_size=18, n_embeds=9, mode="norm"))
    m.load_state_dict(weights.generate_policy_weights(m, 7, t_bias=-1.0, head_gain=8.0))
    ev = GreedyEvaluator(m, OracleEnv(), max_timesteps=GSTEPS, block_size=18, device_type="cpu", sync_every=2)

    def load_shard(a, b):
        p = synthetic.make_problem(b - a, GH, GH, accel=4.0, seed=77, first_slice=a)
        mat = {k: torch.from_numpy(np.asarray(v)) for k, v in p.items()}
        return mat, torch.full((b - a,), D.normalised_rtg(10.0)), torch.full((b - a,), 4)
    return ev, load_shard


def _greedy_worker(rank, world, port, out_dir):
    from dt4image_restoration_amd.drivers.sharded import run_sharded_greedy
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ev, load_shard = _greedy_parts()
    r = run_sharded_greedy(ev, GT, load_shard)
    assert r.local_range == sharding.shard_range(GT, rank, world)
    np.savez(os.path.join(out_dir, f"g{rank}.npz"), reward=r.reward.numpy(), stop=r.stop_time.numpy(), init=r.initial_reward.numpy())
    dist.destroy_process_group()


def test_two_rank_sharded_greedy_rollout_equals_single_process(tmp_path):
    """configs[2]'s code path: DT-driven rollout per rank on its shard + the PSNR / stop-iteration gather, 2 gloo ranks with
    ragged shards (2 + 1 slices), equals the unsharded run of the same entry point."""
    from dt4image_restoration_amd.drivers.sharded import run_sharded_greedy
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_greedy_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    ev, load_shard = _greedy_parts()
    want = run_sharded_greedy(ev, GT, load_shard)          # no process group: world size 1
    assert want.reward.shape == (GT, 1) and want.stop_time.shape == (GT,)
    for r in range(2):
        got = np.load(tmp_path / f"g{r}.npz")
        np.testing.assert_array_equal(got["stop"], want.stop_time.numpy())
        # FLOAT TOLERANCE: a slice alone or in a batch of 2 takes another oneDNN blocking of the same f32 convolutions
        np.testing.assert_allclose(got["reward"], want.reward.numpy(), rtol=0, atol=1e-3)
        np.testing.assert_allclose(got["init"], want.initial_reward.numpy(), rtol=0, atol=1e-4)


# ---- BASELINE configs[3] over several ranks (drivers/sharded.run_sharded_mcts; reference loop mcts.py:212-258) ---------------
MT_TOTAL, MT_STEPS, MT_ROUNDS, MT_K = 3, 4, 3, 3


def _mcts_parts():
    import torch.nn.functional as F
    from dt4image_restoration_amd import data as D
    from dt4image_restoration_amd.drivers.greedy import GreedyEvaluator
    from dt4image_restoration_amd.drivers.mcts import MCTS
    from dt4image_restoration_amd.policy import DecisionTransformer, DecisionTransformerConfig

    class OracleEnv:
        """PnPEnv-shaped wrapper over the CPU oracle with the HIP env's state conventions (x real from reset on)."""
        def __init__(self):
            self.sd = O.torch_weights(weights.generate_unet_weights(0, "unit_gain"))

        def reset(self, mat, device):
            st = O.reset({k: (v.numpy() if hasattr(v, "numpy") else v) for k, v in mat.items()})
            st["x"] = st["x"].real.clone()
            return st

        def step(self, st, action):
            with torch.no_grad():
                return O.admm_step(self.sd, st, action["mu"], action["sigma_d"], action["T"])

        def compute_reward(self, x, gt):
            return O.psnr(x, gt)

    def scorer(states):
        x = states["x"]
        return 1.0 / (1e-3 + ((x - F.avg_pool2d(x, 3, 1, 1)) ** 2).mean(dim=(1, 2, 3)))

    m = DecisionTransformer(DecisionTransformerConfig(block_size=18, n_embeds=9, mode="norm"))
    m.load_state_dict(weights.generate_policy_weights(m, 7, t_bias=-1.0, head_gain=12.0))
    ev = GreedyEvaluator(m, OracleEnv(), max_timesteps=MT_STEPS, block_size=18, device_type="cpu")
    tree = MCTS(ev, scorer, n_children=MT_K, rounds=MT_ROUNDS, seed=3)

    def load_shard(a, b):
        p = synthetic.make_problem(b - a, GH, GH, accel=4.0, seed=91, first_slice=a)
        mat = {k: torch.from_numpy(np.asarray(v)) for k, v in p.items()}
        return mat, torch.full((b - a,), D.normalised_rtg(10.0)), torch.full((b - a,), 4)
    return tree, load_shard


def _mcts_worker(rank, world, port, out_dir):
    from dt4image_restoration_amd.drivers.sharded import run_sharded_mcts
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tree, load_shard = _mcts_parts()
    psnr, rollouts, secs = run_sharded_mcts(tree, MT_TOTAL, load_shard)
    np.savez(os.path.join(out_dir, f"m{rank}.npz"), psnr=psnr.numpy(), rollouts=rollouts, secs=secs)
    dist.destroy_process_group()


def test_two_rank_sharded_tree_search_equals_single_process(tmp_path):
    """configs[3]'s multi-rank entry point: every rank searches the trees of its own images (ragged shards 2 + 1), the
    per-image PSNR of the best program is gathered; image i samples from the stream of its GLOBAL index, so the job equals
    the unsharded search image for image."""
    from dt4image_restoration_amd.drivers.sharded import run_sharded_mcts
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_mcts_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    tree, load_shard = _mcts_parts()
    want, want_rollouts, _ = run_sharded_mcts(tree, MT_TOTAL, load_shard)      # no process group: world size 1
    assert want.shape == (MT_TOTAL, 1) and want_rollouts >= MT_TOTAL
    for r in range(2):
        got = np.load(tmp_path / f"m{r}.npz")
        assert float(got["rollouts"]) == want_rollouts                          # summed over the ranks
        # FLOAT TOLERANCE: an image alone or beside batch mates takes another oneDNN blocking of the same f32 convolutions
        np.testing.assert_allclose(got["psnr"], want.numpy(), rtol=0, atol=2e-3)
