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
