"""Greedy decision-transformer driver and the `.mat` evaluation format (8f #2).

G7 (tests/golden/g7_greedy.npz) was produced by the reference's own Evaluator.get_initial_policy_setup + run_greedy with
its own DecisionTransformer / PnPEnv / U-Net on the CPU.  The CPU tests drive the SAME driver code with an oracle-backed
env; the GPU test drives it with the HIP PnPEnv."""
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from dt4image_restoration_amd import data as D, synthetic, weights
from dt4image_restoration_amd.drivers.greedy import GreedyEvaluator
from dt4image_restoration_amd.policy import DecisionTransformer, DecisionTransformerConfig, policy_observation
from oracle import pnp_oracle as O

CASES = ["full30", "stop_now", "stop_mid", "raw_first"]


def _first_state(problem, case):
    """What the golden generator handed the reference as the policy's first state token: the clipped Re x0 for the three
    round-1 cases; for `raw_first` the UNclipped Re x0 the reference's datasets read from a file that stores the raw
    zero-filled reconstruction (datasets.py:162,201) - which is also the driver's default when the batch carries `x0_raw`."""
    if case == "raw_first":
        assert float(problem["x0_raw"].min()) < -0.01          # the fixture really has negative pixels
        return None                                            # default path: mat['x0_raw']
    return torch.from_numpy(problem["x0"][..., 0].copy())


def _policy(cfg):
    pseed, t_bias, gain = int(cfg[0]), float(cfg[1]), float(cfg[2])
    m = DecisionTransformer(DecisionTransformerConfig(block_size=18, n_embeds=9, mode="norm"))
    m.load_state_dict(weights.generate_policy_weights(m, pseed, t_bias=t_bias, head_gain=gain))
    return m


class OracleEnv:
    """PnPEnv-shaped wrapper over the CPU oracle (tests only)."""

    def __init__(self):
        self.sd = O.torch_weights(weights.generate_unet_weights(0, "unit_gain"))

    def reset(self, mat, device):
        return O.reset({k: (v.numpy() if hasattr(v, "numpy") else v) for k, v in mat.items()})

    def step(self, st, action):
        with torch.no_grad():
            st, done = O.admm_step(self.sd, st, action["mu"], action["sigma_d"], action["T"])
        return st, done

    def compute_reward(self, x, gt):
        return O.psnr(x, gt)


def test_policy_forward_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "g7_greedy.npz"))
    m = DecisionTransformer(DecisionTransformerConfig(block_size=18, n_embeds=9, mode="norm"))
    m.load_state_dict(weights.generate_policy_weights(m, 7, t_bias=0.0, head_gain=12.0))
    b, t = 2, 6
    rtg = torch.from_numpy((synthetic.hash_uniform(70, 1, b * t).reshape(b, t, 1) + 1) * 0.5)
    st = torch.from_numpy((synthetic.hash_uniform(70, 2, b * t * 16384).reshape(b, t, 16384) + 1) * 0.5)
    ts = torch.arange(t).reshape(1, t, 1).repeat(b, 1, 1)
    task = torch.tensor([[3], [7]]).repeat(1, t)
    act = torch.from_numpy((synthetic.hash_uniform(70, 3, b * t * 3).reshape(b, t, 3) + 1) * 0.5)
    with torch.no_grad():
        np.testing.assert_allclose(m(rtg, st, ts, task, actions=None)[0].numpy(), g["policy_noact"], atol=1e-6)
        pa, ad = m(rtg, st, ts, task, act, eval_actions=True)
        np.testing.assert_allclose(pa.numpy(), g["policy_act"], atol=1e-6)
        assert list(ad.keys()) == ["T", "sigma_d", "mu"] and float(ad["sigma_d"].max()) <= 70 / 255 + 1e-6
        np.testing.assert_allclose(m(rtg, st, ts, task, act, eval_rtg=True).numpy(), g["policy_rtg"], atol=1e-6)
        np.testing.assert_allclose(m(rtg, st, ts, task, act)[0].numpy(), g["policy_train"], atol=1e-6)
    flex = DecisionTransformer(DecisionTransformerConfig(block_size=18, n_embeds=6, mode="flex"))
    assert list(flex.action_range.keys()) == ["mu", "sigma_d", "T"]


@pytest.mark.parametrize("case", CASES)
def test_greedy_driver_on_oracle_env_matches_reference_rollout(golden_dir, case):
    g = np.load(os.path.join(golden_dir, "g7_greedy.npz"))
    cfg = g[f"{case}_cfg"]
    problem = synthetic.make_problem(1, 128, 128, accel=4.0, sigma_n=10.0 / 255.0, seed=1234)
    ev = GreedyEvaluator(_policy(cfg), OracleEnv(), max_timesteps=30, block_size=18, device_type="cpu")
    res = ev.run({k: torch.from_numpy(np.asarray(v)) for k, v in problem.items()},
                 rtg=torch.tensor([D.normalised_rtg(cfg[3])]), task=torch.tensor([int(cfg[4])]),
                 first_state=_first_state(problem, case))
    stop = int(g[f"{case}_time"])
    assert int(res.stop_time[0]) == stop
    handed = g[f"{case}_handed"]                      # (T, sigma_d, mu) the reference handed to env.step, per call
    # actions written before the stop are the same tokens the reference wrote
    np.testing.assert_allclose(res.actions[0, :min(stop, 30)].numpy(), g[f"{case}_eval_actions"][:min(stop, 30)], atol=2e-5)
    np.testing.assert_allclose(res.actions[0, :len(handed)].numpy()[:, 0], handed[:, 0], atol=2e-5)
    assert abs(float(res.reward[0]) - float(g[f"{case}_reward"])) < 1e-3
    assert abs(float(res.initial_reward[0]) - float(g[f"{case}_old_reward"])) < 1e-4
    xr = res.x.real if res.x.is_complex() else res.x          # the oracle keeps complex x0 until the first step, like the reference
    np.testing.assert_allclose(xr.numpy().reshape(128, 128), g[f"{case}_x"].reshape(128, 128), atol=5e-5)


def test_policy_observation_downsamples_larger_slices():
    x = torch.rand(2, 1, 256, 256)
    ob = policy_observation(x)
    assert ob.shape == (2, 16384)
    np.testing.assert_allclose(ob[0].reshape(128, 128)[3, 5], x[0, 0, 6:8, 10:12].mean(), rtol=1e-6)
    assert torch.equal(policy_observation(torch.ones(1, 1, 128, 128)), torch.ones(1, 16384))


def test_mat_round_trip_and_task_tokens(tmp_path):
    p = synthetic.make_problem(3, 32, 32, seed=8)
    for i, tag in enumerate(("4_10", "4_10", "8_5")):
        D.save_mat(str(tmp_path / f"img{'abc'[i]}_{tag}_x.mat"), p, i)
    batch, tasks = D.load_dir(str(tmp_path))
    for k in ("x0", "y0", "ATy0", "gt", "x0_raw"):
        assert np.array_equal(batch[k], p[k]) and batch[k].dtype == np.float32
    # the file stores the raw zero-filled reconstruction; the env's x0 is clipped (datasets.py:160), the policy's first
    # state token is not (:162)
    assert float(batch["x0_raw"].min()) < 0.0 and float(batch["x0"].min()) == 0.0
    assert np.array_equal(batch["x0"][..., 0], np.clip(batch["x0_raw"], 0, None))
    assert np.array_equal(batch["mask"], p["mask"])
    assert tasks == ["4x_10", "4x_10", "8x_5"]
    assert D.task_tokens(tasks).tolist() == [4, 4, 6]
    assert D.task_tokens(tasks, flex_target=3.5).tolist() == [2, 2, 2]
    assert abs(D.normalised_rtg(10.0) - (10 + 1.08) / (16.6 + 1.08)) < 1e-12
    with pytest.raises(ValueError):
        D.task_from_filename("no_task_here.mat")
    with pytest.raises(FileNotFoundError):
        D.load_dir(str(tmp_path / "nothing") if os.makedirs(tmp_path / "nothing") is None else "")


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_greedy_driver_on_hip_env_matches_reference_rollout(golden_dir, case):
    from dt4image_restoration_amd.denoiser import UNetDenoiser2D
    from dt4image_restoration_amd.env import PnPEnv
    g = np.load(os.path.join(golden_dir, "g7_greedy.npz"))
    cfg = g[f"{case}_cfg"]
    problem = synthetic.make_problem(1, 128, 128, accel=4.0, sigma_n=10.0 / 255.0, seed=1234)
    env = PnPEnv(30, UNetDenoiser2D.seeded(0, "unit_gain"), "cuda")
    ev = GreedyEvaluator(_policy(cfg), env, max_timesteps=30, block_size=18, device_type="cuda")
    res = ev.run({k: torch.from_numpy(np.asarray(v)) for k, v in problem.items()},
                 rtg=torch.tensor([D.normalised_rtg(cfg[3])]), task=torch.tensor([int(cfg[4])]),
                 first_state=_first_state(problem, case))
    stop = int(g[f"{case}_time"])
    assert int(res.stop_time[0]) == stop
    handed = g[f"{case}_handed"]
    # FLOAT TOLERANCE: the policy sees the HIP engine's images (f32, ~1e-6 from the reference's); its sigmoid outputs
    # move by < 1e-4
    np.testing.assert_allclose(res.actions[0, :len(handed)].numpy()[:, 0], handed[:, 0], atol=2e-4)
    assert abs(float(res.reward[0]) - float(g[f"{case}_reward"])) < 0.01       # north_star PSNR tolerance
    np.testing.assert_allclose(res.x.cpu().numpy().reshape(128, 128), g[f"{case}_x"].reshape(128, 128), atol=2e-4)


@pytest.mark.gpu
def test_greedy_batch_equals_single_slice_rollouts():
    """4 slices with different stop behaviour rolled out as ONE batch == each rolled out alone."""
    from dt4image_restoration_amd.denoiser import UNetDenoiser2D
    from dt4image_restoration_amd.env import PnPEnv
    den = UNetDenoiser2D.seeded(0, "unit_gain")
    m = DecisionTransformer(DecisionTransformerConfig(block_size=18, n_embeds=9, mode="norm"))
    m.load_state_dict(weights.generate_policy_weights(m, 7, t_bias=0.0, head_gain=12.0))
    problem = synthetic.make_problem(4, 128, 128, accel=4.0, seed=77)
    rtg = torch.tensor([D.normalised_rtg(v) for v in (10.0, 4.0, 14.0, 8.0)])
    task = torch.tensor([5, 4, 5, 3])
    mat = {k: torch.from_numpy(np.asarray(v)) for k, v in problem.items()}
    full = GreedyEvaluator(m, PnPEnv(30, den, "cuda"), device_type="cuda").run(mat, rtg, task)
    for i in range(4):
        one_mat = {k: (v[i:i + 1] if k != "mask" else v) for k, v in mat.items()}
        one = GreedyEvaluator(m, PnPEnv(30, den, "cuda"), device_type="cuda").run(one_mat, rtg[i:i + 1], task[i:i + 1])
        assert int(one.stop_time[0]) == int(full.stop_time[i])
        assert abs(float(one.reward[0]) - float(full.reward[i])) < 1e-3


@pytest.mark.gpu
def test_mcts_mechanics_with_stub_scorer():
    """Tree mechanics with a deterministic positive scorer (smoothness of x; ARNIQA scores are positive too and the
    reference's max-backup starts from reward 0) and a fixed sampling seed: reproducible, the
    tree grows by n_children per expanded node, rewards back up as maxima, snapshots are not aliased."""
    import torch.nn.functional as F
    from dt4image_restoration_amd.denoiser import UNetDenoiser2D
    from dt4image_restoration_amd.drivers.mcts import MCTS, select_p_ucb
    from dt4image_restoration_amd.env import PnPEnv
    den = UNetDenoiser2D.seeded(0, "unit_gain")
    m = DecisionTransformer(DecisionTransformerConfig(block_size=18, n_embeds=9, mode="norm"))
    m.load_state_dict(weights.generate_policy_weights(m, 7, t_bias=-1.0, head_gain=12.0))

    def scorer(states):
        x = states["x"]
        return 1.0 / (1e-3 + ((x - F.avg_pool2d(x, 3, 1, 1)) ** 2).mean(dim=(1, 2, 3)))

    problem = synthetic.make_problem(1, 128, 128, accel=4.0, seed=5)
    mat = {k: torch.from_numpy(np.asarray(v)) for k, v in problem.items()}

    def search(seed):
        ev = GreedyEvaluator(m, PnPEnv(30, den, "cuda"), max_timesteps=8, device_type="cuda")
        return MCTS(ev, scorer, n_children=3, rounds=4, seed=seed).run(mat, torch.tensor([D.normalised_rtg(10.0)]),
                                                                      torch.tensor([4]))
    p1, root1 = search(3)
    p2, root2 = search(3)
    assert float(p1) == float(p2)                                  # seeded -> reproducible
    assert len(root1.children) == 3 and root1.visits == 4
    n_nodes, stack = 0, [root1]
    while stack:
        nd = stack.pop()
        n_nodes += 1
        stack += nd.children
        for c in nd.children:
            assert c.time == nd.time + 1 and c.reward <= nd.reward + 1e-12       # max-backup
            assert c.snap["x"].data_ptr() != nd.snap["x"].data_ptr()            # no aliasing of node states
    assert n_nodes == 1 + 3 * 4                                                  # one expansion per round
    assert 15.0 < float(p1) < 45.0
    assert select_p_ucb(root1, root1.children) in root1.children


@pytest.mark.gpu
def test_mcts_at_256_with_five_children():
    """BASELINE configs[3] geometry: 256x256 slice, the reference's 5 children per expansion (mcts.py:100) stepped as ONE
    batch of 5 through the engine; seeded -> reproducible; the returned PSNR is that of the best leaf's image."""
    import torch.nn.functional as F
    from dt4image_restoration_amd.denoiser import UNetDenoiser2D
    from dt4image_restoration_amd.drivers.mcts import MCTS
    from dt4image_restoration_amd.env import PnPEnv
    den = UNetDenoiser2D.seeded(0, "unit_gain")
    m = DecisionTransformer(DecisionTransformerConfig(block_size=18, n_embeds=9, mode="norm"))
    m.load_state_dict(weights.generate_policy_weights(m, 7, t_bias=-1.0, head_gain=12.0))

    def scorer(states):
        x = states["x"]
        return 1.0 / (1e-3 + ((x - F.avg_pool2d(x, 3, 1, 1)) ** 2).mean(dim=(1, 2, 3)))

    problem = synthetic.make_problem(1, 256, 256, accel=4.0, seed=9)
    mat = {k: torch.from_numpy(np.asarray(v)) for k, v in problem.items()}

    def search():
        ev = GreedyEvaluator(m, PnPEnv(30, den, "cuda"), max_timesteps=6, device_type="cuda")
        return MCTS(ev, scorer, n_children=5, rounds=3, seed=11).run(mat, torch.tensor([D.normalised_rtg(10.0)]),
                                                                    torch.tensor([4]))
    p1, root1 = search()
    p2, _ = search()
    assert float(p1) == float(p2)
    assert len(root1.children) == 5 and root1.visits == 3
    assert all(c.snap["x"].shape == (1, 1, 256, 256) for c in root1.children)
    assert 15.0 < float(p1) < 45.0


@pytest.mark.gpu
def test_cli_subcommands_run_on_synthetic_data():
    from dt4image_restoration_amd import cli
    ev = cli.main(["--block_size", "18", "--n_embeds", "9", "--limit", "2", "eval", "--rtg", "10", "--max_timesteps", "5"])
    assert len(ev) == 2 and all(20 < e["psnr"] < 45 and e["n"] == 2 for e in ev)
    fx = cli.main(["--block_size", "18", "--n_embeds", "6", "--limit", "2", "flex", "--max_timesteps", "3"])
    assert [f["rtg_target"] for f in fx] == [1.5, 3, 3.5, 4, 4.5]
    mc = cli.main(["--block_size", "18", "--n_embeds", "9", "--limit", "1", "mcts", "--rtg", "5", "--max_timesteps", "4",
                   "--rollouts", "2"])
    assert len(mc) == 2 and all("mcts_psnr" in m for m in mc)


def test_cli_rejects_train_and_requires_mode():
    from dt4image_restoration_amd import cli
    with pytest.raises(SystemExit):
        cli.main(["--block_size", "18", "--n_embeds", "9", "train"])
    with pytest.raises(SystemExit):
        cli.main(["--block_size", "18", "--n_embeds", "9"])


@pytest.mark.gpu
def test_cli_eval_on_a_directory_of_mat_files(tmp_path):
    """SURVEY 8f #2 through the product path: `cli --data <dir>` reads the reference's on-disk evaluation format
    (datasets.py:135-207: x0/y0/ATy0 [...,2], mask, gt; task from the file name), runs the DT-driven rollout on the HIP env and
    agrees with the same rollout started from the in-memory problem.  The files store the raw zero-filled reconstruction
    (negative pixels), so the clip for the env and the UNclipped first policy token both matter."""
    from dt4image_restoration_amd import cli
    from dt4image_restoration_amd.denoiser import UNetDenoiser2D
    from dt4image_restoration_amd.env import PnPEnv
    p = synthetic.make_problem(3, 128, 128, accel=4.0, sigma_n=10.0 / 255.0, seed=61)
    d = tmp_path / "set_4_10"
    d.mkdir()
    for i in range(3):
        D.save_mat(str(d / ("img_" + "abc"[i] + "_4_10.mat")), p, i)      # the task regex takes the FIRST <digits>_<digits> of the name
    out = cli.main(["--block_size", "18", "--n_embeds", "9", "--data", str(d), "--limit", "0", "eval", "--rtg", "10",
                    "--max_timesteps", "6"])
    assert len(out) == 1 and out[0]["n"] == 3 and out[0]["set"] == str(d)
    # the same rollout from memory (the CLI's seeded stand-in weights: policy seed 0, t_bias -1, head gain 8)
    m = DecisionTransformer(DecisionTransformerConfig(block_size=18, n_embeds=9, mode="norm"))
    m.load_state_dict(weights.generate_policy_weights(m, 0, t_bias=-1.0, head_gain=8.0))
    ev = GreedyEvaluator(m, PnPEnv(30, UNetDenoiser2D.seeded(0), "cuda"), max_timesteps=6, device_type="cuda")
    r = ev.run({k: torch.from_numpy(np.asarray(v)) for k, v in p.items()}, torch.full((3,), D.normalised_rtg(10.0)),
               torch.tensor([D.OPTIMAL_TASKS.index("4x_10")] * 3))
    assert abs(out[0]["psnr"] - float(r.reward.mean())) < 1e-4
    assert abs(out[0]["psnr_increment"] - float((r.reward - r.initial_reward).mean())) < 1e-4
    assert 20 < out[0]["psnr"] < 45


def test_per_row_clock_rollout_equals_scalar_rollout():
    """`rollout_rows` (per-row clock, what the batched tree search uses) with equal start times is the scalar `rollout`
    bit for bit - on the oracle-backed env, 2 slices with different stop behaviour; and the cached state embeddings give
    the same actions as re-encoding the window in both forwards like the reference (eval.py:150-186)."""
    m = DecisionTransformer(DecisionTransformerConfig(block_size=18, n_embeds=9, mode="norm"))
    m.load_state_dict(weights.generate_policy_weights(m, 7, t_bias=0.0, head_gain=12.0))
    problem = synthetic.make_problem(2, 128, 128, accel=4.0, sigma_n=10.0 / 255.0, seed=1234)
    mat = {k: torch.from_numpy(np.asarray(v)) for k, v in problem.items()}
    rtg = torch.tensor([D.normalised_rtg(10.0), D.normalised_rtg(6.0)])
    task = torch.tensor([5, 4])

    def go(rows: bool, cache: bool):
        env = OracleEnv()
        ev = GreedyEvaluator(m, env, max_timesteps=9, block_size=18, device_type="cpu", cache_state_embeddings=cache)
        st = env.reset(mat, "cpu")
        ctx = ev.buffers(2, task)
        ev.observe(ctx, 0, policy_observation(torch.from_numpy(problem["x0_raw"]).float()))
        ctx.er[:, 0, 0] = rtg
        action, prtg = ev._initial(ctx)
        if rows:
            reward, stop = ev.rollout_rows(st, action, prtg, torch.tensor([1, 1]), ctx)
        else:
            reward, stop = ev.rollout(st, action, prtg, 1, ctx)
        return reward, stop, ctx.ea.clone()

    r0, s0, a0 = go(False, True)
    r1, s1, a1 = go(True, True)
    assert torch.equal(s0, s1) and torch.equal(a0, a1) and torch.equal(r0, r1)
    r2, s2, a2 = go(False, False)
    assert torch.equal(s0, s2)
    # FLOAT TOLERANCE: the encoder runs on 2 images at a time instead of 12 (another blocking of the same f32 convolution)
    np.testing.assert_allclose(a0.numpy(), a2.numpy(), atol=1e-6)
    np.testing.assert_allclose(r0.numpy(), r2.numpy(), atol=1e-4)


@pytest.mark.gpu
def test_mcts_configs3_scale_batched_over_images():
    """BASELINE configs[3]: 256x256, 64 rollouts per image, the reference's 5 children per expansion - searched for 4 images
    AT ONCE (one tree per image; per round one policy call, one env.step on 4 x 5 children and one per-row-clock rollout of 4
    rows), stub scorer (ARNIQA is a network fetch), fixed seed.  Checks: reproducible; every image got 64 rollouts and a
    tree of 1 + 5 x 64 nodes; rewards back up as maxima; the reported PSNR is that of the best-scored rollout's image; and an
    image's search does not depend on its batch mates (per-image sampling streams): searched alone it finds the same tree
    size and (to rounding: another tile plan at batch 1) the same PSNR."""
    import torch.nn.functional as F
    from dt4image_restoration_amd.denoiser import UNetDenoiser2D
    from dt4image_restoration_amd.drivers.mcts import MCTS
    from dt4image_restoration_amd.env import PnPEnv
    den = UNetDenoiser2D.seeded(0, "unit_gain")
    m = DecisionTransformer(DecisionTransformerConfig(block_size=18, n_embeds=9, mode="norm"))
    m.load_state_dict(weights.generate_policy_weights(m, 7, t_bias=-1.0, head_gain=12.0))

    def scorer(states):
        x = states["x"]
        return 1.0 / (1e-3 + ((x - F.avg_pool2d(x, 3, 1, 1)) ** 2).mean(dim=(1, 2, 3)))

    B, R = 4, 64
    problem = synthetic.make_problem(B, 256, 256, accel=4.0, seed=9)
    mat = {k: torch.from_numpy(np.asarray(v)) for k, v in problem.items()}
    rtg = torch.full((B,), D.normalised_rtg(10.0))
    task = torch.full((B,), 4)

    def search(sel=None):
        ev = GreedyEvaluator(m, PnPEnv(30, den, "cuda"), max_timesteps=30, device_type="cuda", sync_every=4)
        tree = MCTS(ev, scorer, n_children=5, rounds=R, seed=11)
        mm = mat if sel is None else {k: (v[sel] if k != "mask" else v) for k, v in mat.items()}
        n = B if sel is None else len(sel)
        psnr, roots = tree.run_batch(mm, rtg[:n], task[:n])
        return psnr, roots, tree.last_stats

    p1, roots, stats = search()
    p2, _, _ = search()
    assert torch.equal(p1, p2)                                                    # seeded -> reproducible
    # a round whose selection ends on a node at the last time step expands nothing (mcts.py:240-241), so an image gets at most
    # R rollouts; the search spends its first ~29 rounds walking one path to the horizon like the reference's does
    assert stats["rounds"] == R and 29 * B <= stats["rollouts"] <= B * R
    print(f"\\nMCTS configs[3] geometry: {B} images x {R} rollouts at 256x256 in {stats['seconds']:.2f} s = "
          f"{stats['rollouts_per_s']:.1f} rollouts/s")
    for b in range(B):
        n_nodes, n_roll, best, stack = 0, 0, None, [roots[b]]
        while stack:
            nd = stack.pop()
            n_nodes += 1
            stack += nd.children
            if nd.rollout_reward is not None:
                n_roll += 1
                best = nd if best is None or nd.rollout_reward > best.rollout_reward else best
            for c in nd.children:
                assert c.time == nd.time + 1 and c.reward <= nd.reward + 1e-12       # max-backup
        assert n_nodes == 1 + 5 * n_roll and 29 <= n_roll <= R and roots[b].visits == R
        assert abs(roots[b].reward - best.rollout_reward) < 1e-9                     # the root holds the best rollout score
        assert 15.0 < float(p1[b]) < 45.0
    # per-image streams: image 0 alone (batch 1: another tile plan, rounding-level differences in the images)
    q, r1, s1 = search([0])
    assert s1["rounds"] == R and s1["rollouts"] >= 29
    assert abs(float(q[0]) - float(p1[0])) < 0.05


@pytest.mark.gpu
def test_graph_replayed_policy_equals_eager_policy():
    """`use_graphs`: the steady-state policy calls (state encoder + the two transformer forwards) replayed from hipGraphs over
    static window buffers are the same kernels on the same values as the eager calls: action sequences, stop times and final
    PSNR of a 12-step rollout (6 eager steps, then replayed ones, one slice stopping on the way) are identical, bit for bit."""
    from dt4image_restoration_amd.denoiser import UNetDenoiser2D
    from dt4image_restoration_amd.env import PnPEnv
    den = UNetDenoiser2D.seeded(0, "unit_gain")
    m = DecisionTransformer(DecisionTransformerConfig(block_size=18, n_embeds=9, mode="norm"))
    m.load_state_dict(weights.generate_policy_weights(m, 7, t_bias=-1.0, head_gain=12.0))
    problem = synthetic.make_problem(3, 128, 128, accel=4.0, seed=21)
    mat = {k: torch.from_numpy(np.asarray(v)) for k, v in problem.items()}
    rtg = torch.full((3,), D.normalised_rtg(10.0))
    task = torch.full((3,), 4)
    res = {}
    for graphs in (False, True):
        ev = GreedyEvaluator(m, PnPEnv(30, den, "cuda"), max_timesteps=12, device_type="cuda", sync_every=3, use_graphs=graphs)
        assert ev.use_graphs == graphs
        r1 = ev.run(mat, rtg, task)
        r2 = ev.run(mat, rtg, task)                                  # second episode on the same evaluator: graphs are reused
        assert torch.equal(r1.actions, r2.actions) and torch.equal(r1.reward, r2.reward)
        res[graphs] = r1
        if graphs:
            assert {k[0] for k in ev._graphs} == {"enc", "predict"}
    assert torch.equal(res[True].actions, res[False].actions)
    assert torch.equal(res[True].stop_time, res[False].stop_time)
    assert torch.equal(res[True].reward, res[False].reward)
    assert torch.equal(res[True].x.cpu(), res[False].x.cpu())


@pytest.mark.gpu
def test_cli_loads_checkpoint_files(tmp_path):
    """`cli --denoiser-ckpt --policy-ckpt` (the reference's hard-coded paths main.py:175,178 made options): the 56-key U-Net file
    and the 85-key policy file written with torch.save give exactly the run the same weights give when handed over in memory
    (the CLI's seeded defaults are those very weights)."""
    import collections
    from dt4image_restoration_amd import cli
    from dt4image_restoration_amd.denoiser import UNetDenoiser2D
    sd = weights.generate_unet_weights(0, "unit_gain")
    torch.save(collections.OrderedDict((k, torch.from_numpy(v.copy())) for k, v in sd.items()), tmp_path / "unet-nm.pt")
    m = DecisionTransformer(DecisionTransformerConfig(block_size=18, n_embeds=9, mode="norm"))
    torch.save(weights.generate_policy_weights(m, 0, t_bias=-1.0, head_gain=8.0), tmp_path / "model_experiment_1.pt")
    base = ["--block_size", "18", "--n_embeds", "9", "--limit", "2"]
    tail = ["eval", "--rtg", "10", "--max_timesteps", "5"]
    from_files = cli.main(base + ["--denoiser-ckpt", str(tmp_path / "unet-nm.pt"), "--policy-ckpt",
                                  str(tmp_path / "model_experiment_1.pt")] + tail)
    seeded = cli.main(base + tail)
    assert [e["psnr"] for e in from_files] == [e["psnr"] for e in seeded]
    assert [e["mean_stop_iteration"] for e in from_files] == [e["mean_stop_iteration"] for e in seeded]
    # and the denoiser object itself: file == mapping, bit for bit on the GPU
    x = torch.rand(2, 1, 64, 64, device="cuda")
    sg = torch.tensor([0.05, 0.1], device="cuda")
    a = UNetDenoiser2D(ckpt_path=str(tmp_path / "unet-nm.pt"))(x, sg)
    b = UNetDenoiser2D(state_dict=sd)(x, sg)
    assert torch.equal(a, b)


@pytest.mark.gpu
def test_dt_driven_episode_on_a_full_64x256_shard():
    """BASELINE configs[2]'s per-GPU workload through its product entry point (`run_sharded_greedy`, world size 1; the loop is
    eval.py:189-220): 64 slices of 256x256, 30 steps, the decision transformer choosing (T, sigma_d, mu) per slice and step.
    Clamp range, per-slice stop times in range with the engine's T clock agreeing, and two slices of the batch == the same two
    slices rolled out alone (another tile plan underneath: to rounding)."""
    from dt4image_restoration_amd.denoiser import UNetDenoiser2D
    from dt4image_restoration_amd.drivers.sharded import run_sharded_greedy
    from dt4image_restoration_amd.env import PnPEnv
    n, hw, steps = 64, 256, 30
    m = DecisionTransformer(DecisionTransformerConfig(block_size=18, n_embeds=9, mode="norm"))
    m.load_state_dict(weights.generate_policy_weights(m, 7, t_bias=-1.0, head_gain=8.0))
    den = UNetDenoiser2D.seeded(0, "unit_gain")
    keep = {}

    def loader(first_only=None):
        def load_shard(a, b):
            idx = list(range(a, b)) if first_only is None else first_only
            p = synthetic.make_problem(n, hw, hw, accel=4.0, sigma_n=10.0 / 255.0, seed=1234)
            mat = {k: torch.from_numpy(np.asarray(v if k == "mask" else v[idx])) for k, v in p.items()}
            return mat, torch.full((len(idx),), D.normalised_rtg(10.0)), torch.full((len(idx),), 4)
        return load_shard

    ev = GreedyEvaluator(m, PnPEnv(steps, den, "cuda"), max_timesteps=steps, device_type="cuda", sync_every=10)
    orig_run = ev.run

    def run_and_keep(*a, **k):
        keep["res"] = orig_run(*a, **k)
        return keep["res"]
    ev.run = run_and_keep
    r = run_sharded_greedy(ev, n, loader(), sync=torch.cuda.synchronize)
    assert r.local_range == (0, n) and r.reward.shape == (n, 1) and r.stop_time.shape == (n,)
    assert int(r.stop_time.min()) >= 1 and int(r.stop_time.max()) <= steps and r.steps == int(r.stop_time.max())
    x = keep["res"].x
    assert x.shape == (n, 1, hw, hw) and bool(torch.isfinite(x).all()) and float(x.min()) >= 0.0 and float(x.max()) <= 1.0
    assert bool(torch.isfinite(r.reward).all()) and float(r.reward.min()) > 15.0
    assert len(set(r.stop_time.tolist())) > 1                           # the policy stops slices at different iterations
    acts = keep["res"].actions                                          # [n, steps, 3] model order (T, sigma_d, mu)
    assert float(acts[..., 1].max()) <= 70.0 / 255.0 + 1e-6 and float(acts[..., 2].max()) <= 1.0 and float(acts.min()) >= 0.0
    sel = [5, 40]
    ev2 = GreedyEvaluator(m, PnPEnv(steps, den, "cuda"), max_timesteps=steps, device_type="cuda", sync_every=10)
    r2 = run_sharded_greedy(ev2, 2, loader(first_only=sel), sync=torch.cuda.synchronize)
    assert r2.stop_time.tolist() == r.stop_time[sel].tolist()
    # FLOAT TOLERANCE: batch 64 and batch 2 run different tile plans (f32 summation order); 30 policy-driven steps amplify it
    assert float((r2.reward - r.reward[sel]).abs().max()) < 5e-3


@pytest.mark.gpu
def test_pipelined_rollout_equals_the_plain_rollout():
    """`run_pipelined`: the batch as two sub-batches on two streams (each with its own engine replica, policy context and
    captured graphs), one's policy call under the other's env step - the same episode, slice for slice: stop iterations equal,
    actions and PSNR to the rounding of another tile plan (sub-batches of 3 and 4 slices are planned unlike a batch of 7).
    Twice in a row: the second run replays the graphs captured by the first."""
    from dt4image_restoration_amd.denoiser import UNetDenoiser2D
    from dt4image_restoration_amd.env import PnPEnv
    n, steps = 7, 12                                                   # ragged on purpose
    m = DecisionTransformer(DecisionTransformerConfig(block_size=18, n_embeds=9, mode="norm"))
    m.load_state_dict(weights.generate_policy_weights(m, 7, t_bias=-1.0, head_gain=8.0))
    den = UNetDenoiser2D.seeded(0, "unit_gain")
    p = synthetic.make_problem(n, 128, 128, accel=4.0, sigma_n=10.0 / 255.0, seed=404)
    mat = {k: torch.from_numpy(np.asarray(v)) for k, v in p.items()}
    rtg, task = torch.full((n,), D.normalised_rtg(10.0)), torch.full((n,), 4)
    plain = GreedyEvaluator(m, PnPEnv(steps, den, "cuda"), max_timesteps=steps, device_type="cuda", sync_every=4).run(mat, rtg, task)
    ev = GreedyEvaluator(m, PnPEnv(steps, den, "cuda"), max_timesteps=steps, device_type="cuda", sync_every=4)
    for _ in range(2):
        piped = ev.run_pipelined(mat, rtg, task, parts=2)
        torch.cuda.synchronize()
        assert piped.stop_time.tolist() == plain.stop_time.tolist()
        assert piped.x.shape == plain.x.shape and piped.actions.shape == plain.actions.shape
        # FLOAT TOLERANCE: f32 summation order of another tile plan, carried through up to 12 policy-driven steps
        assert float((piped.reward - plain.reward).abs().max()) < 2e-3
        assert float((piped.initial_reward - plain.initial_reward).abs().max()) < 1e-4
        np.testing.assert_allclose(piped.actions.numpy(), plain.actions.numpy(), rtol=0, atol=2e-4)
    assert len(set(plain.stop_time.tolist())) >= 1 and float(plain.x.min()) >= 0.0


@pytest.mark.gpu
def test_fused_attention_equals_the_explicit_form():
    """The GPU policy runs its causal attention as one fused kernel (F.scaled_dot_product_attention); the explicit
    scores / mask / softmax form of the reference (decision_transformer.py:60-75) is the CPU path.  Same outputs to f32 rounding."""
    from dt4image_restoration_amd import policy as P
    m = DecisionTransformer(DecisionTransformerConfig(block_size=18, n_embeds=9, mode="norm")).cuda()
    m.load_state_dict(weights.generate_policy_weights(m, 7, t_bias=0.0, head_gain=12.0))
    b, t = 5, 6
    g = torch.Generator().manual_seed(3)
    rtg, st = torch.rand(b, t, 1, generator=g).cuda(), torch.rand(b, t, 16384, generator=g).cuda()
    ts, task = torch.arange(t).reshape(1, t, 1).repeat(b, 1, 1).cuda(), torch.randint(0, 9, (b, 1), generator=g).repeat(1, t).cuda()
    act = torch.rand(b, t, 3, generator=g).cuda()
    with torch.no_grad():
        fused = m(rtg, st, ts, task, act)[0]
        P.FUSED_ATTENTION = False
        try:
            plain = m(rtg, st, ts, task, act)[0]
        finally:
            P.FUSED_ATTENTION = True
    # FLOAT TOLERANCE: the fused kernel sums the softmax in another order (f32, 18 tokens)
    assert float((fused - plain).abs().max()) < 2e-6


def test_node_pool_grows_on_demand_and_reports_its_needs():
    """The tree search's node storage starts small and doubles (rows kept), up to the bound of every round expanding every
    image; past it `alloc` refuses (drivers/mcts.NodePool)."""
    from dt4image_restoration_amd.drivers.mcts import NodePool
    pool = NodePool(2, 4, 4, "cpu", max_capacity=7)
    a = pool.alloc(2)
    st = {"x": torch.arange(32.0).reshape(2, 1, 4, 4), "z": torch.ones(2, 1, 4, 4, dtype=torch.complex64),
          "u": torch.zeros(2, 1, 4, 4, dtype=torch.complex64), "T": torch.tensor([0.1, 0.2])}
    pool.store(a, st)
    b = pool.alloc(3)                                      # grows: 2 -> 5 rows, the first two kept
    assert b.tolist() == [2, 3, 4] and pool.x.shape[0] >= 5
    assert torch.equal(pool.x[:2], st["x"].reshape(2, -1)) and torch.equal(pool.T[:2], st["T"])
    pool.alloc(2)
    with pytest.raises(RuntimeError, match="node pool exhausted"):
        pool.alloc(1)


@pytest.mark.gpu
def test_batched_tree_search_with_one_mask_per_image():
    """The reference reads a mask per .mat file (datasets.py:153-160): `run_batch` repeats a per-image mask [B,H,W] for the k
    children of each image like the images themselves.  Two images with DIFFERENT masks searched at once == each searched alone."""
    import torch.nn.functional as F
    from dt4image_restoration_amd.denoiser import UNetDenoiser2D
    from dt4image_restoration_amd.drivers.mcts import MCTS
    from dt4image_restoration_amd.env import PnPEnv
    den = UNetDenoiser2D.seeded(0, "unit_gain")
    m = DecisionTransformer(DecisionTransformerConfig(block_size=18, n_embeds=9, mode="norm"))
    m.load_state_dict(weights.generate_policy_weights(m, 7, t_bias=-1.0, head_gain=12.0))

    def scorer(states):
        x = states["x"]
        return 1.0 / (1e-3 + ((x - F.avg_pool2d(x, 3, 1, 1)) ** 2).mean(dim=(1, 2, 3)))

    p4 = synthetic.make_problem(1, 128, 128, accel=4.0, seed=21)
    p8 = synthetic.make_problem(1, 128, 128, accel=8.0, seed=22, first_slice=1)
    assert not np.array_equal(p4["mask"], p8["mask"])
    both = {k: np.concatenate([p4[k], p8[k]], axis=0) for k in ("x0", "y0", "ATy0", "gt", "x0_raw")}
    both["mask"] = np.stack([p4["mask"], p8["mask"]])      # [2, H, W]: one mask per image

    def search(prob, n, first):
        ev = GreedyEvaluator(m, PnPEnv(30, den, "cuda"), max_timesteps=6, device_type="cuda")
        mat = {k: torch.from_numpy(np.asarray(v)) for k, v in prob.items()}
        return MCTS(ev, scorer, n_children=3, rounds=3, seed=5).run_batch(mat, torch.full((n,), D.normalised_rtg(10.0)),
                                                                         torch.full((n,), 4), first_image=first)[0]
    together = search(both, 2, 0)
    alone = torch.cat([search(p4, 1, 0), search(p8, 1, 1)])
    # FLOAT TOLERANCE: batch 2 x 3 children and batch 1 x 3 run other tile plans (f32 summation order)
    assert float((together - alone).abs().max()) < 2e-3
