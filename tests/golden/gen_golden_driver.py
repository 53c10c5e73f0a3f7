#!/usr/bin/env python3
"""G7: golden vectors for the greedy decision-transformer driver, produced by RUNNING THE REFERENCE's own
`Evaluator.get_initial_policy_setup` + `run_greedy` (evaluation/eval.py:62-100,189-220) with its own
`DecisionTransformer`, `PnPEnv.step` and `UNetDenoiser2D` on the CPU.  Build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_driver.py

Same import route as gen_golden.py plus an empty placeholder for `h5py` (dataset/datasets.py:6 imports it for the
training set only).  Policy and denoiser weights come from this repo's deterministic generators and are loaded
through the reference's own `load_state_dict` paths.  Output: inputs' seeds and the reference's outputs (DATA only).
"""
from __future__ import annotations

import os
import sys
import tempfile
import types
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from dt4image_restoration_amd import data as D, synthetic, weights  # noqa: E402
from gen_golden import import_reference, ref_denoiser  # noqa: E402

# (name, policy seed, stop-logit bias, action-head gain, rtg target, task)
CASES = [("full30", 0, -3.0, 8.0, 10.0, "4x_10"),
         ("stop_now", 1, +3.0, 8.0, 10.0, "4x_10"),
         ("stop_mid", 7, 0.0, 12.0, 10.0, "4x_15"),           # found by the search below (kept for re-runs: pseed=None)
         # the policy's first state token as the reference's datasets build it from a file whose stored x0 is the raw
         # zero-filled reconstruction (negative pixels): UNclipped Re x0 (datasets.py:162,201), while the environment gets
         # the clipped copy (:160,199)
         ("raw_first", 0, -3.0, 8.0, 10.0, "4x_10")]
RAW_FIRST = {"raw_first"}


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    PnPEnv, torch_psnr, UNet, UNetDenoiser2D, fft, ifft = import_reference()
    sys.modules.setdefault("h5py", types.ModuleType("h5py"))
    from evaluation.eval import Evaluator
    from transformer.decision_transformer import DecisionTransformer, DecisionTransformerConfig

    sd_unet = weights.generate_unet_weights(0, "unit_gain")
    env = PnPEnv.__new__(PnPEnv)
    env.denoiser = ref_denoiser(UNetDenoiser2D, sd_unet)
    problem = synthetic.make_problem(1, 128, 128, accel=4.0, sigma_n=10.0 / 255.0, seed=1234)

    def run_case(pseed, t_bias, gain, rtg_target, task, raw_first=False):
        model = DecisionTransformer(DecisionTransformerConfig(block_size=18, n_embeds=9, mode="norm"))
        sd = weights.generate_policy_weights(model, pseed, t_bias=t_bias, head_gain=gain)
        with tempfile.NamedTemporaryFile(suffix=".pt", delete=False) as f:
            path = f.name
        torch.save(sd, path)
        try:
            ev = Evaluator(model=model, model_path=path, action_dim=3, max_timesteps=30, env=env, compile=False,
                           device_type="cpu", block_size=18, rtg_target=rtg_target)
        finally:
            os.unlink(path)
        # what EvaluationOptimalDataset.__getitem__ + DataLoader(batch_size=1) hand over (datasets.py:181-207)
        x0 = problem["ATy0"][0] if raw_first else problem["x0"][0]
        states = torch.from_numpy(x0[..., 0].reshape(1, 1, -1).copy())
        rtg = torch.tensor([[[D.normalised_rtg(rtg_target)]]], dtype=torch.float32)
        task_t = torch.tensor([[D.OPTIMAL_TASKS.index(task)]])
        mat = {k: torch.from_numpy(np.asarray(v)) for k, v in problem.items()}
        Ts = []
        orig_step = env.step

        def logging_step(st, act):
            Ts.append([float(act["T"]), float(act["sigma_d"]), float(act["mu"])])
            return orig_step(st, act)

        env.step = logging_step
        try:
            with torch.no_grad():
                model_inputs, env_inputs = ev.get_initial_policy_setup((states, rtg, None, task_t), mat)
                es, ea, er, _, et, ek = model_inputs
                st, pred_rtg, _, action_dict = env_inputs
                old = env.compute_reward(st["x"].real.squeeze(dim=0), st["gt"])
                reward, time, x = ev.run_greedy(st, pred_rtg, 1, action_dict, es, ea, er, et, ek)
        finally:
            env.step = orig_step
        return {"handed": np.array(Ts), "eval_actions": ea.numpy()[0], "eval_rtg": er.numpy()[0, :, 0], "time": np.array(time),
                "reward": np.array(float(reward)), "old_reward": np.array(float(old)),
                "x": np.array(x.real.numpy() if x.is_complex() else x.numpy())}

    out = {}
    for name, pseed, t_bias, gain, rtg_target, task in CASES:
        if pseed is None:      # search for a policy whose stop logit crosses 0.5 mid-episode with a clear margin
            found = None
            for s in range(2, 60):
                for tb in (-0.6, -0.3, 0.0):
                    r = run_case(s, tb, 12.0, rtg_target, task)
                    T = r["handed"][:, 0]
                    if 4 <= int(r["time"]) <= 24 and np.abs(T - 0.5).min() > 0.04:
                        found = (s, tb, 12.0, r)
                        break
                if found:
                    break
            assert found, "no mid-episode stop found"
            pseed, t_bias, gain, r = found
        else:
            r = run_case(pseed, t_bias, gain, rtg_target, task, raw_first=name in RAW_FIRST)
        print(name, "seed", pseed, "t_bias", t_bias, "stop time", int(r["time"]), "reward", float(r["reward"]),
              "min |T-0.5|", float(np.abs(r["handed"][:, 0] - 0.5).min()))
        for k, v in r.items():
            out[f"{name}_{k}"] = v
        out[f"{name}_cfg"] = np.array([pseed, t_bias, gain, rtg_target, D.OPTIMAL_TASKS.index(task)], dtype=np.float64)
    # policy forward on fixed inputs (all three call modes of decision_transformer.py:212-263), policy seed 7
    model = DecisionTransformer(DecisionTransformerConfig(block_size=18, n_embeds=9, mode="norm")).eval()
    model.load_state_dict(weights.generate_policy_weights(model, 7, t_bias=0.0, head_gain=12.0))
    b, t = 2, 6
    rtg = torch.from_numpy((synthetic.hash_uniform(70, 1, b * t).reshape(b, t, 1) + 1) * 0.5)
    st = torch.from_numpy((synthetic.hash_uniform(70, 2, b * t * 16384).reshape(b, t, 16384) + 1) * 0.5)
    ts = torch.arange(t).reshape(1, t, 1).repeat(b, 1, 1)
    task = torch.tensor([[3], [7]]).repeat(1, t)
    act = torch.from_numpy((synthetic.hash_uniform(70, 3, b * t * 3).reshape(b, t, 3) + 1) * 0.5)
    with torch.no_grad():
        out["policy_noact"] = model(rtg, st, ts, task, actions=None)[0].numpy()
        out["policy_act"] = model(rtg, st, ts, task, act, eval_actions=True)[0].numpy()
        out["policy_rtg"] = model(rtg, st, ts, task, act, eval_rtg=True).numpy()
        out["policy_train"] = model(rtg, st, ts, task, act)[0].numpy()
    np.savez_compressed(os.path.join(HERE, "g7_greedy.npz"), **out)
    print("written g7_greedy.npz")


if __name__ == "__main__":
    main()
