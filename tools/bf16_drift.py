#!/usr/bin/env python3
"""CPU experiment (oracle only): where does the bf16-operand mode's PSNR offset to the f32 reference come from, per layer
group and per operand, over a configs[4]-length episode?  Picks the layers whose WEIGHTS must carry a second bf16 term
(hi + lo) for the offset to stay under north_star's 0.01 dB at iteration 50.

    python tools/bf16_drift.py --size 128 --accel 4 --iters 53 --slices 2
    python tools/bf16_drift.py --size 512 --accel 8 --iters 53 --slices 2 --plans all,w2-l0,w2-l01,w2-all

Plans (comma separated): f32 | all (every Cin >= 32 conv rounds activations and weights to bf16) | act (activations only)
| w (weights only) | w2-l0 / w2-l01 / w2-l012 / w2-all (all, with two-term bf16 weights on the layers of levels 0 / 0-1 / ... / everywhere
= the engine's default) | w2-<layer+layer+...> explicit layer indices.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dt4image_restoration_amd import synthetic, unet_spec, weights  # noqa: E402
from oracle import pnp_oracle as O  # noqa: E402


def plan_for(name: str):
    L = unet_spec.UNET_LAYERS[1:27]
    if name == "f32":
        return False
    if name == "all":                       # one bf16 term per weight everywhere: the round-3 arithmetic (PNP_BF16_W1)
        return O.Bf16Plan(weight_terms=1)
    if name == "act":
        return O.Bf16Plan(weight_terms=0)
    if name == "w":
        return O.Bf16Plan(acts=False, weight_terms=1)
    if name == "lo-e4m3":                   # round 5 gate: the second weight term on fp8 e4m3 operands (weights pre-scaled per layer)
        return O.Bf16Plan(lo_format="e4m3")
    if name.startswith("w2-"):
        arg = name[3:]
        if arg == "all":                    # the engine's default since round 4
            return O.Bf16Plan()
        if arg.startswith("l"):
            lv = {int(c) for c in arg[1:]}
            idx = [l.index for l in L if l.level in lv]
        else:
            idx = [int(t) for t in arg.split("+")]
        return O.Bf16Plan(weight_terms=1, layer_terms={i: 2 for i in idx})
    raise SystemExit(f"unknown plan {name}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--accel", type=float, default=4.0)
    ap.add_argument("--iters", type=int, default=53)
    ap.add_argument("--slices", type=int, default=2)
    ap.add_argument("--plans", default="all,act,w,w2-l0,w2-l01,w2-l012,w2-all")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    torch.set_num_threads(8)
    n, h = args.slices, args.size
    sd = O.torch_weights(weights.generate_unet_weights(0, "unit_gain"))
    data = synthetic.make_problem(n, h, h, accel=args.accel, sigma_n=10.0 / 255.0, seed=1234)
    mu, sg = synthetic.param_table(n, args.iters, seed=77)
    res = {}
    with torch.no_grad():
        t0 = time.time()
        _, ref = O.run_episode(sd, data, mu, sg, args.iters)
        print(f"f32: {time.time() - t0:.1f} s, final PSNR {ref[:, -1].numpy()}", flush=True)
        for name in args.plans.split(","):
            t0 = time.time()
            _, hist = O.run_episode(sd, data, mu, sg, args.iters, bf16_operands=plan_for(name))
            d = (hist - ref).abs().max(dim=0).values.numpy()
            res[name] = d.tolist()
            marks = [0, 5, 9, 19, 29, 39, 49, args.iters - 1]
            print(f"{name:10s} {time.time() - t0:6.1f} s  |dPSNR| at it " +
                  "  ".join(f"{m + 1}:{d[m]:.4f}" for m in marks if m < args.iters) + f"   max {d.max():.4f}", flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump({"size": h, "accel": args.accel, "iters": args.iters, "slices": n, "abs_dpsnr_vs_f32": res}, f, indent=1)


if __name__ == "__main__":
    main()
