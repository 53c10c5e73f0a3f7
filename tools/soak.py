#!/usr/bin/env python3
"""Race screen: the same 40-step episode at the headline size twice (fresh engines), results must be bit-identical
and finite.  Usage on a GPU box: python tools/soak.py [batch size steps] [--bf16]."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dt4image_restoration_amd import synthetic, weights  # noqa: E402
from dt4image_restoration_amd.engine import PnPEngine  # noqa: E402


def run(n, size, steps, bf16=False):
    dev = torch.device("cuda", 0)
    sd = weights.generate_unet_weights(0, "unit_gain")
    data = synthetic.make_problem(n, size, size, seed=1234)
    mu, sg = synthetic.param_table(n, steps, seed=77)
    e = PnPEngine(n, size, size, bf16_convs=bf16)
    e.load_weights(sd)
    x, z, u = e.reset(torch.view_as_complex(torch.from_numpy(data["x0"])).to(dev),
                      torch.view_as_complex(torch.from_numpy(data["y0"])).to(dev), torch.from_numpy(data["mask"]).to(dev))
    for t in range(steps):
        e.step(x, z, u, torch.from_numpy(mu[:, t].copy()).to(dev), torch.from_numpy(sg[:, t].copy()).to(dev))
    torch.cuda.synchronize()
    return x.clone(), z.clone(), u.clone()


if __name__ == "__main__":
    bf16 = "--bf16" in sys.argv
    argv = [v for v in sys.argv[1:] if v != "--bf16"]
    n, size, steps = (int(v) for v in (argv[:3] + ["64", "256", "40"][len(argv):]))
    a = run(n, size, steps, bf16)
    b = run(n, size, steps, bf16)
    ok = all(torch.equal(p, q) for p, q in zip(a, b)) and all(bool(torch.isfinite(torch.view_as_real(p) if p.is_complex() else p).all()) for p in a)
    print("soak", n, size, steps, "bf16-operand convs" if bf16 else "f32", "bit-identical and finite:", ok)
    sys.exit(0 if ok else 1)
