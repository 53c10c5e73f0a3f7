#!/usr/bin/env python3
"""Quick parity probe for kernel experiments (PNP_LIB_PATH selects the build): 4 x 256x256, all-Winograd plan (PNP_WINO_MIN_BLOCKS=1),
3 steps against the CPU oracle - prints the PSNR deltas and max |dx|."""
import os, sys
os.environ.setdefault("PNP_WINO_MIN_BLOCKS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dt4image_restoration_amd import synthetic, weights
from dt4image_restoration_amd.engine import PnPEngine
from oracle import pnp_oracle as O
n, hw, it = 4, 256, 3
sd = weights.generate_unet_weights(0, "unit_gain")
data = synthetic.make_problem(n, hw, hw, accel=4.0, seed=1234)
mu, sg = synthetic.param_table(n, it, seed=77)
e = PnPEngine(n, hw, hw); e.load_weights(sd)
x, z, u = e.reset(torch.view_as_complex(torch.from_numpy(data["x0"])).cuda(), torch.view_as_complex(torch.from_numpy(data["y0"])).cuda(), torch.from_numpy(data["mask"]).cuda())
so = O.reset(data); sdt = O.torch_weights(sd)
for t in range(it):
    e.step(x, z, u, torch.from_numpy(mu[:, t].copy()).cuda(), torch.from_numpy(sg[:, t].copy()).cuda())
    so, _ = O.admm_step(sdt, so, torch.from_numpy(mu[:, t].copy()), torch.from_numpy(sg[:, t].copy()))
p = e.psnr(x, torch.from_numpy(data["gt"]).cuda()).cpu()
print("algos", "".join(map(str, e.conv_algorithms()[1:27])), "dPSNR", float((p - O.psnr(so["x"], so["gt"]).reshape(-1)).abs().max()), "max|dx|", float((x.cpu() - so["x"]).abs().max()))
