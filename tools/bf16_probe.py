"""Exploration: bf16-operand conv mode vs the oracle's bf16 mode and vs f32 (per-stage errors, PSNR offsets)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dt4image_restoration_amd import synthetic, weights
from dt4image_restoration_amd.engine import PnPEngine
from oracle import pnp_oracle as O

sd_np = weights.generate_unet_weights(0, "unit_gain")
sd = O.torch_weights(sd_np)
for (n, h, w) in [(1, 32, 32), (2, 48, 64), (1, 128, 128), (2, 256, 256)]:
    e = PnPEngine(n, h, w, keep_stages=True, bf16_convs=True); e.load_weights(sd_np)
    x = (torch.from_numpy(synthetic.hash_uniform(5, h * 100 + w, n * h * w).reshape(n, 1, h, w)) + 1) * 0.5
    sigma = torch.linspace(5, 50, n) / 255.0
    got = e.denoise(x.cuda(), sigma.cuda()).cpu()
    nm = torch.ones(n, 1, h, w) * sigma.view(n, 1, 1, 1)
    rb, sb = O.unet_forward(sd, torch.cat([x, nm], 1), return_stages=True, bf16_operands=True)
    rf, sf = O.unet_forward(sd, torch.cat([x, nm], 1), return_stages=True)
    print(f"-- {n}x{h}x{w}")
    for which, name in enumerate(sb):
        a = e.read_stage(which).cpu()
        sc = float(sb[name].abs().max())
        print(f"  {name}: vs bf16 oracle max {float((a - sb[name]).abs().max()) / sc:.2e} mean {float((a - sb[name]).abs().mean()) / sc:.2e}"
              f" | bf16 oracle vs f32 oracle max {float((sb[name] - sf[name]).abs().max()) / sc:.2e} mean {float((sb[name] - sf[name]).abs().mean()) / sc:.2e}")
    print(f"  out: vs bf16 oracle {float((got - rb.clamp(0, 1)).abs().max()):.2e}; bf16 vs f32 oracle {float((rb - rf).abs().max()):.2e}")

# trajectory: 128x128, 10 iterations (configs[0] parameters) and PSNR offsets
data = synthetic.make_problem(2, 128, 128, accel=4.0, seed=1234)
mu_tab, sg_tab = synthetic.param_table(2, 10, seed=77)
def run_engine(bf16):
    e = PnPEngine(2, 128, 128, bf16_convs=bf16); e.load_weights(sd_np)
    x, z, u = e.reset(torch.view_as_complex(torch.from_numpy(data["x0"])).cuda(), torch.view_as_complex(torch.from_numpy(data["y0"])).cuda(), torch.from_numpy(data["mask"]).cuda())
    ps = []
    for t in range(10):
        e.step(x, z, u, torch.from_numpy(mu_tab[:, t].copy()).cuda(), torch.from_numpy(sg_tab[:, t].copy()).cuda())
        ps.append(e.psnr(x, torch.from_numpy(data["gt"]).cuda()).cpu().numpy())
    return np.array(ps)
def run_oracle(bf16):
    st = O.reset(data); ps = []
    for t in range(10):
        st, _ = O.admm_step(sd, st, torch.from_numpy(mu_tab[:, t].copy()), torch.from_numpy(sg_tab[:, t].copy()), bf16_operands=bf16)
        ps.append(O.psnr(st["x"], st["gt"]).numpy().reshape(-1))
    return np.array(ps)
gb, gf, ob, of = run_engine(True), run_engine(False), run_oracle(True), run_oracle(False)
print("PSNR per iteration slice 0: f32 oracle", np.round(of[:, 0], 4))
print("  |gpu bf16 - oracle bf16| max dB", np.abs(gb - ob).max(), " |gpu f32 - oracle f32|", np.abs(gf - of).max())
print("  |oracle bf16 - oracle f32| per iter", np.round(np.abs(ob - of).max(axis=1), 4))
print("  |gpu bf16 - oracle f32| per iter", np.round(np.abs(gb - of).max(axis=1), 4))
