#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference; it does not exist on the GPU
box and nothing at test time imports it):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py

Route: `evaluation.env` imports torchvision and skimage at module scope
for code that is off the hot path (env.py:4,7 -> used only at :48 and :142); neither is
installed, so empty placeholder modules are registered for those two names before the
import.  `PnPEnv.__init__` (env.py:31-40) does a `torch.hub.load` network fetch; it is
never executed - the env object is made with `PnPEnv.__new__` and `.denoiser` assigned.
The denoiser weights come from this repo's deterministic generator, saved to a temp
file and loaded by the reference's own `UNetDenoiser2D(ckpt_path=...)`.

Everything written is DATA (inputs + the reference's outputs); no reference source text
is copied.
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
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)

from dt4image_restoration_amd import synthetic, weights  # noqa: E402


def import_reference():
    for name in ("torchvision", "torchvision.transforms", "skimage", "skimage.metrics"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["skimage"].metrics = sys.modules["skimage.metrics"]
    sys.modules["skimage.metrics"].peak_signal_noise_ratio = None
    sys.path.insert(0, REF)
    from evaluation.env import PnPEnv, torch_psnr
    from evaluation.noise import UNet, UNetDenoiser2D
    from evaluation.utils.transformations import fft, ifft
    return PnPEnv, torch_psnr, UNet, UNetDenoiser2D, fft, ifft


def ref_denoiser(UNetDenoiser2D, sd_np, dtype=torch.float32):
    with tempfile.NamedTemporaryFile(suffix=".pt", delete=False) as f:
        path = f.name
    torch.save(weights.to_torch_state_dict(sd_np), path)
    try:
        den = UNetDenoiser2D(ckpt_path=path)
    finally:
        os.unlink(path)
    return den.to(dtype)


def ref_state(data, i, dtype=torch.float32):
    """State dict for slice i exactly as PnPEnv.reset builds it (env.py:57-71), for any H,W
    (reset itself hard-codes 128 in its mask reshape)."""
    x = torch.view_as_complex(torch.from_numpy(data["x0"][i:i + 1]).to(dtype).contiguous())
    y0 = torch.view_as_complex(torch.from_numpy(data["y0"][i:i + 1]).to(dtype).contiguous())
    h, w = x.shape[-2:]
    mask = torch.from_numpy(data["mask"]).reshape(1, 1, h, w).contiguous().to(torch.bool)
    return OrderedDict({"x": x, "y0": y0, "z": x.clone().detach(), "u": torch.zeros_like(x), "mask": mask,
                        "gt": torch.from_numpy(data["gt"][i:i + 1]).to(dtype), "T": 0})


def gen_g8(PnPEnv, torch_psnr, UNetDenoiser2D):
    """G8: BASELINE configs[4] at its stated length - 512x512, 8x radial mask, 50 iterations (+ the bench's 3 warm-up steps
    = 53), the parameter table `bench.py --size 512 --accel 8 --steps 50 --warmup 3` steps with (rows 0, 1 of
    synthetic.param_table(n, 53, seed=77) for any n), 2 independent single-slice runs of the reference's own PnPEnv.step
    (env.py:74-100) in f32: per-iteration PSNR [2,53] and the final image of slice 0."""
    n8, it8 = 2, 53
    sd0 = weights.generate_unet_weights(0, "unit_gain")
    data8 = synthetic.make_problem(n8, 512, 512, accel=8.0, sigma_n=10.0 / 255.0, seed=1234)
    mu_tab, sig_tab = synthetic.param_table(n8, it8, seed=77)
    env = PnPEnv.__new__(PnPEnv)
    env.denoiser = ref_denoiser(UNetDenoiser2D, sd0)
    ps = np.zeros((n8, it8))
    xfin = None
    for i in range(n8):
        st = ref_state(data8, i)
        with torch.no_grad():
            for t in range(it8):
                act = OrderedDict(T=torch.tensor(0.0), mu=torch.tensor(float(mu_tab[i, t])),
                                  sigma_d=torch.tensor([float(sig_tab[i, t])]))
                st, _ = env.step(st, act)
                ps[i, t] = float(torch_psnr(st["x"].reshape(1, 512, 512), st["gt"].reshape(1, 512, 512)))
        if i == 0:
            xfin = st["x"].numpy()[0, 0].astype(np.float32)
        print("G8 slice", i, ps[i, ::13], flush=True)
    np.savez_compressed(os.path.join(HERE, "g8_config4.npz"), psnr=ps, x_final_slice0=xfin, mu_tab=mu_tab, sig_tab=sig_tab,
                        size=np.array(512), accel=np.array(8.0))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    PnPEnv, torch_psnr, UNet, UNetDenoiser2D, fft, ifft = import_reference()
    if "--only-g8" in sys.argv:
        gen_g8(PnPEnv, torch_psnr, UNetDenoiser2D)
        return
    out = {}

    # ---- G1: centred FFT pair ------------------------------------------------------
    g1 = {}
    for n in (8, 16, 128):
        v = synthetic.hash_uniform(11, n, 2 * n * n).reshape(1, 1, n, n, 2)
        c = torch.view_as_complex(torch.from_numpy(v.copy()))
        g1[f"in_{n}"] = v
        g1[f"fft_{n}"] = torch.view_as_real(fft(c)).numpy()
        g1[f"ifft_{n}"] = torch.view_as_real(ifft(c)).numpy()
    # non-square, to pin the axis convention
    v = synthetic.hash_uniform(11, 999, 2 * 16 * 32).reshape(1, 1, 16, 32, 2)
    c = torch.view_as_complex(torch.from_numpy(v.copy()))
    g1["in_16x32"] = v
    g1["fft_16x32"] = torch.view_as_real(fft(c)).numpy()
    g1["ifft_16x32"] = torch.view_as_real(ifft(c)).numpy()
    np.savez_compressed(os.path.join(HERE, "g1_fft.npz"), **g1)

    # ---- G2: U-Net forward with generator weights -----------------------------------
    g2 = {}
    for tag, (seed, init) in {"unit": (0, "unit_gain"), "tdef": (1, "torch_default")}.items():
        sd = weights.generate_unet_weights(seed, init)
        net = UNet(2, 1)
        net.load_state_dict(weights.to_torch_state_dict(sd))
        net.eval()
        for shape in ((1, 2, 32, 32), (2, 2, 48, 64)):
            x = (synthetic.hash_uniform(21, shape[2] * 1000 + shape[3], int(np.prod(shape))).reshape(shape) + 1) * 0.5
            x[:, 1] = x[:, 1, :1, :1] * 0.2          # channel 1 is a constant plane like the sigma map
            key = f"{tag}_{shape[0]}x{shape[2]}x{shape[3]}"
            g2[f"in_{key}"] = x.astype(np.float32)
            with torch.no_grad():
                g2[f"out_{key}"] = net(torch.from_numpy(x.astype(np.float32))).numpy()
        # stage activations at 128x128 as checksums (sum, L2) - small but sensitive
        x = (synthetic.hash_uniform(22, 128, 2 * 128 * 128).reshape(1, 2, 128, 128) + 1) * 0.5
        x[:, 1] = 15.0 / 255.0
        acts = {}
        hooks = []
        for name in ("inc", "down1", "down2", "down3", "down4", "up1", "up2", "up3", "up4", "outc"):
            hooks.append(getattr(net, name).register_forward_hook(
                lambda m, i, o, name=name: acts.__setitem__(name, o.detach())))
        with torch.no_grad():
            y = net(torch.from_numpy(x.astype(np.float32)))
        for h in hooks:
            h.remove()
        g2[f"in_{tag}_128"] = x.astype(np.float32)
        g2[f"out_{tag}_128"] = y.numpy()
        g2[f"stage_{tag}_128"] = np.array([[float(a.double().sum()), float(a.double().pow(2).sum().sqrt())]
                                           for a in acts.values()])
    np.savez_compressed(os.path.join(HERE, "g2_unet.npz"), **g2)

    # ---- G3: config-1 trajectory (128x128, 10 iters, mu=0.1, sigma_d=15/255), f32 and the f64 sensitivity record -----
    sd0 = weights.generate_unet_weights(0, "unit_gain")
    data = synthetic.make_problem(1, 128, 128, accel=4.0, sigma_n=10.0 / 255.0, seed=1234)
    g3 = {}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        env = PnPEnv.__new__(PnPEnv)
        env.denoiser = ref_denoiser(UNetDenoiser2D, sd0, dt)
        mat = {k: torch.from_numpy(np.asarray(v)).to(dt) if v.dtype != bool else torch.from_numpy(v)
               for k, v in data.items()}
        st = env.reset(mat, "cpu")                        # the reference's own reset (128x128 only)
        psnrs = [float(torch_psnr(st["x"].real.squeeze(0), st["gt"].reshape(1, 128, 128)))]
        act = OrderedDict(T=torch.tensor(0.0, dtype=dt), mu=torch.tensor(0.1, dtype=dt),
                          sigma_d=torch.tensor([15.0 / 255.0], dtype=dt))
        with torch.no_grad():
            for _ in range(10):
                st, done = env.step(st, act)
                assert not done
                psnrs.append(float(env.compute_reward(st["x"].reshape(1, 128, 128), st["gt"])))
        g3[f"psnr_{tag}"] = np.array(psnrs)
        g3[f"x_{tag}"] = st["x"].numpy()
        g3[f"z_{tag}"] = torch.view_as_real(st["z"]).numpy()
        g3[f"u_{tag}"] = torch.view_as_real(st["u"]).numpy()
        g3[f"T_{tag}"] = np.array(st["T"])
    np.savez_compressed(os.path.join(HERE, "g3_config1.npz"), **g3)

    # ---- G4: 256x256, 4 independent single-slice runs, 30 iters, per-slice parameter tables
    n4, it4 = 4, 30
    data4 = synthetic.make_problem(n4, 256, 256, accel=4.0, sigma_n=10.0 / 255.0, seed=1234)
    mu_tab, sig_tab = synthetic.param_table(n4, it4, seed=77)
    env = PnPEnv.__new__(PnPEnv)
    env.denoiser = ref_denoiser(UNetDenoiser2D, sd0)
    ps = np.zeros((n4, it4))
    xfin = []
    for i in range(n4):
        st = ref_state(data4, i)
        with torch.no_grad():
            for t in range(it4):
                act = OrderedDict(T=torch.tensor(0.0), mu=torch.tensor(float(mu_tab[i, t])),
                                  sigma_d=torch.tensor([float(sig_tab[i, t])]))
                st, _ = env.step(st, act)
                ps[i, t] = float(torch_psnr(st["x"].reshape(1, 256, 256), st["gt"].reshape(1, 256, 256)))
        xfin.append(st["x"].numpy()[0, 0])
        print("G4 slice", i, ps[i, ::6])
    np.savez_compressed(os.path.join(HERE, "g4_256.npz"), psnr=ps, x_final=np.stack(xfin).astype(np.float32),
                        mu_tab=mu_tab, sig_tab=sig_tab)

    # ---- G5: torch_psnr known answers incl. clamp ------------------------------------
    a = (synthetic.hash_uniform(5, 1, 3 * 64).reshape(3, 8, 8) * 1.5).astype(np.float32)   # outside [0,1] too
    b = ((synthetic.hash_uniform(5, 2, 3 * 64).reshape(3, 8, 8) + 1) * 0.5).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "g5_psnr.npz"), a=a, b=b,
                        psnr=torch_psnr(torch.from_numpy(a), torch.from_numpy(b)).numpy())

    # ---- G6: early stop: action T > 0.5 leaves the state untouched --------------------
    data6 = synthetic.make_problem(1, 128, 128, accel=4.0, seed=4321)
    mat = {k: torch.from_numpy(np.asarray(v)) for k, v in data6.items()}
    st = env.reset(mat, "cpu")
    g6 = {}
    with torch.no_grad():
        for t, Tact in enumerate((0.1, 0.3, 0.7, 0.2)):
            act = OrderedDict(T=torch.tensor(Tact), mu=torch.tensor(0.2), sigma_d=torch.tensor([20.0 / 255.0]))
            st, done = env.step(st, act)
            g6[f"done_{t}"] = np.array(bool(done))
            g6[f"x_{t}"] = np.array(st["x"].real.numpy())
            g6[f"T_{t}"] = np.array(float(st["T"]))
    np.savez_compressed(os.path.join(HERE, "g6_earlystop.npz"), **g6)

    # ---- G8: configs[4] at its stated length (512x512, 8x mask, 53 iterations) ---------------
    gen_g8(PnPEnv, torch_psnr, UNetDenoiser2D)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
