"""GPU parity of each HIP stage against the CPU oracle (same seeded inputs), through the C ABI."""
import os

import numpy as np
import pytest
import torch

from dt4image_restoration_amd import synthetic, weights
from oracle import pnp_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sd_np():
    return weights.generate_unet_weights(0, "unit_gain")


def _engine(n, h, w, sd=None, profile=False, keep_stages=False):
    from dt4image_restoration_amd.engine import PnPEngine
    e = PnPEngine(n, h, w, profile=profile, keep_stages=keep_stages)
    if sd is not None:
        e.load_weights(sd)
    return e


@pytest.mark.parametrize("h,w,b", [(16, 16, 2), (16, 32, 1), (64, 64, 3), (128, 128, 1), (256, 256, 2), (512, 512, 1)])
def test_fft2c_matches_oracle(h, w, b):
    e = _engine(b, h, w)
    v = synthetic.hash_uniform(3, h * 1000 + w, 2 * b * h * w).reshape(b, 1, h, w, 2)
    c = torch.view_as_complex(torch.from_numpy(v.copy()))
    for inverse, ref in ((False, O.fft2c(c)), (True, O.ifft2c(c))):
        got = e.fft2c(c.cuda(), inverse=inverse).cpu()
        # FLOAT TOLERANCE: f32 FFT, |values| ~ 1 after ortho scaling; 2e-6 abs is ~8 ulp at the size of the data
        np.testing.assert_allclose(torch.view_as_real(got).numpy(), torch.view_as_real(ref).numpy(), rtol=0, atol=3e-6)


def test_fft2c_golden_and_roundtrip(golden_dir):
    g = np.load(os.path.join(golden_dir, "g1_fft.npz"))
    e = _engine(1, 128, 128)
    c = torch.view_as_complex(torch.from_numpy(g["in_128"].copy())).cuda()
    f = e.fft2c(c)
    np.testing.assert_allclose(torch.view_as_real(f.cpu()).numpy(), g["fft_128"], rtol=0, atol=3e-6)
    np.testing.assert_allclose(torch.view_as_real(e.fft2c(c, inverse=True).cpu()).numpy(), g["ifft_128"], rtol=0, atol=3e-6)
    back = e.fft2c(f, inverse=True)
    np.testing.assert_allclose(torch.view_as_real(back.cpu()).numpy(), g["in_128"], rtol=0, atol=3e-6)
    # Parseval (ortho): energy preserved
    assert abs(float((f.abs() ** 2).sum()) / float((c.abs() ** 2).sum()) - 1) < 1e-5


@pytest.mark.parametrize("n,h,w", [(1, 32, 32), (2, 48, 64), (1, 128, 128), (3, 64, 16), (2, 16, 16)])
def test_denoiser_matches_oracle_per_stage(sd_np, n, h, w):
    e = _engine(n, h, w, sd_np, keep_stages=True)
    sd = O.torch_weights(sd_np)
    x = (torch.from_numpy(synthetic.hash_uniform(5, h * 100 + w, n * h * w).reshape(n, 1, h, w)) + 1) * 0.5
    sigma = torch.linspace(5, 50, n) / 255.0
    got = e.denoise(x.cuda(), sigma.cuda())
    noise_map = torch.ones(n, 1, h, w) * sigma.view(n, 1, 1, 1)
    ref_raw, stages = O.unet_forward(sd, torch.cat([x, noise_map], 1), return_stages=True)
    for which, (name, ref) in enumerate(stages.items()):
        a = e.read_stage(which).cpu()
        assert a.shape == ref.shape, name
        # FLOAT TOLERANCE: f32 conv with K up to 6912 terms, activations O(1): summation-order noise ~1e-6..1e-5
        err = float((a - ref).abs().max())
        assert err < 5e-5 * max(1.0, float(ref.abs().max())), f"stage {name}: max err {err}"
    ref = torch.clamp(ref_raw, 0, 1)
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=0, atol=1e-5)
    # the production handle fuses the last 1x1 layer into the preceding conv: same output, stage 8 not materialised
    from dt4image_restoration_amd._lib import PnPError
    e2 = _engine(n, h, w, sd_np)
    got2 = e2.denoise(x.cuda(), sigma.cuda())
    np.testing.assert_allclose(got2.cpu().numpy(), ref.numpy(), rtol=0, atol=1e-5)
    with pytest.raises(PnPError):
        e2.read_stage(8)


def test_denoiser_golden_from_reference(sd_np, golden_dir):
    g = np.load(os.path.join(golden_dir, "g2_unet.npz"))
    # the fixture's 2nd channel is a constant plane = what the sigma map is; feed it as sigma
    for key, (n, h, w) in {"1x32x32": (1, 32, 32), "2x48x64": (2, 48, 64)}.items():
        xin = g[f"in_unit_{key}"]
        e = _engine(n, h, w, sd_np)
        sigma = torch.from_numpy(xin[:, 1, 0, 0].copy())
        got = e.denoise(torch.from_numpy(xin[:, :1].copy()).cuda(), sigma.cuda()).cpu().numpy()
        want = np.clip(g[f"out_unit_{key}"], 0, 1)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-5)


def test_denoiser_torch_default_weights(golden_dir):
    g = np.load(os.path.join(golden_dir, "g2_unet.npz"))
    e = _engine(1, 128, 128, weights.generate_unet_weights(1, "torch_default"))
    xin = g["in_tdef_128"]
    got = e.denoise(torch.from_numpy(xin[:, :1].copy()).cuda(), torch.tensor([15.0 / 255.0]).cuda()).cpu().numpy()
    np.testing.assert_allclose(got, np.clip(g["out_tdef_128"], 0, 1), rtol=0, atol=1e-5)


@pytest.mark.parametrize("n,h,w", [(1, 128, 128), (3, 64, 64), (2, 256, 256), (2, 512, 512), (1, 512, 256), (1, 256, 512)])
def test_prox_dual_matches_oracle(n, h, w):
    data = synthetic.make_problem(n, h, w, accel=4.0, seed=99)
    st = O.reset(data)
    e = _engine(n, h, w)
    x0 = st["z"].clone()
    x, z, u = e.reset(x0.cuda(), st["y0"].cuda(), st["mask"].reshape(h, w).cuda())
    np.testing.assert_array_equal(x.cpu().numpy(), x0.real.numpy())
    np.testing.assert_array_equal(torch.view_as_real(z.cpu()).numpy(), torch.view_as_real(x0).numpy())
    assert float(u.abs().max()) == 0.0
    # arbitrary x (as if denoised) and a non-zero u
    xd = torch.clamp(x0.real + 0.05 * torch.from_numpy(synthetic.hash_uniform(8, 1, n * h * w).reshape(n, 1, h, w)), 0, 1)
    u0 = 0.1 * torch.view_as_complex(torch.from_numpy(synthetic.hash_uniform(8, 2, 2 * n * h * w).reshape(n, 1, h, w, 2).copy()))
    mu = torch.linspace(0.05, 0.6, n)
    zf = O.fft2c(xd + u0)
    temp = (mu.view(n, 1, 1, 1) * zf + st["y0"]) / (1 + mu.view(n, 1, 1, 1))
    zn = O.ifft2c(torch.where(st["mask"], temp, zf))
    un = u0 + xd - zn
    xg, ug = xd.cuda(), u0.cuda().clone()
    zg = torch.empty_like(ug)
    e.prox_dual(xg, zg, ug, mu.cuda())
    # FLOAT TOLERANCE: two f32 FFTs + pointwise, data O(1)
    np.testing.assert_allclose(torch.view_as_real(zg.cpu()).numpy(), torch.view_as_real(zn).numpy(), rtol=0, atol=5e-6)
    np.testing.assert_allclose(torch.view_as_real(ug.cpu()).numpy(), torch.view_as_real(un).numpy(), rtol=0, atol=5e-6)


@pytest.mark.parametrize("n,h,per_slice_masks", [(8, 256, False), (19, 256, True), (16, 512, False), (9, 512, True)])
def test_prox_dual_per_xcd_persistent_kernel_is_bit_identical_to_three_launches(n, h, per_slice_masks, monkeypatch):
    """Round 5 (PNP_FFT_XCD=1, an experiment that stays off by default: it halves the stage's HBM bytes and is slower): square 256 / 512 slices,
    >= 8 of them, run the data-fidelity stage (env.py:87-93) as ONE persistent launch whose workgroups draw (slice, pass) tasks from per-XCD queues
    so that the complex scratch stays in that XCD's L2 (admm_xcd_kernel).  Same pass bodies as the three-launch path: bit-identical z and u, call
    after call (the counters are never reset), with a third of the slices stopped (they must stay untouched), ragged slice counts per XCD, one mask
    per slice - and against the oracle."""
    from dt4image_restoration_amd.engine import PnPEngine
    w = h
    datas = [synthetic.make_problem(1, h, w, accel=(2.0, 4.0, 8.0)[i % 3], seed=70 + i) for i in range(n if per_slice_masks else 1)]
    if per_slice_masks:
        masks = torch.stack([torch.from_numpy(np.asarray(d["mask"])).reshape(h, w).bool() for d in datas])
        y0 = torch.cat([torch.view_as_complex(torch.from_numpy(d["y0"])) for d in datas])
        x0 = torch.cat([torch.view_as_complex(torch.from_numpy(d["x0"])) for d in datas])
    else:
        d = synthetic.make_problem(n, h, w, accel=4.0, seed=71)
        masks = torch.from_numpy(np.asarray(d["mask"])).reshape(h, w).bool()
        y0 = torch.view_as_complex(torch.from_numpy(d["y0"])); x0 = torch.view_as_complex(torch.from_numpy(d["x0"]))
    xd = torch.clamp(x0.real + 0.05 * torch.from_numpy(synthetic.hash_uniform(11, 1, n * h * w).reshape(n, 1, h, w)), 0, 1)
    u0 = 0.1 * torch.view_as_complex(torch.from_numpy(synthetic.hash_uniform(11, 2, 2 * n * h * w).reshape(n, 1, h, w, 2).copy()))
    mu = torch.linspace(0.05, 0.6, n)
    tact = torch.zeros(n); tact[1::3] = 0.9                 # every third slice has stopped (T > 0.5, env.py:79-81)
    outs = {}
    for name in ("three", "xcd"):
        if name == "xcd":
            monkeypatch.setenv("PNP_FFT_XCD", "1")
        e = PnPEngine(n, h, w)
        monkeypatch.delenv("PNP_FFT_XCD", raising=False)
        e.reset(x0.cuda(), y0.cuda(), masks.cuda())
        res = []
        for rep in range(4):                                # the scratch and the counters are reused call after call
            xg, ug = xd.cuda(), u0.cuda().clone()
            zg = torch.full_like(ug, 7.0)
            e.prox_dual(xg, zg, ug, mu.cuda(), tact.cuda())
            res.append((torch.view_as_real(zg).clone(), torch.view_as_real(ug).clone()))
        for r in res[1:]:
            assert torch.equal(r[0], res[0][0]) and torch.equal(r[1], res[0][1])
        outs[name] = res[0]
    assert torch.equal(outs["xcd"][0], outs["three"][0]) and torch.equal(outs["xcd"][1], outs["three"][1])
    zg, ug = torch.view_as_complex(outs["xcd"][0].cpu()), torch.view_as_complex(outs["xcd"][1].cpu())
    for i in (0, 1, n - 1):
        if tact[i] > 0.5:                                   # stopped: z left as handed in, u unchanged
            assert float((zg[i] - 7.0).abs().max()) == 0.0 and torch.equal(ug[i], u0[i])
            continue
        mk = masks[i] if per_slice_masks else masks
        zf = O.fft2c(xd[i:i + 1] + u0[i:i + 1])
        zn = O.ifft2c(torch.where(mk.reshape(1, 1, h, w), (mu[i] * zf + y0[i:i + 1]) / (1 + mu[i]), zf))
        # FLOAT TOLERANCE: two f32 FFTs + pointwise, data O(1)
        np.testing.assert_allclose(torch.view_as_real(zg[i:i + 1]).numpy(), torch.view_as_real(zn).numpy(), rtol=0, atol=5e-6)
        np.testing.assert_allclose(torch.view_as_real(ug[i:i + 1]).numpy(), torch.view_as_real(u0[i:i + 1] + xd[i:i + 1] - zn).numpy(), rtol=0, atol=5e-6)


@pytest.mark.parametrize("h,w", [(64, 64), (128, 256), (256, 64)])
def test_prox_dual_per_slice_masks_and_non_square(h, w):
    """mask_n == N (the reference's z[mask] needs a mask of z's shape for N > 1): every slice its own sampling
    pattern, on square and non-square power-of-two sizes; slice n of the batch == the oracle's single-slice solve."""
    n = 3
    datas = [synthetic.make_problem(1, h, w, accel=acc, seed=50 + i) for i, acc in enumerate((2.0, 4.0, 8.0))]
    masks = torch.stack([torch.from_numpy(np.asarray(d["mask"])).reshape(h, w).bool() for d in datas])
    assert not torch.equal(masks[0], masks[2])
    y0 = torch.cat([torch.view_as_complex(torch.from_numpy(d["y0"])) for d in datas])
    x0 = torch.cat([torch.view_as_complex(torch.from_numpy(d["x0"])) for d in datas])
    e = _engine(n, h, w)
    e.reset(x0.cuda(), y0.cuda(), masks.cuda())
    xd = torch.clamp(x0.real + 0.05 * torch.from_numpy(synthetic.hash_uniform(9, 1, n * h * w).reshape(n, 1, h, w)), 0, 1)
    u0 = 0.1 * torch.view_as_complex(torch.from_numpy(synthetic.hash_uniform(9, 2, 2 * n * h * w).reshape(n, 1, h, w, 2).copy()))
    mu = torch.tensor([0.07, 0.3, 0.55])
    xg, ug = xd.cuda(), u0.cuda().clone()
    zg = torch.empty_like(ug)
    e.prox_dual(xg, zg, ug, mu.cuda())
    for i in range(n):
        zf = O.fft2c(xd[i:i + 1] + u0[i:i + 1])
        temp = (mu[i] * zf + y0[i:i + 1]) / (1 + mu[i])
        zn = O.ifft2c(torch.where(masks[i].reshape(1, 1, h, w), temp, zf))
        # FLOAT TOLERANCE: two f32 FFTs + pointwise, data O(1)
        np.testing.assert_allclose(torch.view_as_real(zg[i:i + 1].cpu()).numpy(), torch.view_as_real(zn).numpy(), rtol=0, atol=5e-6)
        np.testing.assert_allclose(torch.view_as_real(ug[i:i + 1].cpu()).numpy(),
                                   torch.view_as_real(u0[i:i + 1] + xd[i:i + 1] - zn).numpy(), rtol=0, atol=5e-6)


def test_psnr_matches_oracle(golden_dir):
    n, h, w = 3, 64, 64
    e = _engine(n, h, w)
    a = torch.from_numpy((synthetic.hash_uniform(6, 1, n * h * w).reshape(n, 1, h, w) * 1.3).astype(np.float32))
    b = (torch.from_numpy(synthetic.hash_uniform(6, 2, n * h * w).reshape(n, 1, h, w)) + 1) * 0.5
    got = e.psnr(a.cuda(), b.cuda()).cpu()
    np.testing.assert_allclose(got.numpy(), O.psnr(a, b)[:, 0].numpy(), rtol=1e-6)


@pytest.mark.parametrize("n,h,w", [(64, 128, 128), (1, 32, 32), (2, 48, 64), (3, 64, 16)])
def test_winograd_path_matches_oracle_per_stage(sd_np, n, h, w, monkeypatch):
    """The Winograd kernel on every layer it can take (PNP_WINO_MIN_BLOCKS=1 lifts the workgroup-count gate so that small
    and odd-sized problems - 3x4-pixel bottom level, 8/16/32-wide tile variants - run it too), per stage against the oracle."""
    if n < 64:
        monkeypatch.setenv("PNP_WINO_MIN_BLOCKS", "1")
    monkeypatch.setenv("PNP_NO_WINO_F4", "1")              # this test is about the F(2x2) kernel; F(4x4) has its own below
    e = _engine(n, h, w, sd_np, keep_stages=True)
    algos = e.conv_algorithms()
    assert sum(1 for v in algos if v == 1) >= (20 if n == 64 else 26) and 4 not in algos
    sd = O.torch_weights(sd_np)
    x = (torch.from_numpy(synthetic.hash_uniform(9, h * 100 + w, n * h * w).reshape(n, 1, h, w)) + 1) * 0.5
    sigma = torch.linspace(5, 50, n) / 255.0
    got = e.denoise(x.cuda(), sigma.cuda())
    noise_map = torch.ones(n, 1, h, w) * sigma.view(n, 1, 1, 1)
    ref_raw, stages = O.unet_forward(sd, torch.cat([x, noise_map], 1), return_stages=True)
    for which, (name, ref) in enumerate(stages.items()):
        a = e.read_stage(which).cpu()
        err = float((a - ref).abs().max())
        # FLOAT TOLERANCE: Winograd F(2x2,3x3) in f32: same order of error as the direct sum (K up to 6912 terms)
        assert err < 5e-5 * max(1.0, float(ref.abs().max())), f"stage {name}: max err {err}"
    np.testing.assert_allclose(got.cpu().numpy(), torch.clamp(ref_raw, 0, 1).numpy(), rtol=0, atol=1e-5)


@pytest.mark.parametrize("n,h,w,min_cin", [(2, 128, 128, 128), (2, 96, 112, 128), (1, 144, 64, 128), (1, 256, 256, 128), (3, 128, 64, 64),
                                           (3, 256, 256, 64),       # 16 x 16 bottom level: two slices stacked per workgroup (odd batch)
                                           (2, 128, 128, 32), (1, 80, 48, 32),    # + the Cout = 32 / Cin = 32 layers (4-wave variant)
                                           (8, 272, 272, 32), (2, 256, 144, 32)])  # level widths 34 / 18, 9: not multiples of 4 - F(2x2) / direct there
@pytest.mark.parametrize("schedule", ["default", "in-step", "independent"])
def test_winograd_f4_path_matches_oracle_per_stage(sd_np, n, h, w, min_cin, schedule, monkeypatch):
    """F(4x4,3x3) (winograd4_kernels.hip, points 0, +-3/4, +-3/2, inf) on every layer it can take - workgroup gate lifted so
    small and ragged sizes run it (partial 4x4 tiles, 16- and 32-wide tile variants, upsample+concat sources, pooled copies
    written by its epilogue) - per stage against the oracle.  min_cin = 64 also sends the 64-channel layers through it.
    schedule: the per-layer default mix; every 64-channel-block layer on the 8-wave in-step kernel (incl. the stacked 16 x 16
    level); every one on 16-tile M-blocks (two independent workgroups per CU)."""
    monkeypatch.setenv("PNP_WINO_MIN_BLOCKS", "1")
    monkeypatch.setenv("PNP_WINO_F4_MIN_CIN", str(min_cin))
    if schedule == "in-step":
        monkeypatch.setenv("PNP_NO_WINO_F4_PHASED", "1")
        monkeypatch.setenv("PNP_WINO_F4_MT16", "1")
    elif schedule == "independent":
        monkeypatch.setenv("PNP_WINO_F4_MT16", "3")
    e = _engine(n, h, w, sd_np, keep_stages=True)
    algos = e.conv_algorithms()
    assert sum(1 for v in algos if v == 4) >= (9 if min_cin <= 64 and min(h, w) >= 128 else 1), algos
    # the F(4x4) epilogue stores 4-wide tiles whole in x: a level whose width is no multiple of 4 must not be planned on it
    from dt4image_restoration_amd import unet_spec
    assert all(algos[l.index] != 4 for l in unet_spec.UNET_LAYERS[1:27] if (w >> l.level) % 4 != 0), algos
    sd = O.torch_weights(sd_np)
    x = (torch.from_numpy(synthetic.hash_uniform(19, h * 100 + w, n * h * w).reshape(n, 1, h, w)) + 1) * 0.5
    sigma = torch.linspace(5, 50, n) / 255.0
    got = e.denoise(x.cuda(), sigma.cuda())
    noise_map = torch.ones(n, 1, h, w) * sigma.view(n, 1, 1, 1)
    ref_raw, stages = O.unet_forward(sd, torch.cat([x, noise_map], 1), return_stages=True)
    for which, (name, ref) in enumerate(stages.items()):
        a = e.read_stage(which).cpu()
        err = float((a - ref).abs().max())
        # FLOAT TOLERANCE: F(4x4,3x3) in f32 with the 3/4, 3/2 points: ~2e-6 of the output scale per layer (3x F(2x2), 6x the
        # direct sum; winograd4_kernels.hip header), accumulated over the up to 9 F(4x4) layers in front of a stage
        assert err < 5e-5 * max(1.0, float(ref.abs().max())), f"stage {name}: max err {err}"
    np.testing.assert_allclose(got.cpu().numpy(), torch.clamp(ref_raw, 0, 1).numpy(), rtol=0, atol=1e-5)


# ---- bf16-operand conv mode (PNP_FLAG_BF16_CONVS, BASELINE configs[4]) ------------------------------------------------
@pytest.mark.parametrize("n,h,w", [(1, 32, 32), (2, 48, 64), (1, 128, 128), (2, 256, 256), (3, 64, 16)])
def test_bf16_convs_match_bf16_oracle_per_stage(sd_np, n, h, w):
    """Parity target = the oracle with the same operand rounding (conv inputs and weights to bf16, nearest even; f32
    products and sums).  The first stage's first bf16 layer has a single chunk and differs by summation order only; deeper
    stages also see rare bf16 rounding flips of activations that sit within one f32 ulp of a rounding boundary."""
    from dt4image_restoration_amd.engine import PnPEngine
    e = PnPEngine(n, h, w, keep_stages=True, bf16_convs=True)
    e.load_weights(sd_np)
    sd = O.torch_weights(sd_np)
    x = (torch.from_numpy(synthetic.hash_uniform(5, h * 100 + w, n * h * w).reshape(n, 1, h, w)) + 1) * 0.5
    sigma = torch.linspace(5, 50, n) / 255.0
    got = e.denoise(x.cuda(), sigma.cuda()).cpu()
    nm = torch.ones(n, 1, h, w) * sigma.view(n, 1, 1, 1)
    ref_raw, stages = O.unet_forward(sd, torch.cat([x, nm], 1), return_stages=True, bf16_operands=True)
    f32_raw = O.unet_forward(sd, torch.cat([x, nm], 1))
    for which, (name, ref) in enumerate(stages.items()):
        a = e.read_stage(which).cpu()
        sc = float(ref.abs().max())
        # FLOAT TOLERANCE (bf16 operands, 2^-9 relative rounding): a flipped activation moves a downstream sum by ~1e-3
        assert float((a - ref).abs().max()) < 3e-2 * sc, name
        assert float((a - ref).abs().mean()) < 2e-3 * sc, name
    assert float((got - ref_raw.clamp(0, 1)).abs().max()) < 2e-3
    # and the mode is really on: it differs from the f32 arithmetic by more than the f32 path's own 1e-5
    assert float((got - f32_raw.clamp(0, 1)).abs().max()) > 2e-5
    assert all(v in (0, 5) for v in e.conv_algorithms()[1:27])       # every 3x3 layer on a direct bf16 kernel (5: producer/consumer form)


def test_bf16_convs_trajectory_psnr_offsets(sd_np):
    """128x128, 10 iterations (configs[0] parameters): PSNR within 0.01 dB of the bf16-operand oracle; the offset to the f32
    reference arithmetic stays under north_star's 0.01 dB as well (measured 0.006)."""
    from dt4image_restoration_amd.engine import PnPEngine
    data = synthetic.make_problem(2, 128, 128, accel=4.0, seed=1234)
    mu_tab, sg_tab = synthetic.param_table(2, 10, seed=77)
    sd = O.torch_weights(sd_np)
    e = PnPEngine(2, 128, 128, bf16_convs=True)
    e.load_weights(sd_np)
    gt = torch.from_numpy(data["gt"]).cuda()
    x, z, u = e.reset(torch.view_as_complex(torch.from_numpy(data["x0"])).cuda(),
                      torch.view_as_complex(torch.from_numpy(data["y0"])).cuda(), torch.from_numpy(data["mask"]).cuda())
    sb, sf = O.reset(data), O.reset(data)
    for t in range(10):
        mu, sg = torch.from_numpy(mu_tab[:, t].copy()), torch.from_numpy(sg_tab[:, t].copy())
        e.step(x, z, u, mu.cuda(), sg.cuda())
        sb, _ = O.admm_step(sd, sb, mu, sg, bf16_operands=True)
        sf, _ = O.admm_step(sd, sf, mu, sg)
        p = e.psnr(x, gt).cpu()
        assert float((p - O.psnr(sb["x"], sb["gt"]).reshape(-1)).abs().max()) < 0.01
        assert float((p - O.psnr(sf["x"], sf["gt"]).reshape(-1)).abs().max()) < 0.01


@pytest.mark.parametrize("n,h,w", [(2, 128, 128), (3, 48, 80), (1, 256, 256), (16, 256, 256), (64, 256, 256)])
def test_bf16_activation_storage_is_bit_neutral(sd_np, n, h, w, monkeypatch):
    """bf16 mode keeps activations in HBM as bf16 wherever producer and consumers can (ConvArgs.act16: level 0 always; at
    16 x 256x256 also levels 1-2, whose layers run the producer/consumer kernel, with f32 hand-overs to levels 3-4; at
    64 x 256x256 every tensor that does not feed an upsample): the producer rounds once with the rounding the consumer's
    staging would apply, so the denoiser output must equal - bit for bit - the output of a handle that keeps those
    tensors in f32 (PNP_BF16_F32_ACTS, the layout of rounds 1-2).  Stage 0 of such a handle is not readable as f32 and says so."""
    from dt4image_restoration_amd.engine import PnPEngine
    x = ((torch.from_numpy(synthetic.hash_uniform(5, h * 1000 + w, n * h * w).reshape(n, 1, h, w)) + 1) * 0.5).cuda()
    sigma = (torch.linspace(5, 50, n) / 255.0).cuda()
    e16 = PnPEngine(n, h, w, bf16_convs=True)
    e16.load_weights(sd_np)
    monkeypatch.setenv("PNP_BF16_F32_ACTS", "1")
    e32 = PnPEngine(n, h, w, bf16_convs=True)
    e32.load_weights(sd_np)
    monkeypatch.delenv("PNP_BF16_F32_ACTS")
    a, b = e16.denoise(x, sigma), e32.denoise(x, sigma)
    assert torch.equal(a, b)
    assert float((a - x).abs().max()) > 1e-3                       # the network did something
    e32.read_stage(0)
    with pytest.raises(RuntimeError, match="bf16"):
        e16.read_stage(0)


# ---- shape sweep: every tile-width variant, ragged tile grids, every batch remainder ---------------------------------------
_SWEEP = [(1, 16, 48), (5, 32, 16), (2, 80, 48), (3, 96, 112), (1, 144, 64), (4, 16, 16), (2, 64, 176), (1, 208, 32)]


@pytest.mark.parametrize("mode", ["default", "winograd", "direct", "bf16"])
@pytest.mark.parametrize("n,h,w", _SWEEP)
def test_denoiser_shape_sweep(sd_np, n, h, w, mode, monkeypatch):
    """The U-Net needs H, W multiples of 16 only (noise.py:49-53 pad is then a no-op); sizes that are NOT multiples of the
    kernels' 32-/16-/8-wide tiles exercise partial tiles, odd tile grids and the low-res upsample staging at image borders,
    on each kernel family: plan defaults, Winograd forced on every eligible layer, direct f32 only, bf16 operands."""
    from dt4image_restoration_amd.engine import PnPEngine
    if mode == "winograd":
        monkeypatch.setenv("PNP_WINO_MIN_BLOCKS", "1")
    if mode == "direct":
        monkeypatch.setenv("PNP_NO_WINOGRAD", "1")
    e = PnPEngine(n, h, w, bf16_convs=(mode == "bf16"))
    e.load_weights(sd_np)
    algos = e.conv_algorithms()[1:27]
    if mode == "winograd":
        assert any(v in (1, 4) for v in algos)
    if mode in ("direct", "bf16"):
        assert all(v == 0 for v in algos)
    sd = O.torch_weights(sd_np)
    x = (torch.from_numpy(synthetic.hash_uniform(21, h * 1000 + w, n * h * w).reshape(n, 1, h, w)) + 1) * 0.5
    sigma = torch.linspace(3, 60, n) / 255.0
    got = e.denoise(x.cuda(), sigma.cuda()).cpu()
    ref = O.denoise(sd, x, sigma, bf16_operands=(mode == "bf16"))
    # FLOAT TOLERANCE: f32 summation order (1e-5); bf16 operands: rounding flips reach the output at ~1e-3 (see above)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=0, atol=2e-3 if mode == "bf16" else 1e-5)


def test_slice128_single_workgroup_stage_matches_three_launch_path_and_oracle(monkeypatch):
    """128 x 128 (the reference's size): the whole data-fidelity stage in one workgroup per slice (admm_slice128_kernel:
    one read, one write per slice; default from 192 slices on, forced here with PNP_SLICE128_MIN_N) against the three-launch
    path and the oracle - per-slice mu, per-slice masks, a stopped slice left untouched."""
    from dt4image_restoration_amd.engine import PnPEngine
    n, h, w = 5, 128, 128
    data = synthetic.make_problem(n, h, w, accel=4.0, seed=314)
    rng = np.random.default_rng(3)
    masks = torch.from_numpy(rng.random((n, h, w)) < 0.3)
    x0 = torch.view_as_complex(torch.from_numpy(data["x0"]))
    y0 = torch.view_as_complex(torch.from_numpy(data["y0"])) * masks.reshape(n, 1, h, w)
    xr = torch.from_numpy((synthetic.hash_uniform(8, 1, n * h * w).reshape(n, 1, h, w) + 1) * 0.5)
    u0 = torch.view_as_complex(torch.from_numpy(synthetic.hash_uniform(8, 2, n * h * w * 2).reshape(n, 1, h, w, 2) * 0.1))
    mu = torch.tensor([0.05, 0.2, 0.4, 0.6, 0.9])
    tact = torch.tensor([0.0, 0.0, 0.8, 0.0, 0.0])

    def run(disable):
        monkeypatch.setenv("PNP_SLICE128_MIN_N", "100000" if disable else "1")
        e = PnPEngine(n, h, w, denoiser=False)
        _, z, u = e.reset(x0.cuda(), y0.cuda(), masks.cuda())
        u.copy_(u0.cuda())
        z0 = z.clone()
        e.prox_dual(xr.cuda(), z, u, mu.cuda(), t_action=tact.cuda())
        return z.cpu(), u.cpu(), z0.cpu()

    z1, u1, zin = run(False)
    z3, u3, _ = run(True)
    # FLOAT TOLERANCE: same f32 butterflies in another order and exact 2^-7 scalings instead of two rsqrt(128) factors
    assert float((z1 - z3).abs().max()) < 2e-6 and float((u1 - u3).abs().max()) < 2e-6
    assert torch.equal(z1[2], zin[2]) and torch.equal(u1[2], u0[2])                       # stopped slice untouched
    zf = O.fft2c(xr + u0)                                                                  # env.py:87-93 restated
    temp = (mu.view(n, 1, 1, 1) * zf + y0) / (1 + mu.view(n, 1, 1, 1))
    zo = O.ifft2c(torch.where(masks.reshape(n, 1, h, w), temp, zf))
    uo = u0 + xr - zo
    live = [0, 1, 3, 4]
    assert float((z1[live] - zo[live]).abs().max()) < 3e-6 and float((u1[live] - uo[live]).abs().max()) < 3e-6


def test_fused_first_and_last_layer_match_their_own_kernels(sd_np, monkeypatch):
    """At the headline size the first layer (2 -> 32) is evaluated inside inc.conv-1's staging and the last one (1x1, residual, clamp)
    inside up4.conv-2's epilogue; `PNP_NO_F4_FUSED_FIRST` / `PNP_NO_F4_FUSED_LAST` put them back on their own kernels.  Same
    arithmetic up to the order of a few additions: the denoiser output agrees to 2e-6, and both agree with the oracle."""
    n, h, w = 3, 256, 256
    x = (torch.from_numpy(synthetic.hash_uniform(23, 7, n * h * w).reshape(n, 1, h, w)) + 1) * 0.5
    sigma = torch.linspace(4, 40, n) / 255.0
    outs = {}
    for fused in (True, False):
        if not fused:
            monkeypatch.setenv("PNP_NO_F4_FUSED_FIRST", "1")
            monkeypatch.setenv("PNP_NO_F4_FUSED_LAST", "1")
        e = _engine(n, h, w, sd_np)
        algos = e.conv_algorithms()
        assert algos[1] == 4 and algos[26] == (4 if fused else 1), algos   # inc.conv-1 / up4.conv-2: the layers that carry the fusions
        outs[fused] = e.denoise(x.cuda(), sigma.cuda()).cpu()
    # FLOAT TOLERANCE: f32, different summation order in two layers
    np.testing.assert_allclose(outs[True].numpy(), outs[False].numpy(), rtol=0, atol=2e-6)
    ref = O.denoise(O.torch_weights(sd_np), x, sigma)
    np.testing.assert_allclose(outs[True].numpy(), ref.numpy(), rtol=0, atol=1e-5)


# ---- round 4: repeatability -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("size", [(32, 256, 256), (16, 512, 512), (6, 256, 128)])
@pytest.mark.parametrize("mode", ["f32", "bf16", "bf16-direct-kernels", "bf16-one-term"])
def test_denoiser_passes_are_bit_repeatable(sd_np, mode, size, monkeypatch):
    """Passes over the same input give the same bits on every kernel family: at a chip-filling size, on the 16 x 512 x 512 plan BASELINE
    configs[4] is timed on, and on a non-square size (tile rows and columns differ) - no kernel depends on workgroup timing.
    Round 4 met variants that passed every oracle tolerance and were not repeatable, and shipped the separable producers of
    conv3x3_bf16ws_kernel behind an inline-assembly workaround with the cause open.  Round 5 found it (profiles/r05_race.md): a packed-FP32
    operand form hipcc emits reads zero in lanes 48-63 now and then beside bf16 MFMAs - an instruction-level fault, reproduced stand-alone;
    the kernels keep off that form (conv_staging.h lerp_np) and tools/isa_audit.py checks the built library for it."""
    from dt4image_restoration_amd.engine import PnPEngine
    n, h, w = size
    if mode == "bf16-direct-kernels":
        monkeypatch.setenv("PNP_BF16_NO_WS", "1")
    if mode == "bf16-one-term":
        monkeypatch.setenv("PNP_BF16_W1", "1")
    e = PnPEngine(n, h, w, bf16_convs=mode != "f32")
    e.load_weights(sd_np)
    x = ((torch.from_numpy(synthetic.hash_uniform(31, 7, n * h * w).reshape(n, 1, h, w)) + 1) * 0.5).cuda()
    sigma = (torch.linspace(4, 55, n) / 255.0).cuda()
    first = e.denoise(x, sigma).clone()
    for _ in range(40):                                     # (a pass is ~5 ms; the faulty build differs in a third to all of its passes)
        assert torch.equal(e.denoise(x, sigma), first)


@pytest.mark.parametrize("shape", [(1, 128, 128), (5, 256, 256), (3, 96, 80)])
@pytest.mark.parametrize("bf16", [False, True])
def test_splitk_inlaunch_combine_is_bit_identical(sd_np, shape, bf16, monkeypatch):
    """PNP_SPLITK_INLAUNCH=1 (small problems): the split-K planes are summed inside the conv launch by the last workgroup of a
    tile to arrive - agent-scope stores / loads around one relaxed atomic, no fence - in the plane order of splitk_reduce_kernel:
    the same bits as the two-launch path, pass after pass."""
    from dt4image_restoration_amd.engine import PnPEngine
    n, h, w = shape
    x = ((torch.from_numpy(synthetic.hash_uniform(57, 3, n * h * w).reshape(n, 1, h, w)) + 1) * 0.5).cuda()
    sigma = (torch.linspace(6, 50, n) / 255.0).cuda()
    ref_engine = PnPEngine(n, h, w, bf16_convs=bf16)
    ref_engine.load_weights(sd_np)
    assert 0 in ref_engine.conv_algorithms()                # some layers run the direct kernel (split-K on these sizes)
    want = ref_engine.denoise(x, sigma).clone()
    monkeypatch.setenv("PNP_SPLITK_INLAUNCH", "1")
    e = PnPEngine(n, h, w, bf16_convs=bf16)
    monkeypatch.delenv("PNP_SPLITK_INLAUNCH")
    e.load_weights(sd_np)
    for _ in range(25):
        assert torch.equal(e.denoise(x, sigma), want)
