"""GPU box: the persistent bf16 producer/consumer kernels under many stop patterns (every barrier count must still match between the
two roles of a workgroup): 64 x 256x256 and 16 x 512x512, tact = none / all / one live slice / random densities; live slices must equal
the step with nobody stopped, stopped ones must stay untouched.  Run under `timeout`."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dt4image_restoration_amd.engine import PnPEngine
from dt4image_restoration_amd.weights import generate_unet_weights
from dt4image_restoration_amd import synthetic

sdn = generate_unet_weights(0, "unit_gain")
for n, h, w in ((64, 256, 256), (16, 512, 512), (40, 128, 512)):
    e = PnPEngine(n, h, w, bf16_convs=True); e.load_weights(sdn)
    data = synthetic.make_problem(n, h, w, accel=4.0, seed=7)
    x0, z0, u0 = e.reset(torch.view_as_complex(torch.from_numpy(data["x0"])).cuda(),
                         torch.view_as_complex(torch.from_numpy(data["y0"])).cuda(), torch.from_numpy(data["mask"]).cuda())
    mu, sg = torch.full((n,), 0.4).cuda(), torch.full((n,), 0.08).cuda()
    e.step(x0, z0, u0, mu, sg)
    xb, zb, ub = x0.clone(), z0.clone(), u0.clone()
    e.step(xb, zb, ub, mu, sg)
    g = torch.Generator().manual_seed(1)
    pats = [torch.zeros(n), torch.ones(n)]
    one = torch.ones(n); one[n // 2] = 0; pats.append(one)
    first = torch.zeros(n); first[0] = 1; pats.append(first)
    for dens in (0.1, 0.5, 0.9, 0.97):
        for _ in range(3):
            pats.append((torch.rand(n, generator=g) < dens).float())
    for i, tact in enumerate(pats):
        xa, za, ua = x0.clone(), z0.clone(), u0.clone()
        e.step(xa, za, ua, mu, sg, t_action=tact.cuda())
        torch.cuda.synchronize()
        stop = tact > 0.5
        for a_, b_, o_ in ((xa, xb, x0), (za, zb, z0), (ua, ub, u0)):
            assert torch.equal(a_[stop], o_[stop]) and torch.equal(a_[~stop], b_[~stop]), (n, h, w, i)
    print(f"{n} x {h}x{w}: {len(pats)} stop patterns ok, algos {e.conv_algorithms()[1:27]}", flush=True)
