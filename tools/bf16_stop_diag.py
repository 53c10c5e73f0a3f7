"""GPU box diagnostic: where do live slices of a step with stopped slices differ from the step with nobody stopped (bf16 mode)?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dt4image_restoration_amd.engine import PnPEngine
from dt4image_restoration_amd.weights import generate_unet_weights
from dt4image_restoration_amd import synthetic

n, h, w = 64, 256, 256
sdn = generate_unet_weights(0, "unit_gain")
e = PnPEngine(n, h, w, bf16_convs=True); e.load_weights(sdn)
print("algos", e.conv_algorithms()[1:27], "terms", e.bf16_weight_terms())
x = ((torch.from_numpy(synthetic.hash_uniform(9, 64256, n * h * w).reshape(n, 1, h, w)) + 1) * 0.5).cuda()
sigma = (torch.linspace(3, 60, n) / 255.0).cuda()
d0 = e.denoise(x, sigma).clone()
for rep in range(3):
    d1 = e.denoise(x, sigma)
    print("denoise repeat", rep, "differs:", int((d1 != d0).sum()))
data = synthetic.make_problem(n, h, w, accel=4.0, seed=99)
x0, z0, u0 = e.reset(torch.view_as_complex(torch.from_numpy(data["x0"])).cuda(),
                     torch.view_as_complex(torch.from_numpy(data["y0"])).cuda(), torch.from_numpy(data["mask"]).cuda())
mu, sg = torch.full((n,), 0.4).cuda(), torch.full((n,), 0.08).cuda()
e.step(x0, z0, u0, mu, sg)
xb, zb, ub = x0.clone(), z0.clone(), u0.clone()
e.step(xb, zb, ub, mu, sg)
for pat in range(4):
    tact = torch.zeros(n)
    if pat < 3:
        tact[pat::3] = 1.0
    xa, za, ua = x0.clone(), z0.clone(), u0.clone()
    e.step(xa, za, ua, mu, sg, t_action=tact.cuda())
    torch.cuda.synchronize()
    stop = tact > 0.5
    diff = (xa != xb)
    diff[stop] = False
    per = diff.flatten(1).sum(1).cpu()
    print("pattern", pat, "live slices with differences:", [(int(i), int(c)) for i, c in enumerate(per) if c > 0][:20])
    if int(per.sum()) > 0:
        i = int(torch.nonzero(per)[0])
        ys, xs = torch.nonzero(diff[i, 0], as_tuple=True)
        print("  slice", i, "rows", int(ys.min()), int(ys.max()), "cols", int(xs.min()), int(xs.max()), "max abs", float((xa[i] - xb[i]).abs().max()))
