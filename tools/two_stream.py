#!/usr/bin/env python3
"""Does overlapping the drain of one kernel with the next kernel of an INDEPENDENT half-batch pay?  64 slices of 256x256 as one
64-slice engine on one stream against two 32-slice engines stepped on two streams (and 4 x 16 on four)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dt4image_restoration_amd import synthetic, weights
from dt4image_restoration_amd.engine import PnPEngine

dev = torch.device("cuda", 0)
sd = weights.generate_unet_weights(0, "unit_gain")
N, HW, STEPS = 64, 256, 30
data = synthetic.make_problem(N, HW, HW, accel=4.0, sigma_n=10.0 / 255.0, seed=1234)
mu_tab, sg_tab = synthetic.param_table(N, STEPS, seed=77)
mask = torch.from_numpy(data["mask"]).to(dev)


def run(parts):
    n = N // parts
    engs, states, streams, mus, sgs = [], [], [], [], []
    for k in range(parts):
        e = PnPEngine(n, HW, HW, device=0)
        e.load_weights(sd)
        sl = slice(k * n, (k + 1) * n)
        x0 = torch.view_as_complex(torch.from_numpy(data["x0"][sl])).to(dev)
        y0 = torch.view_as_complex(torch.from_numpy(data["y0"][sl])).to(dev)
        engs.append(e); states.append(e.reset(x0, y0, mask)); streams.append(torch.cuda.Stream() if parts > 1 else torch.cuda.current_stream())
        mus.append(torch.from_numpy(mu_tab[sl]).to(dev).t().contiguous()); sgs.append(torch.from_numpy(sg_tab[sl]).to(dev).t().contiguous())
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(STEPS):
            for k in range(parts):
                with torch.cuda.stream(streams[k]):
                    x, z, u = states[k]
                    engs[k].step(x, z, u, mus[k][t], sgs[k][t])
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / STEPS)
    algos = engs[0].conv_algorithms()[1:27]
    return {"parts": parts, "slices_per_part": n, "ms_per_64_slice_step": round(best * 1e3, 4), "algos": "".join(str(a) for a in algos)}


for parts in (1, 2, 4):
    print(json.dumps(run(parts)), flush=True)
