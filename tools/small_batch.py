#!/usr/bin/env python3
"""Launch-bound regime (round-2 review item 6): one `pnp_step` on small batches - the reference's own N = 1, 128 x 128 call
shape (eval.py:232, main.py:232) and the tree search's 5-child step - eager, replayed from a hipGraph (torch.cuda.CUDAGraph
capture of the very same launches on the capture stream: bit-identical by construction, checked), and with the Winograd
workgroup gate varied.  Prints one JSON line per case.

    python tools/small_batch.py [--gate 192 64 1]
"""
import argparse, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_case(n, hw, steps, reps):
    import numpy as np, torch
    from dt4image_restoration_amd import synthetic, weights
    from dt4image_restoration_amd.engine import PnPEngine
    dev = torch.device("cuda", 0)
    sd = weights.generate_unet_weights(0, "unit_gain")
    data = synthetic.make_problem(n, hw, hw, accel=4.0, sigma_n=10.0 / 255.0, seed=1234)
    mu_tab, sg_tab = synthetic.param_table(n, steps, seed=77)
    eng = PnPEngine(n, hw, hw, device=0)
    eng.load_weights(sd)
    x0 = torch.view_as_complex(torch.from_numpy(data["x0"])).to(dev)
    y0 = torch.view_as_complex(torch.from_numpy(data["y0"])).to(dev)
    mask = torch.from_numpy(data["mask"]).to(dev)
    mu = torch.from_numpy(mu_tab).to(dev).t().contiguous()
    sg = torch.from_numpy(sg_tab).to(dev).t().contiguous()

    def eager():
        x, z, u = eng.reset(x0, y0, mask)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(steps):
            eng.step(x, z, u, mu[t], sg[t])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps, x.clone()

    # graph: the step reads its parameters from static buffers; per step one small copy + one replay
    mu_s, sg_s = mu[0].clone(), sg[0].clone()
    xg, zg, ug = eng.reset(x0, y0, mask)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        eng.step(xg, zg, ug, mu_s, sg_s)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eng.step(xg, zg, ug, mu_s, sg_s)

    def graphed():
        x, z, u = eng.reset(x0, y0, mask)
        xg.copy_(x); zg.copy_(z); ug.copy_(u)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(steps):
            mu_s.copy_(mu[t]); sg_s.copy_(sg[t])
            g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps, xg.clone()

    eager(); graphed()
    te = sorted(eager()[0] for _ in range(reps))[reps // 2]
    tg = sorted(graphed()[0] for _ in range(reps))[reps // 2]
    same = bool(torch.equal(eager()[1], graphed()[1]))
    return {"n": n, "hw": hw, "steps": steps, "eager_ms_per_step": round(te * 1e3, 4), "graph_ms_per_step": round(tg * 1e3, 4),
            "graph_bit_identical": same, "algos": eng.conv_algorithms()[1:27], "gate": os.environ.get("PNP_WINO_MIN_BLOCKS", "192")}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--gate", nargs="*", default=["192"])
    ap.add_argument("--case", default=None)
    a = ap.parse_args()
    if a.case:
        n, hw = map(int, a.case.split("x"))
        print(json.dumps(run_case(n, hw, 10, 7)), flush=True)
    else:
        for gate in a.gate:                               # the gate is read at pnp_create: one process per setting
            for case in ("1x128", "5x128", "5x256"):
                env = dict(os.environ, PNP_WINO_MIN_BLOCKS=gate)
                subprocess.run([sys.executable, __file__, "--case", case], env=env, check=False)
