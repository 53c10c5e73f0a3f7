"""GPU box diagnostic: per-stage error of the bf16 mode against the oracle's bf16 mode + repeatability, for the library in PNP_LIB_PATH."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dt4image_restoration_amd.engine import PnPEngine
from dt4image_restoration_amd.weights import generate_unet_weights
from dt4image_restoration_amd import synthetic
from oracle import pnp_oracle as O

n, h, w = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sdn = generate_unet_weights(0, "unit_gain")
e = PnPEngine(n, h, w, bf16_convs=True, keep_stages=True); e.load_weights(sdn)
terms = e.bf16_weight_terms()
x = (torch.from_numpy(synthetic.hash_uniform(5, h * 100 + w, n * h * w).reshape(n, 1, h, w)) + 1) * 0.5
sigma = torch.linspace(5, 50, n) / 255.0
nm = torch.ones(n, 1, h, w) * sigma.view(n, 1, 1, 1)
with torch.no_grad():
    _, stages = O.unet_forward(O.torch_weights(sdn), torch.cat([x, nm], 1), return_stages=True, bf16_operands=O.Bf16Plan(weight_terms=terms))
e.denoise(x.cuda(), sigma.cuda())
first = [e.read_stage(k).clone() for k in range(9)]
errs = [float((a.cpu() - r).abs().max()) / float(r.abs().max()) for a, r in zip(first, stages.values())]
e.denoise(x.cuda(), sigma.cuda())
rep = [int((e.read_stage(k) != first[k]).sum()) for k in range(9)]
print(os.environ.get("PNP_LIB_PATH", "in-tree").split("/")[-1], "terms", terms, "rel err per stage", ["%.1e" % v for v in errs], "repeat diffs", rep)
if errs[8] > 1e-2:
    d = (first[8].cpu() - stages["y4"]).abs()
    nn, cc, yy, xx = torch.nonzero(d > 0.02 * float(stages["y4"].abs().max()), as_tuple=True)
    print("  y4 bad elements", len(nn), "x % 32 hist", torch.bincount(xx % 32, minlength=32).tolist(), "y % 8 hist", torch.bincount(yy % 8, minlength=8).tolist())
