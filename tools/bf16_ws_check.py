"""GPU box: bf16 mode at 64 x 256x256 (or N H W from the command line) - producer/consumer kernels vs the round-2 kernel (PNP_BF16_NO_WS) vs the bf16 oracle."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dt4image_restoration_amd.engine import PnPEngine
from dt4image_restoration_amd.weights import generate_unet_weights
from dt4image_restoration_amd import synthetic
from oracle import pnp_oracle as O
n, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 256, 256)
sdn = generate_unet_weights(0, "unit_gain")
x = ((torch.from_numpy(synthetic.hash_uniform(9, 64256, n * h * w).reshape(n, 1, h, w)) + 1) * 0.5)
sigma = torch.linspace(3, 60, n) / 255.0
e = PnPEngine(n, h, w, bf16_convs=True); e.load_weights(sdn)
a = e.denoise(x.cuda(), sigma.cuda()).cpu()
os.environ["PNP_BF16_NO_WS"] = "1"
e2 = PnPEngine(n, h, w, bf16_convs=True); e2.load_weights(sdn)
b = e2.denoise(x.cuda(), sigma.cuda()).cpu()
print("algos ws", e.conv_algorithms()); print("algos no", e2.conv_algorithms())
d = (a - b).abs()
print("ws vs no-ws: max %.3e mean %.3e" % (d.max(), d.mean()), "worst slice", int(d.reshape(n, -1).max(1)[0].argmax()))
ref = O.denoise(O.torch_weights(sdn), x[:8], sigma[:8], bf16_operands=True)
for nm, t in (("ws", a), ("no-ws", b)):
    dd = (t[:8] - ref).abs()
    print(nm, "vs bf16 oracle (8 slices): max %.3e mean %.3e" % (dd.max(), dd.mean()), "per-slice max", [round(float(v), 5) for v in dd.reshape(8, -1).max(1)[0]])
