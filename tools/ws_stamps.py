"""GPU box, diagnostic build (-DPNP_WS_STAMPS of conv_bf16_kernels.hip): s_memtime stamps of one workgroup's consumer wave 0 and
producer wave 4 per producer/consumer launch of one bf16 denoiser pass.  PNP_LIB_PATH=dt4image_restoration_amd/csrc/libpnpadmm_stamps.so python tools/ws_stamps.py"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dt4image_restoration_amd import _lib
from dt4image_restoration_amd.engine import PnPEngine
from dt4image_restoration_amd.unet_spec import UNET_LAYERS
from dt4image_restoration_amd.weights import generate_unet_weights
n, size = 64, 256
lib = _lib.load()
eng = PnPEngine(n, size, size, bf16_convs=True); eng.load_weights(generate_unet_weights(0))
x = torch.rand(n, 1, size, size, device="cuda"); sigma = torch.full((n,), 0.05, device="cuda")
eng.denoise(x, sigma); torch.cuda.synchronize()
lib.pnp_debug_ws_stamps_reset(); eng.denoise(x, sigma); torch.cuda.synchronize()
buf = np.zeros((64, 2, 128), np.uint64)
assert lib.pnp_debug_ws_stamps_read(C.c_void_p(buf.ctypes.data)) == 0
ws_layers = [l for l, a in zip(UNET_LAYERS, eng.conv_algorithms()) if a == 5]
for s, l in enumerate(ws_layers):
    c = buf[s, 0].astype(np.int64); p = buf[s, 1].astype(np.int64)
    nc, np_ = int((c > 0).sum()), int((p > 0).sum())
    t0 = min(c[0], p[0])
    print(f"== {l.key} cin {l.cin} cout {l.cout}: consumer stamps {nc}, producer stamps {np_}")
    print("  C:", " ".join(str(int(v - t0)) for v in c[:nc]))
    print("  P:", " ".join(str(int(v - t0)) for v in p[:np_]))
