"""Per-phase cycle accounting of the F(4x4) conv kernel (diagnostic build, `make -C dt4image_restoration_amd/csrc stamps`).

Wave 0 of up to 1024 mid-grid workgroups per launch accumulates s_memtime differences: setup (entry -> chunk loop), and over
the chunk loop {wait + commit + barriers, transform + barrier, MFMA phase}, then the epilogue {partial transforms + LDS writes,
reduce + stores}.  Shares of the workgroup's lifetime, per layer.

    PNP_LIB_PATH=dt4image_restoration_amd/csrc/libpnpadmm_stamps.so python tools/wino4_stamps.py [batch] [size]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dt4image_restoration_amd import _lib  # noqa: E402
from dt4image_restoration_amd.engine import PnPEngine  # noqa: E402
from dt4image_restoration_amd.unet_spec import UNET_LAYERS  # noqa: E402
from dt4image_restoration_amd.weights import generate_unet_weights  # noqa: E402

WGS, NST = 1024, 16


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    lib = C.CDLL(_lib.LIB_PATH)
    if not hasattr(lib, "pnp_debug_stamps4_read"):
        raise SystemExit("not a stamps build: set PNP_LIB_PATH to libpnpadmm_stamps.so")
    eng = PnPEngine(n, size, size)
    eng.load_weights(generate_unet_weights(0))
    x = torch.rand(n, 1, size, size, device="cuda")
    sigma = torch.full((n,), 0.05, device="cuda")
    eng.denoise(x, sigma)
    torch.cuda.synchronize()
    lib.pnp_debug_stamps4_reset()
    eng.denoise(x, sigma)
    torch.cuda.synchronize()
    algo = eng.conv_algorithms()
    layers = [l for l, a in zip(UNET_LAYERS, algo) if a == 4]
    buf = np.zeros((len(layers), WGS, NST), np.uint64)
    rc = lib.pnp_debug_stamps4_read(C.c_void_p(buf.ctypes.data), C.c_int(len(layers)))
    assert rc == 0, rc
    print(f"{'layer':30s} {'cin':>4s} {'cout':>4s} {'chunks':>6s} {'setup':>7s} {'commit':>8s} {'transf':>8s} {'mfma':>8s} {'epi:wr':>8s} {'epi:red':>8s} "
          f"{'life':>8s}   shares: setup commit transf mfma epilogue | MFMA cycles (64 each) / life")
    for s, l in enumerate(layers):
        b = buf[s].astype(np.int64)
        ok = b[:, 0] == 1
        if not ok.any():
            print(f"{l.key:30s} no stamps")
            continue
        m = b[ok].mean(axis=0)
        setup, commit, trans, mfma, loop, ew, er, epi, life, nch = m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8], m[9], int(round(m[10]))
        per_chunk_mfma = 72 if l.cout >= 64 else 36
        pipe = nch * per_chunk_mfma * 64 / life
        print(f"{l.key:30s} {l.cin:4d} {l.cout:4d} {nch:6d} {setup:7.0f} {commit:8.0f} {trans:8.0f} {mfma:8.0f} {ew:8.0f} {er:8.0f} {life:8.0f}   "
              f"{setup / life:5.2f} {commit / life:5.2f} {trans / life:5.2f} {mfma / life:5.2f} {epi / life:5.2f} | {pipe:.3f} (x2 waves per SIMD)")


if __name__ == "__main__":
    main()
