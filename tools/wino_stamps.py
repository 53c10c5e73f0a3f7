"""Per-phase cycle accounting of the Winograd conv kernels (diagnostic build, `make -C dt4image_restoration_amd/csrc stamps`).

Wave 0 of 1024 mid-grid workgroups per launch records s_memtime at its phase boundaries; this prints, per layer, the mean
cycles a workgroup spends in: prologue (entry -> first chunk), per chunk {wait+commit, transform, MFMA issue}, epilogue
{drain+column transform, exchange+tile, stores}, its lifetime, and MFMA issue cycles / lifetime.

    PNP_LIB_PATH=dt4image_restoration_amd/csrc/libpnpadmm_stamps.so python tools/wino_stamps.py [batch] [size]
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

WGS, NST = 1024, 112


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    lib = _lib.load()
    if not hasattr(lib, "pnp_debug_stamps_read"):
        raise SystemExit("not a stamps build: set PNP_LIB_PATH to libpnpadmm_stamps.so")
    eng = PnPEngine(n, size, size)
    eng.load_weights(generate_unet_weights(0))
    x = torch.rand(n, 1, size, size, device="cuda")
    sigma = torch.full((n,), 0.05, device="cuda")
    eng.denoise(x, sigma)
    torch.cuda.synchronize()
    lib.pnp_debug_stamps_reset()
    eng.denoise(x, sigma)
    torch.cuda.synchronize()
    algo = eng.conv_algorithms()
    wino_layers = [l for l, a in zip(UNET_LAYERS, algo) if a == 1]
    slots = len(wino_layers)
    buf = np.zeros((slots, WGS, NST), np.uint64)
    rc = lib.pnp_debug_stamps_read(C.c_void_p(buf.ctypes.data), C.c_int(slots))
    assert rc == 0, rc
    print(f"{'layer':28s} {'cin':>4s} {'cout':>4s} {'chunks':>6s} {'P:issue':>7s} {'P:init':>7s} {'wait0':>6s} {'wait+commit':>11s} {'transf':>7s} "
          f"{'mfma':>7s} {'E:drain':>7s} {'E:gift':>7s} {'E:read':>7s} {'E:tile':>7s} {'E:store':>7s} {'life':>8s} {'pipe':>6s}")
    for s, l in enumerate(wino_layers):
        b = buf[s]
        ns = (b[:, NST - 1] >> np.uint64(32)).astype(np.int64)
        ok = ns > 0
        if not ok.any():
            print(f"{l.key:28s} no stamps (grid smaller than the sampled window)")
            continue
        k = int(ns[ok][0])
        if k > NST - 1:
            print(f"{l.key:28s} {k} stamps > buffer")
            continue
        t = b[ok][:, :k].astype(np.int64)
        nch = (k - 7) // 4
        ch = t[:, 2:2 + 4 * nch].reshape(-1, nch, 4)
        d = lambda i, j: (t[:, j] - t[:, i]).mean()
        commit = (ch[:, 1:, 1] - ch[:, 1:, 0]).mean() if nch > 1 else 0.0
        transf = (ch[:, :, 2] - ch[:, :, 1]).mean()
        mfma = (ch[:, :, 3] - ch[:, :, 2]).mean()
        e = 2 + 4 * nch - 1                   # index of the last MFMA stamp; then E0, Ea, Eb, E1, E2
        life = d(0, k - 1)
        nmfma = nch * (l.cin // nch // 8) * 8 * 4          # MFMAs per wave and workgroup
        print(f"{l.key:28s} {l.cin:4d} {l.cout:4d} {nch:6d} {d(0, 1):7.0f} {d(1, 2):7.0f} {(ch[:, 0, 1] - ch[:, 0, 0]).mean():6.0f} {commit:11.0f} "
              f"{transf:7.0f} {mfma:7.0f} {d(e, e + 1):7.0f} {d(e + 1, e + 2):7.0f} {d(e + 2, e + 3):7.0f} {d(e + 3, e + 4):7.0f} "
              f"{d(e + 4, e + 5):7.0f} {life:8.0f} {nmfma * 64 / life:6.3f}")
        if os.environ.get("STAMPS_PLACEMENT"):
            hw = (b[ok][:, NST - 1] & np.uint64(0xFFFFFFFF)).astype(np.int64)
            xcc = (b[ok][:, NST - 2] & np.uint64(0xF)).astype(np.int64)
            cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 0x7
            key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
            wgs = np.nonzero(ok)[0]
            t0 = t[:, 2]
            groups = {}
            for w, kk, tt in zip(wgs, key, t0):
                groups.setdefault(int(kk), []).append((int(w), int(tt)))
            print(f"{'':28s} {len(groups)} distinct CUs for {len(wgs)} workgroups; first CUs: " +
                  "; ".join(f"cu{k}: " + ",".join(f"wg{w}@{(tt - min(x[1] for x in v)):d}" for w, tt in v) for k, v in list(sorted(groups.items()))[:6]))
        if os.environ.get("STAMPS_PER_CHUNK"):
            for nm, j0, j1 in (("wait+commit", 0, 1), ("transform", 1, 2), ("mfma", 2, 3)):
                print(f"{'':28s} {nm:12s} per chunk: " + " ".join(f"{v:.0f}" for v in (ch[:, :, j1] - ch[:, :, j0]).mean(axis=0)))


if __name__ == "__main__":
    main()
