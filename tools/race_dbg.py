#!/usr/bin/env python3
"""GPU box, diagnostic build (-DPNP_WS_LERP_PACKED -DPNP_WS_RACE_DBG of conv_bf16_kernels.hip, round 5): the separable producers of
conv3x3_bf16ws_kernel check every patch row they store (second evaluation from the same registers, LDS read-back, row weights re-read) and
record the first mismatches; this prints them next to the exact candidates computed on the host.

    PNP_LIB_PATH=.../libpnpadmm_dbgpacked.so python tools/race_dbg.py [--passes 20] [--size 16x512x512]
"""
import argparse, ctypes as C, json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dt4image_restoration_amd import synthetic, weights, _lib  # noqa: E402
from dt4image_restoration_amd.engine import PnPEngine            # noqa: E402

REC = 32


def f32(u):
    return np.array([u], dtype=np.uint32).view(np.float32)[0]


def bf16_rne(x):
    u = np.array([x], dtype=np.float32).view(np.uint32)[0]
    return int(((int(u) + 0x7FFF + ((int(u) >> 16) & 1)) >> 16) & 0xFFFF)


def fma32(a, b, c):
    return np.float32(np.float64(a) * np.float64(b) + np.float64(c))   # a * b exact in f64


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--passes", type=int, default=20)
    ap.add_argument("--size", default="16x512x512")
    ap.add_argument("--show", type=int, default=24)
    args = ap.parse_args()
    n, h, w = (int(v) for v in args.size.split("x"))
    lib = _lib.load()
    x = ((torch.from_numpy(synthetic.hash_uniform(31, 7, n * h * w).reshape(n, 1, h, w)) + 1) * 0.5).cuda()
    sigma = (torch.linspace(4, 55, n) / 255.0).cuda()
    e = PnPEngine(n, h, w, bf16_convs=True)
    e.load_weights(weights.generate_unet_weights(0, "unit_gain"))
    buf = np.zeros(4 + 256 * REC, dtype=np.uint32)
    first = None
    shown = 0
    stats = {}
    addends = []
    for p in range(args.passes):
        assert lib.pnp_debug_race_reset() == 0
        out = e.denoise(x, sigma)
        torch.cuda.synchronize()
        assert lib.pnp_debug_race_read(C.c_void_p(buf.ctypes.data)) == 0
        if first is None:
            first = out.clone()
        nbad, nrows = int(buf[0]), int(buf[1])
        print(json.dumps({"pass": p, "row_checks_by_lane0": nrows, "mismatching_lane_rows": nbad, "output_equals_first": bool(torch.equal(out, first))}), flush=True)
        for k in range(min(nbad, 256)):
            r = buf[4 + k * REC: 4 + (k + 1) * REC]
            # summary key: (H, row, wrong channels, what the wrong value equals, lane quarter)
            wa_, wb_ = f32(r[4]), f32(r[5])
            st = [int(r[14]) & 0xFFFF, int(r[14]) >> 16, int(r[15]) & 0xFFFF, int(r[15]) >> 16]
            e2 = [int(r[16]) & 0xFFFF, int(r[16]) >> 16, int(r[17]) & 0xFFFF, int(r[17]) >> 16]
            what = []
            for c in range(4):
                if st[c] != e2[c]:
                    a_ = bf16_rne(np.float32(wa_ * f32(r[6 + c])))
                    b_ = bf16_rne(np.float32(wb_ * f32(r[10 + c])))
                    what.append("xyzw"[c] + ("=wa*ha" if st[c] == a_ else "=wb*hb" if st[c] == b_ else "=other(%04x)" % st[c]))
            # the addend the wrong FMA must have seen: X = o_wrong - wa * ha (f64), per wrong channel
            for c in range(4):
                if st[c] != e2[c]:
                    xadd = float(np.float64(f32(r[24 + c])) - np.float64(wa_) * np.float64(f32(r[6 + c])))
                    addends.append((int(r[3]) & 255, "xyzw"[c], xadd, float(np.float32(wb_ * f32(r[10 + c])))))
            key = (int(r[23]), "wave %d" % (int(r[2]) >> 8), "round %d" % (int(r[3]) >> 16), "row %d" % (int(r[3]) & 255), ",".join(what), "quarter %d" % ((int(r[2]) & 255) // 16), int(r[0]))
            stats[key] = stats.get(key, 0) + 1
            if shown >= args.show:
                continue
            kind, blk, wl, pos = int(r[0]), int(r[1]), int(r[2]), int(r[3])
            wa, wb = f32(r[4]), f32(r[5])
            ha = [f32(r[6 + c]) for c in range(4)]
            hb = [f32(r[10 + c]) for c in range(4)]
            s = [int(r[14]) & 0xFFFF, int(r[14]) >> 16, int(r[15]) & 0xFFFF, int(r[15]) >> 16]     # first evaluation, as converted
            ev = [int(r[16]) & 0xFFFF, int(r[16]) >> 16, int(r[17]) & 0xFFFF, int(r[17]) >> 16]   # second evaluation
            g = [int(r[18]) & 0xFFFF, int(r[18]) >> 16, int(r[19]) & 0xFFFF, int(r[19]) >> 16]    # LDS
            cand_ab = [bf16_rne(fma32(wb, hb[c], np.float32(wa * ha[c]))) for c in range(4)]       # fma(wb, hb, wa * ha)
            cand_ba = [bf16_rne(fma32(wa, ha[c], np.float32(wb * hb[c]))) for c in range(4)]       # fma(wa, ha, wb * hb)
            print(f"  rec kind={kind} (1: LDS != stored, 2: 2nd evaluation != stored, 4: row weights changed) block {blk} wave {wl >> 8} lane {wl & 255} "
                  f"round {pos >> 16} group {(pos >> 8) & 255} row {pos & 255} H {int(r[23])}")
            print(f"      wa {wa!r} wb {wb!r}  re-read {f32(r[20])!r} {f32(r[21])!r}")
            print(f"      ha {[float(v) for v in ha]}\n      hb {[float(v) for v in hb]}")
            print("      stored  " + " ".join(f"{v:04x}" for v in s) + "   2nd eval " + " ".join(f"{v:04x}" for v in ev) + "   LDS " + " ".join(f"{v:04x}" for v in g))
            print("      host fma(wb,hb,wa*ha) " + " ".join(f"{v:04x}" for v in cand_ab) + "   fma(wa,ha,wb*hb) " + " ".join(f"{v:04x}" for v in cand_ba))
            shown += 1
    print("addend the wrong FMA saw (o_wrong - wa * ha in f64) next to the addend it should have seen (wb * hb), first 40:")
    for a in addends[:40]:
        print("   row %d %s  saw %.9g  wanted %.9g" % a)
    print("summary over the recorded mismatches: (H, wave, round, row, wrong channels, lane quarter, kind) -> count")
    for k, v in sorted(stats.items()):
        print("  ", k, v)


if __name__ == "__main__":
    main()
