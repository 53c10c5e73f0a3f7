#!/usr/bin/env python3
"""Fold the files `tools/collect_profiles.sh TAG ...` left under gpurun_out/ into profiles/ and print the headline numbers.

    python tools/refresh_profiles.py TAG ROUND [N H W MODE]

With N H W MODE (e.g. 16 512 512 bf16) the outputs carry the suffix _<N>x<H>x<W>_<MODE> - the names bench.py's pmc_traffic()
looks up for that configuration; without, they are the headline's (configs[1]: 64 x 256 x 256, f32) unsuffixed files."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
tag, RND = sys.argv[1], sys.argv[2]
if len(sys.argv) > 6:
    n, h, w, mode = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    SUF = f"_{n}x{h}x{w}_{mode}"
else:
    n, h, w, mode, SUF = 64, 256, 256, "f32", ""


def one(*pats):
    for p in pats:
        g = glob.glob(p)
        if g:
            return g[0]
    raise SystemExit(f"none of {pats}")


shutil.copy(one(f"{G}/prof_{tag}/runc/*_kernel_stats.csv", f"{G}/prof_{tag}/*kernel_stats.csv"), f"{P}/{RND}_kernel_stats{SUF}.csv")
shutil.copy(f"{G}/layers_{tag}.json", f"{P}/{RND}_layers{SUF}.json")
shutil.copy(f"{G}/bench_prof_{tag}.json", f"{P}/{RND}_bench_under_rocprof{SUF}.json")
shutil.copy(f"{G}/bench_{tag}.json", f"{P}/{RND}_bench{SUF}.json")


def load(d):
    return list(csv.DictReader(open(one(f"{G}/{d}/runc/*_counter_collection.csv", f"{G}/{d}/*counter_collection.csv"))))


out = {}
for d, name in ((f"pmcF_{tag}", "FETCH_SIZE"), (f"pmcW_{tag}", "WRITE_SIZE")):
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for r in load(d):
        if r["Counter_Name"] == name:
            k = r["Kernel_Name"].split("(")[0]
            agg[k] += float(r["Counter_Value"])
            cnt[k] += 1
    out[name] = {k: (agg[k] / cnt[k], cnt[k]) for k in agg}
tf = tw = ff = fw = 0
rows = []
STEPS = 3        # the PMC passes run bench.py --steps 2 --warmup 1 --reps 1: 3 steps
for k, (v, c) in out["FETCH_SIZE"].items():
    conv, fft = "conv3x3" in k, ("fft_rows_kernel" in k or "fft_cols_kernel" in k or "admm_" in k)
    if not (conv or fft):
        continue
    wv = out["WRITE_SIZE"].get(k, (0, 0))[0]
    if conv:
        tf += 2 * v * 1024 * c / STEPS
        tw += wv * 1024 * c / STEPS
    else:
        ff += 2 * v * 1024 * c / STEPS
        fw += wv * 1024 * c / STEPS
    rows.append((k[5:] if k.startswith("pnp::") else k, c / STEPS, round(2 * v * 1024 / 1e6, 1), round(wv * 1024 / 1e6, 1)))
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 2 --warmup 1 --reps 1; FETCH_SIZE x2 "
                     "(gfx950: the counter tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section), both x1024 (KB units)",
           "configuration": {"slices": n, "h": h, "w": w, "mode": mode},
           "conv_kernels_fetch_bytes_per_step": tf, "conv_kernels_write_bytes_per_step": tw,
           "fft_kernels_fetch_bytes_per_step": ff, "fft_kernels_write_bytes_per_step": fw,
           "fft_algorithmic_bytes_per_step": 37 * n * h * w,
           "per_kernel_launches_per_step_fetchMB_writeMB": rows},
          open(f"{P}/{RND}_traffic{SUF}.json", "w"), indent=1)

rows = load(f"pmcS_{tag}")
kt = {}
for r in csv.DictReader(open(one(f"{G}/pmcS_{tag}/runc/*_kernel_trace.csv", f"{G}/pmcS_{tag}/*kernel_trace.csv"))):
    kt[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]] += float(r["Counter_Value"])
for d, (dur, name) in kt.items():
    agg[name.split("(")[0]]["dur_ns"] += dur
    cnt[name.split("(")[0]] += 1
lines = ["| kernel | launches | avg us | clock GHz | MFMA pipe busy | VALU issue busy (incl. MFMA issue) | LDS busy | bank-conflict share of LDS cycles | bank-conflict cycles / kernel cycles | WAIT_ANY | WAIT_INST_ANY |",
         "|---|---|---|---|---|---|---|---|---|---|---|"]
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["dur_ns"]):
    if "conv3x3" not in k and "fft_" not in k and "admm_" not in k:
        continue
    c = cnt[k]
    dur = v["dur_ns"] / c
    gui = v["GRBM_GUI_ACTIVE"] / c
    cyc = gui / 8                                     # kernel duration in core cycles (GRBM_GUI_ACTIVE sums the 8 XCDs)
    lines.append("| %s | %d | %.0f | %.2f | %.3f | %.3f | %.3f | %.3f | %.3f | %.2f | %.2f |" % (
        k[5:] if k.startswith("pnp::") else k, c, dur / 1e3, cyc / dur, v["SQ_VALU_MFMA_BUSY_CYCLES"] / c / (cyc * 256 * 4),
        v["SQ_ACTIVE_INST_VALU"] * 4 / c / (cyc * 1024), v["SQ_LDS_IDX_ACTIVE"] / c / (cyc * 256),
        v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1), v["SQ_LDS_BANK_CONFLICT"] / c / (cyc * 256),
        v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"]))
open(f"{P}/{RND}_pmc_current{SUF}.md", "w").write(
    f"# SQ counters of the conv and data-fidelity kernels, {n} x {h} x {w}, {mode} (own rocprofv3 --pmc pass, bench.py --steps 2 --warmup 1 --reps 1)\n\n"
    "Normalisation (per kernel launch, to the kernel's duration in core cycles = GRBM_GUI_ACTIVE / 8 XCDs): MFMA pipe busy = "
    "SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs); VALU issue busy = SQ_ACTIVE_INST_VALU x 4 / (cycles x 1024 SIMDs); LDS busy = "
    "SQ_LDS_IDX_ACTIVE / (cycles x 256 CUs); bank-conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.\n\n"
    + "\n".join(lines) + f"\n\nHBM traffic of the conv kernels per step: {tf / 1e9:.2f} GB fetched (FETCH_SIZE x2) + {tw / 1e9:.2f} GB written.\n"
    f"HBM traffic of the data-fidelity kernels per step: {ff / 1e6:.1f} MB fetched (FETCH_SIZE x2) + {fw / 1e6:.1f} MB written "
    f"(algorithmic: {37 * n * h * w / 1e6:.1f} MB; the three-launch path moves {81 * n * h * w / 1e6:.1f} MB through L2).\n")
d = json.load(open(f"{P}/{RND}_bench{SUF}.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["algorithmic_tflops"], "conv traffic GB", tf / 1e9, tw / 1e9,
      "fft traffic MB", ff / 1e6, fw / 1e6)
print("\n".join(lines))
