#!/usr/bin/env python3
"""Fold the files a GPU run left under gpurun_out/ into profiles/ (names prefixed with the round) and print the headline
numbers.  Usage: python tools/refresh_profiles.py <suffix of bench/prof dirs, e.g. r02a> <suffix of pmc dirs, e.g. 1> [round, default r02]"""
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
tag, pmc = sys.argv[1], sys.argv[2]
RND = sys.argv[3] if len(sys.argv) > 3 else "r03"

shutil.copy((glob.glob(f"{G}/prof_{tag}/runc/*_kernel_stats.csv") + glob.glob(f"{G}/prof_{tag}/*kernel_stats.csv"))[0], f"{P}/{RND}_kernel_stats.csv")
shutil.copy(f"{G}/layers_{tag}.json", f"{P}/{RND}_layers.json")
shutil.copy(f"{G}/bench_prof_{tag}.json", f"{P}/{RND}_bench_under_rocprof.json")
shutil.copy(f"{G}/bench_{tag}.json", f"{P}/{RND}_bench.json")


def load(d):
    return list(csv.DictReader(open((glob.glob(f"{G}/{d}/runc/*_counter_collection.csv") + glob.glob(f"{G}/{d}/*counter_collection.csv"))[0])))


out = {}
for d, name in ((f"pmcF{pmc}", "FETCH_SIZE"), (f"pmcW{pmc}", "WRITE_SIZE")):
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
    conv, fft = "conv3x3" in k, ("fft_rows_kernel" in k or "fft_cols_kernel" in k)
    if not (conv or fft):
        continue
    w = out["WRITE_SIZE"].get(k, (0, 0))[0]
    if conv:
        tf += 2 * v * 1024 * c / STEPS
        tw += w * 1024 * c / STEPS
    else:
        ff += 2 * v * 1024 * c / STEPS
        fw += w * 1024 * c / STEPS
    rows.append((k[5:], c / STEPS, round(2 * v * 1024 / 1e6, 1), round(w * 1024 / 1e6, 1)))
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 2 --warmup 1 --reps 1; FETCH_SIZE x2 "
                     "(gfx950: the counter tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section), both x1024 (KB units)",
           "conv_kernels_fetch_bytes_per_step": tf, "conv_kernels_write_bytes_per_step": tw,
           "fft_kernels_fetch_bytes_per_step": ff, "fft_kernels_write_bytes_per_step": fw,
           "fft_algorithmic_bytes_per_step": 37 * 64 * 256 * 256,
           "per_kernel_launches_per_step_fetchMB_writeMB": rows},
          open(f"{P}/{RND}_traffic.json", "w"), indent=1)

rows = load(f"pmcS{pmc}")
kt = {}
for r in csv.DictReader(open((glob.glob(f"{G}/pmcS{pmc}/runc/*_kernel_trace.csv") + glob.glob(f"{G}/pmcS{pmc}/*kernel_trace.csv"))[0])):
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
    if "conv3x3" not in k and "fft_" not in k:
        continue
    n = cnt[k]
    dur = v["dur_ns"] / n
    gui = v["GRBM_GUI_ACTIVE"] / n
    cyc = gui / 8                                     # kernel duration in core cycles (GRBM_GUI_ACTIVE sums the 8 XCDs)
    lines.append("| %s | %d | %.0f | %.2f | %.3f | %.3f | %.3f | %.3f | %.3f | %.2f | %.2f |" % (
        k[5:], n, dur / 1e3, cyc / dur, v["SQ_VALU_MFMA_BUSY_CYCLES"] / n / (cyc * 256 * 4),
        v["SQ_ACTIVE_INST_VALU"] * 4 / n / (cyc * 1024), v["SQ_LDS_IDX_ACTIVE"] / n / (cyc * 256),
        v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1), v["SQ_LDS_BANK_CONFLICT"] / n / (cyc * 256),
        v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"]))
open(f"{P}/{RND}_pmc_current.md", "w").write(
    "# SQ counters of the conv and FFT kernels, current build (own rocprofv3 --pmc pass, bench.py --steps 2 --warmup 1 --reps 1)\n\n"
    "Normalisation (per kernel launch, to the kernel's duration in core cycles = GRBM_GUI_ACTIVE / 8 XCDs): MFMA pipe busy = "
    "SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs); VALU issue busy = SQ_ACTIVE_INST_VALU x 4 / (cycles x 1024 SIMDs); LDS busy = "
    "SQ_LDS_IDX_ACTIVE / (cycles x 256 CUs); bank-conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (the round-1 table "
    "divided the conflict count, summed over all CUs, by SQ_BUSY_CYCLES, which is not per CU: those ratios of 0.65-0.88 were not "
    "fractions of anything).\n\n"
    + "\n".join(lines) + f"\n\nHBM traffic of the conv kernels per step: {tf / 1e9:.2f} GB fetched (FETCH_SIZE x2) + {tw / 1e9:.2f} GB written.\n"
    f"HBM traffic of the three data-fidelity kernels per step: {ff / 1e6:.1f} MB fetched (FETCH_SIZE x2) + {fw / 1e6:.1f} MB written "
    f"(algorithmic: {37 * 64 * 256 * 256 / 1e6:.1f} MB; moved through L2: {81 * 64 * 256 * 256 / 1e6:.1f} MB).\n")
d = json.load(open(f"{P}/{RND}_bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["algorithmic_tflops"], "conv traffic GB", tf / 1e9, tw / 1e9,
      "fft traffic MB", ff / 1e6, fw / 1e6)
print("\n".join(lines))
