#!/bin/bash
# A/B of library builds / environment switches on ONE GPU box (boxes differ by ~2 %): tools/ab_bench.sh TAG name[:lib[:ENV=V,...]] ...
#   lib: path relative to the repo root (default: the in-tree library); each variant runs `bench.py --dump-layers` once.
# Output: gpurun_out/bench_TAG_NAME.json, gpurun_out/layers_TAG_NAME.json and a summary line per variant.
tag=$1; shift
args=${AB_ARGS:---no-cpu-baseline --no-greedy --reps 5}
mkdir -p gpurun_out
for spec in "$@"; do
    IFS=: read -r name lib envs <<< "$spec"
    L=$PWD/dt4image_restoration_amd/csrc/libpnpadmm.so
    [ -n "$lib" ] && L=$PWD/$lib
    (
        IFS=, read -ra kv <<< "$envs"
        for e in "${kv[@]}"; do [ -n "$e" ] && export "$e"; done
        PNP_LIB_PATH=$L timeout -k 10 300 python bench.py $args --dump-layers gpurun_out/layers_${tag}_$name.json \
            > gpurun_out/bench_${tag}_$name.json 2> gpurun_out/bench_${tag}_$name.err
    ) || { echo "$name: bench failed"; tail -3 gpurun_out/bench_${tag}_$name.err; exit 1; }
    python - "$tag" "$name" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/bench_{sys.argv[1]}_{sys.argv[2]}.json"))
r = d["roofline"]
print(f"{sys.argv[2]:12s} {d['value']:8.3f} it/s  {d['ms_per_step']:.4f} ms/step  conv {r['kernel_ms_per_step']:.4f}  fft {d.get('roofline_fft', {}).get('kernel_ms_per_step')}  psnr {d['psnr_mean_db']}")
PY
done
