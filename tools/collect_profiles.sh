#!/bin/bash
# Runs ON THE GPU BOX:  gpurun -- 'bash tools/collect_profiles.sh TAG [bench.py arguments selecting the configuration]'
# e.g.  bash tools/collect_profiles.sh r04                                   (the default line: configs[1], 64 x 256 x 256, f32)
#       bash tools/collect_profiles.sh r04_512bf16 --size 512 --batch 16 --accel 8 --convs bf16
# Leaves under gpurun_out/: the bench line of that configuration, its per-layer table, the rocprofv3 kernel-trace stats of the
# same command and three counter passes (FETCH_SIZE, WRITE_SIZE, SQ) - separate runs with --kernel-trace only, no other trace
# domain - which tools/refresh_profiles.py folds into profiles/.
set -e -o pipefail
TAG=${1:-r05}; shift || true
ARGS=("$@")
R=$(pwd); G=$R/gpurun_out; mkdir -p "$G"
if [ ${#ARGS[@]} -eq 0 ]; then
  python3 bench.py > "$G/bench_$TAG.json" 2> "$G/bench_$TAG.err"
else
  python3 bench.py "${ARGS[@]}" --no-greedy > "$G/bench_$TAG.json" 2> "$G/bench_$TAG.err"
fi
echo "bench done: $(cut -c1-120 "$G/bench_$TAG.json")"
python3 bench.py "${ARGS[@]}" --no-cpu-baseline --no-greedy --no-config4 --dump-layers "$G/layers_$TAG.json" > "$G/bench_layers_$TAG.json" 2>> "$G/bench_$TAG.err"   # an event pair per launch
echo "layer table done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$G/prof_$TAG" -o runc -- python3 "$R/bench.py" "${ARGS[@]}" --steps 10 --warmup 2 --reps 3 --no-cpu-baseline --no-greedy --no-config4 > "$G/bench_prof_$TAG.json" 2> "$G/prof_$TAG.err"
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$G/pmcF_$TAG" -o runc -- python3 "$R/bench.py" "${ARGS[@]}" --steps 2 --warmup 1 --reps 1 --no-cpu-baseline --no-greedy --no-config4 > /dev/null 2> "$G/pmcF_$TAG.err"
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$G/pmcW_$TAG" -o runc -- python3 "$R/bench.py" "${ARGS[@]}" --steps 2 --warmup 1 --reps 1 --no-cpu-baseline --no-greedy --no-config4 > /dev/null 2> "$G/pmcW_$TAG.err"
echo "write done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$G/pmcS_$TAG" -o runc -- python3 "$R/bench.py" "${ARGS[@]}" --steps 2 --warmup 1 --reps 1 --no-cpu-baseline --no-greedy --no-config4 > /dev/null 2> "$G/pmcS_$TAG.err"
echo "sq done"
