#!/bin/bash
# Runs ON THE GPU BOX: kernel-trace stats and one SQ counter pass of the bf16-operand conv mode (bench.py --convs bf16).
set -e -o pipefail
TAG=${1:-r03bf16}
R=$(pwd); G=$R/gpurun_out; mkdir -p "$G"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$G/prof_$TAG" -o runc -- python3 "$R/bench.py" --convs bf16 --steps 10 --warmup 2 --reps 3 --no-cpu-baseline --no-greedy > "$G/bench_prof_$TAG.json" 2> "$G/prof_$TAG.err"
echo "stats done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$G/pmcS_$TAG" -o runc -- python3 "$R/bench.py" --convs bf16 --steps 2 --warmup 1 --reps 1 --no-cpu-baseline --no-greedy > /dev/null 2> "$G/pmcS_$TAG.err"
echo "sq done"
