#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/collect_profiles.sh TAG N'): the default bench line, the rocprofv3 kernel
# stats of the same command and the three counter passes tools/refresh_profiles.py folds into profiles/.
# Counter passes are separate runs with --kernel-trace only (no other trace domains).
set -e -o pipefail
TAG=${1:-r03}; N=${2:-1}
R=$(pwd); G=$R/gpurun_out; mkdir -p "$G"
python3 bench.py > "$G/bench_$TAG.json" 2> "$G/bench_$TAG.err"
echo "bench done: $(cut -c1-120 "$G/bench_$TAG.json")"
python3 bench.py --no-cpu-baseline --no-greedy --dump-layers "$G/layers_$TAG.json" > "$G/bench_layers_$TAG.json" 2>> "$G/bench_$TAG.err"   # an event pair per launch
echo "layer table done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$G/prof_$TAG" -o runc -- python3 "$R/bench.py" --steps 10 --warmup 2 --reps 3 --no-cpu-baseline --no-greedy > "$G/bench_prof_$TAG.json" 2> "$G/prof_$TAG.err"
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$G/pmcF$N" -o runc -- python3 "$R/bench.py" --steps 2 --warmup 1 --reps 1 --no-cpu-baseline --no-greedy > /dev/null 2> "$G/pmcF$N.err"
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$G/pmcW$N" -o runc -- python3 "$R/bench.py" --steps 2 --warmup 1 --reps 1 --no-cpu-baseline --no-greedy > /dev/null 2> "$G/pmcW$N.err"
echo "write done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$G/pmcS$N" -o runc -- python3 "$R/bench.py" --steps 2 --warmup 1 --reps 1 --no-cpu-baseline --no-greedy > /dev/null 2> "$G/pmcS$N.err"
echo "sq done"
