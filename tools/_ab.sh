timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_env.py -x -q -k "bf16 or repeat or stop or config4" 2>&1 | tail -3
timeout -k 10 300 python tools/bf16_stop_stress.py 2>&1 | tail -3
AB_ARGS="--size 512 --batch 16 --accel 8 --steps 50 --warmup 3 --convs bf16 --no-cpu-baseline --no-greedy --reps 3" bash tools/ab_bench.sh f32e base:exp/abl/libpnpadmm_off2.so f32epi base2:exp/abl/libpnpadmm_off2.so f32epi2
