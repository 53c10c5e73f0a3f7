set -e -o pipefail
G=gpurun_out
python3 bench.py --size 512 --batch 16 --accel 8 --steps 50 --no-greedy > $G/x_512_f32.json 2> $G/x_512_f32.err
echo 512f32 done
python3 bench.py --size 512 --batch 16 --accel 8 --steps 50 --convs bf16 --no-greedy > $G/x_512_bf16.json 2> $G/x_512_bf16.err
echo 512bf16 done
python3 bench.py --size 128 --batch 1 --steps 10 --no-greedy > $G/x_config0.json 2> $G/x_config0.err
echo config0 done
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu-baseline > $G/x_torchrun1.json 2> $G/x_torchrun1.err
echo torchrun done
python3 bench.py --mode greedy --no-cpu-baseline > $G/x_greedy.json 2> $G/x_greedy.err
echo greedy done
python3 tools/mcts_scale.py > $G/x_mcts.json 2> $G/x_mcts.err
echo mcts done
python3 bench.py --convs bf16 --no-greedy > $G/x_bf16_64x256.json 2> $G/x_bf16_64x256.err
python3 bench.py --convs bf16 --no-greedy --no-cpu-baseline --dump-layers $G/x_layers_bf16_64x256.json > $G/x_bf16_64x256_layers_run.json 2>> $G/x_bf16_64x256.err
echo bf16-64x256 done
