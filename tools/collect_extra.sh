# Runs ON THE GPU BOX: the secondary records of a round (bash tools/collect_extra.sh [TAG]); files gpurun_out/x_<TAG>_*.json
set -e -o pipefail
T=${1:-r04}; G=gpurun_out
python3 bench.py --size 128 --batch 1 --steps 10 --no-greedy > $G/x_${T}_config0.json 2> $G/x_${T}_config0.err
echo config0 done
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu-baseline > $G/x_${T}_torchrun1.json 2> $G/x_${T}_torchrun1.err
echo torchrun done
python3 bench.py --mode greedy --no-cpu-baseline > $G/x_${T}_greedy.json 2> $G/x_${T}_greedy.err
echo greedy done
python3 tools/mcts_scale.py > $G/x_${T}_mcts.json 2> $G/x_${T}_mcts.err
echo mcts done
python3 bench.py --convs bf16 --no-greedy > $G/x_${T}_bf16_64x256.json 2> $G/x_${T}_bf16_64x256.err
PNP_BF16_W1=1 python3 bench.py --convs bf16 --no-greedy --no-cpu-baseline > $G/x_${T}_bf16_64x256_oneterm.json 2>> $G/x_${T}_bf16_64x256.err
PNP_BF16_W1=1 python3 bench.py --size 512 --batch 16 --accel 8 --steps 50 --warmup 3 --convs bf16 --no-greedy --no-cpu-baseline > $G/x_${T}_bf16_512_oneterm.json 2>> $G/x_${T}_bf16_64x256.err
echo bf16 done
