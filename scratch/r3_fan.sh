#!/bin/bash
python -m pytest tests/test_gpu_fanout.py -x -q 2>&1 | tail -5 || exit 1
# two ranks on the one GPU (rehearsal): the fan-out leg must come back with a report or the precise RCCL error
P264AMD_BENCH_DEVICE=0 P264AMD_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --streams 256 > gpurun_out/r3_n2.log 2>&1
tail -c 1500 gpurun_out/r3_n2.log
