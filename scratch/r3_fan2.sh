#!/bin/bash
# config 5's streams through the fan-out, two ranks sharing the box's GPU over the TCP transport (RCCL needs one GPU per rank)
python -m p264decoder_amd.tools.fan_bench --rank 1 --world 2 --transport tcp --port 29777 --device 0 > gpurun_out/r3_fan_w1.log 2>&1 &
W=$!
timeout -k 5 200 python -m p264decoder_amd.tools.fan_bench --rank 0 --world 2 --transport tcp --port 29777 --device 0 --streams 8 --pictures 12 2>&1 | tail -3
wait $W; tail -2 gpurun_out/r3_fan_w1.log
