#!/bin/bash
# round 5: the parser's output buffers pinned (hipHostMalloc) or pageable, with and without the device
cd $GRAFT_REPO_ROOT
for cfg in "-1:0" "-1:1" "0:0" "0:1"; do
  dev=${cfg%%:*}; pin=${cfg##*:}
  echo -n "device $dev pinned $pin: "
  P264AMD_PIPE_PINNED=$pin P264AMD_PIPE_DEBUG=1 python -m p264decoder_amd.tools.pipe_bench --streams 128 --threads 16 --pictures 24 --device $dev 2>&1 | grep "24 rounds" | cut -c1-220
done 2>&1 | tee gpurun_out/r5_pipe2.log
