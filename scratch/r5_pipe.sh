#!/bin/bash
# round 5: where the end-to-end pipeline's time goes (P264AMD_PIPE_DEBUG=1: the main thread's waits for the parsers / for the device)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in ${@:-"16:-1" "16:0" "14:0" "20:0" "32:0" "32:-1"}; do
  th=${cfg%%:*}; dev=${cfg##*:}
  echo "threads $th device $dev:"
  P264AMD_PIPE_DEBUG=1 python -m p264decoder_amd.tools.pipe_bench --streams 128 --threads $th --pictures 24 --device $dev 2>&1 | grep -v "^$" | tail -3
done 2>&1 | tee gpurun_out/r5_pipe.log
