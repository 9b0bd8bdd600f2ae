#!/bin/bash
# round 5: kinds of host memory for the parser's picture buffers (P264AMD_HOST_ALLOC 0 coherent pinned / 1 non-coherent pinned / 2 registered pages)
cd $GRAFT_REPO_ROOT
for m in ${MODES:-0 1 2}; do
  for dev in -1 0; do
  echo -n "alloc mode $m device $dev: "
  P264AMD_HOST_ALLOC=$m P264AMD_PIPE_PINNED=1 P264AMD_PIPE_DEBUG=1 python -m p264decoder_amd.tools.pipe_bench --streams 128 --threads 16 --pictures 24 --device $dev 2>&1 | grep "24 rounds" | cut -c1-220
  done
done 2>&1 | tee gpurun_out/r5_pipe3.log
