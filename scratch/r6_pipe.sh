#!/bin/bash
# round 6: the pipeline with one host -> HBM copy per picture (parser arrays laid out like an input slot); A/B by P264AMD_PIPE_DEBUG waits
export P264AMD_PIPE_DEBUG=1
for i in 1 2 3; do
  python -m p264decoder_amd.tools.pipe_bench --streams 128 --threads 16 --pictures 72 --device 0 2>&1 | grep -v "^p264amd"
  python -m p264decoder_amd.tools.pipe_bench --streams 128 --threads 16 --pictures 72 --device -1 2>&1 | grep -v "^p264amd"
done
