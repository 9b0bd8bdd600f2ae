#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp p264decoder_amd/libp264amd.so scratch/lib_cur.so
bash scratch/variants_run.sh "cur cur:P264AMD_DEBLOCK_RB_LOG2=3 cur:P264AMD_DEBLOCK_RB_LOG2=1 cur:P264AMD_DEBLOCK_WAVES=12 cur:P264AMD_DEBLOCK_RB_LOG2=3,P264AMD_DEBLOCK_WAVES=12" 2048 2>&1 | tee gpurun_out/r4_dbshape3.log
