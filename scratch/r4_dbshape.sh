#!/bin/bash
# round 4: k_deblock band shapes with 4 pictures per workgroup: 2 rows x 4 pictures, 4 rows x 2 pictures (2 groups), 8 rows x 1 picture (4 groups)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp p264decoder_amd/libp264amd.so scratch/lib_cur.so
bash scratch/variants_run.sh "cur cur:P264AMD_DEBLOCK_RB_LOG2=2 cur:P264AMD_DEBLOCK_RB_LOG2=3 cur:P264AMD_DEBLOCK_RB_LOG2=2,P264AMD_DEBLOCK_WAVES=12 cur:P264AMD_DEBLOCK_RB_LOG2=3,P264AMD_DEBLOCK_WAVES=12" 1024 2>&1 | tee gpurun_out/r4_dbshape.log
