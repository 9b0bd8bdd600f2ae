#!/bin/bash
# round 4: k_deblock_pool against k_deblock on the bench's launch
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp p264decoder_amd/libp264amd.so scratch/lib_cur.so
bash scratch/variants_run.sh "cur:P264AMD_DEBLOCK_POOL=0 cur:P264AMD_DEBLOCK_POOL=1 cur:P264AMD_DEBLOCK_POOL=1,P264AMD_DEBLOCK_PICS_PER_WG=2 cur" ${1:-1024} 2>&1 | tee gpurun_out/r4_pool.log
