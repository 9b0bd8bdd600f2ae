#!/bin/bash
# round 4: k_deblock by wavefronts per workgroup (34 bands of 2 rows x 4 pictures: 16 waves = rounds of 16+16+2) and by poll interval
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash scratch/variants_run.sh "base base:P264AMD_DEBLOCK_WAVES=14 base:P264AMD_DEBLOCK_WAVES=13 base:P264AMD_DEBLOCK_WAVES=12 base:P264AMD_DEBLOCK_WAVES=9 sleep4 sleep1 sleep4:P264AMD_DEBLOCK_WAVES=12" 1024 2>&1 | tee gpurun_out/r4_dbwaves.log
