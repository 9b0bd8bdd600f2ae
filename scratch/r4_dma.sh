#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash scratch/variants_run.sh "pf_top pf_mid pf_top:P264AMD_DEBLOCK_RB_LOG2=2 pf_mid:P264AMD_DEBLOCK_RB_LOG2=2" 1024 2>&1 | tee gpurun_out/r4_dma.log
