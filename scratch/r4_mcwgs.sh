#!/bin/bash
# round 4: workgroups per picture of the fused MC launch at 2048 streams (24 by default there; 48 at 1024 streams)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp p264decoder_amd/libp264amd.so scratch/lib_cur.so
bash scratch/variants_run.sh "cur cur:P264AMD_MC_WGS_PER_PIC=32 cur:P264AMD_MC_WGS_PER_PIC=48 cur:P264AMD_MC_WGS_PER_PIC=64 cur:P264AMD_MC_WGS_PER_PIC=96 cur:P264AMD_MC_WGS_PER_PIC=16" 2048 2>&1 | tee gpurun_out/r4_mcwgs.log
