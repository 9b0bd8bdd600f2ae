#!/bin/bash
# round 4: locality band height of the MC work lists at 2048 streams
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp p264decoder_amd/libp264amd.so scratch/lib_cur.so
bash scratch/variants_run.sh "cur cur:P264AMD_MC_BAND_LOG2=3 cur:P264AMD_MC_BAND_LOG2=2 cur:P264AMD_MC_BAND_LOG2=5 cur:P264AMD_MC_BAND_LOG2=3,P264AMD_MC_WGS_PER_PIC=64" 2048 2>&1 | tee gpurun_out/r4_mcband.log
