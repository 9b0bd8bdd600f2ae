#!/bin/bash
# round 6: the coded levels at half their bytes (16 instead of 32 per block, read as one piece; wrong pictures) - what 8-bit levels could save at most
cd $GRAFT_REPO_ROOT
NOGOLD=1 STEPS=12 bash scratch/variants_run.sh "mbase mhalflv mbase mhalflv mbase mhalflv" 2048 2>&1 | tee gpurun_out/r6_exp2.log
