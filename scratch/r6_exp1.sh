#!/bin/bash
# round 6: upper bounds - k_mc without its residual section (what ANY compaction of the coded blocks could save at most), k_deblock
# without the strong filter (what a cheaper road for strength-4 edges could save at most); timing builds, wrong pictures
cd $GRAFT_REPO_ROOT
NOGOLD=1 STEPS=12 bash scratch/variants_run.sh "mbase mnores dnostrong mbase mnores dnostrong" 2048 2>&1 | tee gpurun_out/r6_exp1.log
