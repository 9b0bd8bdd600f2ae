#!/bin/bash
# round 4: what k_mc's time is made of - luma interpolation replaced by a copy out of the staged window, no residual (timing only)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
NOGOLD=1 bash scratch/variants_run.sh "mbase mcopy mnores mcopynores" 1024 2>&1 | tee gpurun_out/r4_mcexp.log
