#!/bin/bash
# round 6: k_deblock with wavefront priorities by band inside a round (s_setprio); results are right, the build is a timing build
cd $GRAFT_REPO_ROOT
STEPS=12 bash scratch/variants_run.sh "mbase dprio mbase dprio mbase dprio" 2048 2>&1 | tee gpurun_out/r6_exp3.log
STEPS=12 bash scratch/variants_run.sh "mbase dprio" 256 2>&1 | tee -a gpurun_out/r6_exp3.log
