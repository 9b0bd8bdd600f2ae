#!/bin/bash
# round 4: the skeleton of k_deblock without any filter arithmetic - what is the 2.2 ms made of?  (timing only)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
NOGOLD=1 bash scratch/variants_run.sh "base nofilter nf_noh nf_nosync nf_novm nf_none nosync" 1024 2>&1 | tee gpurun_out/r4_dbexp2.log
