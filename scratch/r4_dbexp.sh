#!/bin/bash
# round 4: what do the inner edges / the strong filter / chroma cost in k_deblock?  (timing only: results of the variants are wrong)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
NOGOLD=1 bash scratch/variants_run.sh "base noinner nostrong nochroma edge0only nofilter" 1024 2>&1 | tee gpurun_out/r4_dbexp.log
