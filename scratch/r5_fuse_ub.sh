#!/bin/bash
# round 5: what one pass from coefficients to filtered samples could save AT MOST - timing builds (wrong pictures) in which k_mc
# does not store its samples (EXPM_NO_STORE), k_deblock does not load them (EXPD_NO_SAMPLE_LOADS), and both:
#   scratch/variant.sh mc_nost -DEXPM_NO_STORE=1; scratch/variant.sh db_nold -DEXPD_NO_SAMPLE_LOADS=1; scratch/variant.sh fuse_ub -DEXPM_NO_STORE=1 -DEXPD_NO_SAMPLE_LOADS=1
cd $GRAFT_REPO_ROOT
cp p264decoder_amd/libp264amd.so scratch/lib_full.so
NOGOLD=1 STEPS=12 bash scratch/variants_run.sh "full mc_nost db_nold fuse_ub full fuse_ub" 2048 2>&1 | tee gpurun_out/r5_fuse_ub.log
