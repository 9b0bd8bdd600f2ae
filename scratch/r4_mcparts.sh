#!/bin/bash
# round 4: what k_mc's time is made of - timing builds (results wrong) without the sample stores / without the reference-window
# loads / with only the luma or only the chroma roles working, of the full kernel and of its skeleton (interpolation -> copy, no residual)
#   SK="-DEXPM_LUMA_COPY=1 -DEXPM_RESID=0"; scratch/variant.sh sk $SK; sk_nost: + -DEXPM_NO_STORE=1; sk_nowin: + -DEXPM_NO_WINDOW=1;
#   sk_none: both; sk_luma / sk_chroma: + -DEXPM_ONLY=1 / 2; full_luma / full_chroma / full_nost / full_nowin: the same without $SK
cd $GRAFT_REPO_ROOT
cp p264decoder_amd/libp264amd.so scratch/lib_full.so
NOGOLD=1 STEPS=20 bash scratch/variants_run.sh "full full_nost full_nowin full_luma full_chroma sk sk_nost sk_nowin sk_none sk_luma sk_chroma" 2048
