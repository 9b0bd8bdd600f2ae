#!/bin/bash
# round 4: is k_mc's memory skeleton (luma interpolation replaced by a copy out of the staged window, no residual: -DEXPM_LUMA_COPY=1
# -DEXPM_RESID=0, results wrong) bound by bytes in flight?  The same skeleton at 4 / 6 / 8 wavefronts per SIMD (-DMC_WAVES_PER_EU):
#   scratch/variant.sh sk4 -DEXPM_LUMA_COPY=1 -DEXPM_RESID=0; scratch/variant.sh sk6 ... -DMC_WAVES_PER_EU=6; scratch/variant.sh sk8 ... =8
#   scratch/variant.sh full4; scratch/variant.sh full5 -DMC_WAVES_PER_EU=5
# MC stage per 2048 pictures: full kernel 5.24 ms (five wavefronts: 6.07), skeleton 4.68 / 4.70 / 4.69 ms - occupancy changes nothing:
# (nor does the range of the vectors: r4_mvrange.sh)
cd $GRAFT_REPO_ROOT
NOGOLD=1 STEPS=20 bash scratch/variants_run.sh "full4 full5 sk4 sk6 sk8" 2048
