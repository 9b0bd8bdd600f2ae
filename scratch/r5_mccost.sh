#!/bin/bash
# round 5: the MC stage by the role-split weights of k_mc (cost of one chunk per role: luma MB : luma quadrant : chroma MB : chroma quadrant).
# here:  scratch/r5_mccost.sh build        -> scratch/lib_c<ym>_<yq>_<cm>_<cq>.so
# box:   gpurun -- 'bash scratch/r5_mccost.sh run'
SETS="7_10_5_9 8_10_5_9 6_10_5_9 7_12_5_9 7_8_5_9 7_10_6_9 7_10_4_9 7_10_5_11 7_10_5_7 8_10_6_9 6_10_4_9"
if [ "$1" = build ]; then
  for s in $SETS; do IFS=_ read a b c d <<< "$s"; bash scratch/variant.sh c$s -DMC_COST_YM=${a}u -DMC_COST_YQ=${b}u -DMC_COST_CM=${c}u -DMC_COST_CQ=${d}u | tail -1; done
  exit
fi
v=""; for s in $SETS; do v="$v c$s"; done
SKIP_TESTS=1 STEPS=10 bash scratch/r5_ab.sh "$v" "2048"
