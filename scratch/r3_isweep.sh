#!/bin/bash
specs=""
for v in i5 i6 i7; do for w in 2 3 4 6; do specs="$specs $v:P264AMD_INTRA_WAVES=$w"; done; done
bash scratch/variants_run.sh "$specs" | tee gpurun_out/r3_isweep.txt
