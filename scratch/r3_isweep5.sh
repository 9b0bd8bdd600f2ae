#!/bin/bash
for lib in $1; do for iw in 2 4 8; do echo -n "$lib waves/wg=$iw: "; P264AMD_INTRA_WAVES=$iw P264AMD_LIB=$GRAFT_REPO_ROOT/scratch/lib_$lib.so python scratch/r3_p.py 2>/dev/null; done; done
