#!/bin/bash
for lib in $1; do echo -n "$lib: "; P264AMD_LIB=$GRAFT_REPO_ROOT/scratch/lib_$lib.so python scratch/r3_p.py 2>/dev/null; done
