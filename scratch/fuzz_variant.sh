#!/bin/bash
for v in $1; do echo -n "$v: "; P264AMD_LIB=$GRAFT_REPO_ROOT/scratch/lib_$v.so python -m pytest tests/test_gpu_seam_fuzz.py -x -q -m gpu 2>&1 | tail -1; done
