#!/bin/bash
# round 4: config 4 (B launches) with two builds of the library, same box
cd $GRAFT_REPO_ROOT
for v in old new old new; do echo "== $v"; P264AMD_LIB=$GRAFT_REPO_ROOT/scratch/lib_$v.so python profiles/cfg_run.py config4 2>/dev/null | tail -3; done 2>&1 | tee gpurun_out/r4_cfg4ab.log
