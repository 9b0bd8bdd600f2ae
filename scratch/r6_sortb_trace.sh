#!/bin/bash
# round 6: k_mc_sort_b's own duration (kernel trace of the config-4 run) with the tree's library and with scratch/lib_keep.so
# (r6_sortkeep.patch: both passes' classifications kept packed, the arrays read once)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for which in tree keep; do
  if [ $which = tree ]; then unset P264AMD_LIB; else export P264AMD_LIB=$GRAFT_REPO_ROOT/scratch/lib_$which.so; fi
  rm -rf gpurun_out/sortb_$which
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sortb_$which -- python3 profiles/cfg_run.py config4 > gpurun_out/sortb_$which.log 2>&1
  echo "== $which"; cat gpurun_out/sortb_$which/*/*kernel_stats.csv | grep "k_mc_sort\|k_mc(\|k_mc_second" | cut -d, -f1-4 | cut -c1-120
done
