#!/bin/bash
# round 6 A/B on one box: the work-list sort with the classification kept as three words per macroblock (and B pictures' arrays read once)
# libraries: scratch/lib_a.so (before), scratch/lib_w4.so (packed, k_mc_sort at 4 waves per SIMD), the tree's (k_mc_sort at 8: two workgroups per CU)
for i in 1 2; do
  for which in a w4 tree; do
    if [ $which = tree ]; then unset P264AMD_LIB; else export P264AMD_LIB=$PWD/scratch/lib_$which.so; fi
    python3 bench.py --no-extras --no-cpu-baseline --no-live-counters --steps 20 --warmup 3 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.readline()); k=b['kernels']
print('$which headline', b['value'], 'ms/step', b['ms_per_step'], {n:k[n]['avg_ms'] for n in k}, b['golden_check']['checked'])"
    python3 scratch/r6_ahead.py 2>&1 | sed "s/^/$which /"
  done
done
