#!/bin/bash
# round 6 A/B on one box: the work-list sort with one LDS atomic per distinct slot of a wavefront (wave_slot_add, kernel_mc.h)
# libraries: scratch/lib_a.so (before), the tree's (after)
for i in 1 2 3; do
  for which in a tree; do
    if [ $which = tree ]; then unset P264AMD_LIB; else export P264AMD_LIB=$PWD/scratch/lib_$which.so; fi
    python3 bench.py --no-extras --no-cpu-baseline --no-live-counters --steps 20 --warmup 3 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.readline()); k=b['kernels']
print('$which headline', b['value'], 'ms/step', b['ms_per_step'], {n:k[n]['avg_ms'] for n in k}, b['golden_check']['checked'])"
    python3 bench.py --only-batch-256 --steps 20 --warmup 3 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.readline())['extras']['batch_256']
print('$which batch_256', b['value'], b['ms_per_step'], {n:b['stages'][n]['avg_ms'] for n in b['stages']}, b['last_picture_matches_reference'])"
  done
done
for which in a tree; do
  if [ $which = tree ]; then unset P264AMD_LIB; else export P264AMD_LIB=$PWD/scratch/lib_$which.so; fi
  python3 scratch/r6_ahead.py 2>&1 | sed "s/^/$which /"
done
