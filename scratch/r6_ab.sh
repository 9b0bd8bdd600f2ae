#!/bin/bash
# round 6 A/B on one box: scratch/lib_a.so (before) against the tree's library (after); bench.py's headline with its stage times
# usage: bash scratch/r6_ab.sh [runs] [extra bench args]
runs=${1:-2}; shift
for i in $(seq $runs); do
  for which in a b; do
    if [ $which = a ]; then export P264AMD_LIB=$PWD/scratch/lib_a.so; else unset P264AMD_LIB; fi
    python3 bench.py --no-extras --no-cpu-baseline --no-live-counters --steps 20 --warmup 3 "$@" 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.readline()); k=b['kernels']
print('$which', b['value'], 'ms/step', b['ms_per_step'], {n:k[n]['avg_ms'] for n in k}, b['golden_check']['checked'])"
  done
done
