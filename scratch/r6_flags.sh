#!/bin/bash
# round 6: compiler scheduling options (scratch/variant.sh <name> -mllvm ...) against the tree's library, headline + batch_256, one box
for i in 1 2 3; do
  for which in ${LIBS:-tree}; do
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
