#!/bin/bash
# round 6 experiment: which band of a picture that has a CU to itself runs on which SIMD (wavefront w sits on SIMD w % 4; nine bands)
for i in 1 2 3; do
  for perm in 0 1 2 3 4; do
    P264AMD_DB_PERM=$perm python3 bench.py --only-batch-256 --steps 20 --warmup 3 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.readline())['extras']['batch_256']
print('perm $perm batch_256', b['value'], b['ms_per_step'], {n:b['stages'][n]['avg_ms'] for n in b['stages']}, b['last_picture_matches_reference'])"
  done
done
