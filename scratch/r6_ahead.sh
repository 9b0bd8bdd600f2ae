#!/bin/bash
# round 6 A/B on one box: P264AMD_SORT_AHEAD=0 (the sort in front of k_mc on the context's stream) against 1 (side stream, ahead)
for i in 1 2; do
  for a in 0 1; do
    export P264AMD_SORT_AHEAD=$a
    python3 bench.py --no-extras --no-cpu-baseline --no-live-counters --steps 20 --warmup 3 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.readline()); k=b['kernels']
print('ahead=$a headline', b['value'], 'ms/step', b['ms_per_step'], {n:k[n]['avg_ms'] for n in k}, b['golden_check']['checked'], b['launch'].get('sort_ahead'))"
    python3 bench.py --only-batch-256 --steps 20 --warmup 3 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.readline())['extras']['batch_256']
print('ahead=$a batch_256', b['value'], b['ms_per_step'], {n:b['stages'][n]['avg_ms'] for n in b['stages']}, b['last_picture_matches_reference'])"
  done
done
for a in 0 1; do P264AMD_SORT_AHEAD=$a python3 scratch/r6_ahead.py; done
