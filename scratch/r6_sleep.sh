#!/bin/bash
# round 6: k_deblock's poll interval of a band waiting for the band above (s_sleep 16 -> 2) where a picture has a CU to itself
cd $GRAFT_REPO_ROOT
for v in mbase dsleep2 mbase dsleep2; do
  echo -n "$v: "
  P264AMD_LIB=$PWD/scratch/lib_$v.so python bench.py --only-batch-256 --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.readline())['extras']['batch_256']; print(b['value'], {k:v['avg_ms'] for k,v in b['stages'].items()}, b['last_picture_matches_reference'])"
done
