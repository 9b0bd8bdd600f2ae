#!/bin/bash
# round 6: what a k_deblock iteration waits for where a picture has a CU to itself (256 pictures per launch, one picture per launch):
# timing builds without the per-iteration store drain, without the band synchronisation, without the horizontal pass
cd $GRAFT_REPO_ROOT
for v in mbase dnovm dnosync dnoh; do
  echo "== $v"
  P264AMD_TIMING_BUILD_OK=1 P264AMD_LIB=$PWD/scratch/lib_$v.so python bench.py --only-batch-256 --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.readline())['extras']['batch_256']; print(b['value'], {k:v['avg_ms'] for k,v in b['stages'].items()}, b['last_picture_matches_reference'])"
  P264AMD_TIMING_BUILD_OK=1 P264AMD_LIB=$PWD/scratch/lib_$v.so python scratch/r6_single.py 2>&1 | grep submit
done
