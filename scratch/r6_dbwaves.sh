#!/bin/bash
# round 6: k_deblock's wavefronts per workgroup where a picture has a CU to itself (9 units of 8 rows: 9 wavefronts = 3 + 2 + 2 + 2 per SIMD;
# 8 wavefronts = 2 per SIMD, the ninth (half) band behind band 0 on wavefront 0)
cd $GRAFT_REPO_ROOT
for w in 0 8 6 4; do
  if [ $w = 0 ]; then unset P264AMD_DEBLOCK_WAVES; else export P264AMD_DEBLOCK_WAVES=$w; fi
  echo "waves=$w"
  python bench.py --only-batch-256 --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.readline())['extras']['batch_256']; print(b['value'], {k:v['avg_ms'] for k,v in b['stages'].items()}, b['launch']['deblock_waves'], b['last_picture_matches_reference'])"
  python scratch/r6_single.py 2>&1 | grep submit
done
