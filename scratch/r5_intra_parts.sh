#!/bin/bash
# round 5: the intra launch of a P batch taken apart - the edge-info pass inside it (default) or as its own launch (P264AMD_BS_FUSED=0),
# by wavefronts per workgroup of the intra roles.  usage: gpurun -- 'bash scratch/r5_intra_parts.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5_intra_parts
for fused in 1 0; do for w in 2 4 8; do
  echo -n "edge info fused $fused, intra wavefronts $w: "
  P264AMD_BS_FUSED=$fused P264AMD_INTRA_WAVES=$w python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>gpurun_out/r5_intra_parts/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print(round(d['value']), 'inter', k['inter']['avg_ms'], 'intra', k['intra']['avg_ms'], 'deblock', k['deblock']['avg_ms'])" || tail -3 gpurun_out/r5_intra_parts/err.log
done; done 2>&1 | tee gpurun_out/r5_intra_parts/log.txt
