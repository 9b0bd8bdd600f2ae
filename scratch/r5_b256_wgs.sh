#!/bin/bash
# round 5: the MC stage at 256 pictures per launch by k_mc workgroups per picture (P264AMD_MC_WGS_PER_PIC; default there: 192)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in ${@:-0 48 64 96 128 256}; do
  echo -n "wgs $w: "
  P264AMD_MC_WGS_PER_PIC=$w python bench.py --steps 20 --warmup 3 --streams 256 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print(round(d['value']), d['ms_per_step'], 'inter', k['inter']['avg_ms'], 'intra', k['intra']['avg_ms'], 'deblock', k['deblock']['avg_ms'], d['launch']['mc_wgs_per_picture'])"
done 2>&1 | tee gpurun_out/r5_b256_wgs.log
