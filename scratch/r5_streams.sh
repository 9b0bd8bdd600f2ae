#!/bin/bash
# round 5: stage times by streams per GPU (pictures per launch); per-picture microseconds beside them
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for s in ${@:-256 512 768 1024 1280 1536 1792 2048 3072 3840 4096}; do echo -n "streams $s: "; python bench.py --steps 8 --warmup 2 --streams $s --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; n=$s
print(round(d['value']), d['ms_per_step'], 'inter', k['inter']['avg_ms'], 'intra', k['intra']['avg_ms'], 'deblock', k['deblock']['avg_ms'], '| us/picture: inter %.3f intra %.3f deblock %.3f' % (1e3*k['inter']['avg_ms']/n, 1e3*k['intra']['avg_ms']/n, 1e3*k['deblock']['avg_ms']/n), d['launch'])"; done 2>&1 | tee gpurun_out/r5_streams.log
