#!/bin/bash
# round 4: frames/s by streams per GPU (pictures per launch)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for s in 2048 3072 4096; do echo -n "streams $s: "; python bench.py --steps 8 --warmup 2 --streams $s --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print(round(d['value']), d['ms_per_step'], 'inter', k['inter']['avg_ms'], 'intra', k['intra']['avg_ms'], 'deblock', k['deblock']['avg_ms'])"; done 2>&1 | tee gpurun_out/r4_streams.log
