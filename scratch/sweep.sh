#!/bin/bash
# usage: scratch/sweep.sh "<streams list>" "<rb list>"
for s in $1; do for rb in $2; do
  echo -n "S=$s RB_LOG2=$rb: "
  P264AMD_DEBLOCK_RB_LOG2=$rb python bench.py --steps 8 --warmup 2 --streams $s --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print(round(d['value']), d['ms_per_step'], 'inter', k['inter']['avg_ms'], 'intra', k['intra']['avg_ms'], 'deblock', k['deblock']['avg_ms'])"
done; done
