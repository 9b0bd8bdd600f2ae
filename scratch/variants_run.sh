#!/bin/bash
# usage: scratch/variants_run.sh "<names>" [streams]
for n in $1; do
  echo -n "$n: "
  P264AMD_LIB=$GRAFT_REPO_ROOT/scratch/lib_$n.so python bench.py --steps 8 --warmup 2 --streams ${2:-256} --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print(round(d['value']), d['ms_per_step'], 'inter', k['inter']['avg_ms'], 'intra', k['intra']['avg_ms'], 'deblock', k['deblock']['avg_ms'])"
done
