#!/bin/bash
# usage: scratch/variants_run.sh "<name[:ENV=V,ENV=V]> ..." [streams]
for spec in $1; do
  n=${spec%%:*}; envs=""
  if [[ "$spec" == *:* ]]; then envs=$(echo "${spec#*:}" | tr ',' ' '); fi
  echo -n "$spec: "
  env $envs P264AMD_TIMING_BUILD_OK=1 P264AMD_BENCH_NO_GOLDEN=${NOGOLD:-} P264AMD_LIB=$GRAFT_REPO_ROOT/scratch/lib_$n.so python bench.py --steps ${STEPS:-8} --warmup 2 --streams ${2:-1024} --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print(round(d['value']), d['ms_per_step'], 'inter', k['inter']['avg_ms'], 'intra', k['intra']['avg_ms'], 'deblock', k['deblock']['avg_ms'], 'golden', d['golden_check'].get('checked'))"
done
