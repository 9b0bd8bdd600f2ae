#!/bin/bash
for cfg in "3 1" "2 1" "2 2" "1 1"; do set -- $cfg
  echo -n "S=${S:-256} RB_LOG2=$1 PICS=$2: "
  P264AMD_DEBLOCK_RB_LOG2=$1 P264AMD_DEBLOCK_PICS_PER_WG=$2 python bench.py --steps 8 --warmup 2 --streams ${S:-256} --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print(round(d['value']), d['ms_per_step'], 'inter', k['inter']['avg_ms'], 'intra', k['intra']['avg_ms'], 'deblock', k['deblock']['avg_ms'])"
done
