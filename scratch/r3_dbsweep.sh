#!/bin/bash
for cfg in "default" "3 1" "2 2" "2 1" "1 4" "1 2"; do
  set -- $cfg
  if [ "$1" = "default" ]; then unset P264AMD_DEBLOCK_RB_LOG2 P264AMD_DEBLOCK_PICS_PER_WG; else export P264AMD_DEBLOCK_RB_LOG2=$1 P264AMD_DEBLOCK_PICS_PER_WG=$2; fi
  echo -n "rb_log2/pics_per_wg=$cfg: "
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], {k:v['avg_ms'] for k,v in d['kernels'].items()})"
done
