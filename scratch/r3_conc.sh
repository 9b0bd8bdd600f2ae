#!/bin/bash
for c in 0 1 0 1; do echo -n "concurrent=$c: "; P264AMD_CONCURRENT=$c python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items()})"; done
