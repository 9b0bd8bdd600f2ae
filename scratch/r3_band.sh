#!/bin/bash
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernels']['inter']['avg_ms'], end=' ')"; }
for b in 4 5; do echo -n "band_log2=$b: " ; done; echo
for rep in 1 2 3 4 5 6; do for b in 4 5; do export P264AMD_MC_BAND_LOG2=$b; run; done; echo; done
