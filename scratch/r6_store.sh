#!/bin/bash
# round 6: k_mc's store policy again on the final kernels (aux bits of buffer_store on gfx950: 1 = sc0, 2 = nt, 16 = sc1; shipped: 2 for
# the 16-byte rows of macroblock items, 0 for the 8-byte half rows of quadrant items).  Headline only (golden check must hold).
for i in 1 2 3 4; do
  for which in ${LIBS:-tree st0 st1 st3 st16 st18 st19 st2nt}; do
    if [ $which = tree ]; then unset P264AMD_LIB; else export P264AMD_LIB=$PWD/scratch/lib_$which.so; fi
    python3 bench.py --no-extras --no-cpu-baseline --no-live-counters --steps 20 --warmup 3 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.readline()); k=b['kernels']
print('$which headline', b['value'], 'ms/step', b['ms_per_step'], {n:k[n]['avg_ms'] for n in k}, b['golden_check']['checked'])"
  done
done
