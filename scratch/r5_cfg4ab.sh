#!/bin/bash
# round 5: config 4 (Main, CABAC, I + P + B) stage times per picture, library variants side by side (P264AMD_LIB)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in ${@:-base cur}; do
  L=""; [ $v != cur ] && L=$GRAFT_REPO_ROOT/scratch/lib_$v.so
  echo "== $v"
  P264AMD_LIB=$L python profiles/cfg_run.py config4 1024 2>/dev/null | python -c "
import sys, ast
tot = {}
n = {}
for l in sys.stdin:
    if not l.startswith('picture'): continue
    st = int(l.split('slice type')[1].split(':')[0]); d = ast.literal_eval(l.split(':', 1)[1].strip())
    for k, v in d.items(): tot[(st, k)] = tot.get((st, k), 0) + v
    n[st] = n.get(st, 0) + 1
names = {0: 'P', 1: 'B', 2: 'I'}
total = 0
for st in sorted(n):
    print(names[st], n[st], 'launches:', {k: round(tot[(st, k)] / n[st], 3) for (s2, k) in tot if s2 == st})
    total += tot[(st, 'reconstruct')]
print('GOP of %d pictures: %.2f ms per 1024 streams -> %.0f frames/s' % (sum(n.values()), total, 1024 * sum(n.values()) / total * 1e3))"
done 2>&1 | tee gpurun_out/r5_cfg4ab.log
