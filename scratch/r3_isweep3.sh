#!/bin/bash
for lib in $1; do
    echo -n "$lib: "
    P264AMD_LIB=$GRAFT_REPO_ROOT/scratch/lib_$lib.so python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('main intra', d['kernels']['intra']['avg_ms'], 'golden', d['golden_check']['checked'], end='  ')"
    P264AMD_LIB=$GRAFT_REPO_ROOT/scratch/lib_$lib.so python scratch/r3_cfg2c.py 2>/dev/null
done
