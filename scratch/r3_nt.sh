#!/bin/bash
cp scratch/bench_nt.py ./bench_nt.py
for i in 1 2 3; do
python bench_nt.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('timing on ', d['ms_per_step'])"
NOTIMING=1 python bench_nt.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1
done
rm -f bench_nt.py
