#!/bin/bash
# caps on workgroups per CU through unused dynamic LDS, one MC kernel at a time (kernel time = rocprof-free: bench's inter stage)
run() { python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["kernels"]["inter"]["avg_ms"],3))'; }
echo "base $(run)"
for v in YM YQ CM CQ; do
  for pad in 16000 24000 36000 60000; do
    echo "$v pad=$pad $(env P264AMD_MC_LDS_PAD_$v=$pad python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["kernels"]["inter"]["avg_ms"],3))')"
  done
done
