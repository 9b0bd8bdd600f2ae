#!/bin/bash
# round 3, first GPU look: box CPU share, GPU tests, W sweep of the fused MC launch
out=gpurun_out/r3_first; mkdir -p $out
{ nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; lscpu | grep -i "model name\|^CPU(s)\|thread\|socket\|numa"; free -g | head -2; } > $out/box.txt 2>&1
python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -3 $out/tests.log
for w in 24 48 96 160 256; do
  echo "W=$w $(P264AMD_MC_WGS_PER_PIC=$w python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:round(v["avg_ms"],3) for k,v in d["kernels"].items()}, d["ms_per_step"], d["roofline_mc"]["frac"])')" | tee -a $out/wsweep.txt
done
