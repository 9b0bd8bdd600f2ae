#!/bin/bash
# sweep of P264AMD_MC_WGS_PER_PIC: kernel times from bench.py's own event timing
for w in 2 6 12 24 48 96 200 512; do
  echo "W=$w $(P264AMD_MC_WGS_PER_PIC=$w python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:round(v["avg_ms"],3) for k,v in d["kernels"].items()}, d["ms_per_step"])')"
done
