#!/bin/bash
# round 4: does the range of the motion vectors (DRAM / L2 locality of the reference reads) move k_mc?  Same phase statistics
# (vector mod 4 uniform), vectors within +-1, +-16 (the bench's), +-64, +-256 pixels
cd $GRAFT_REPO_ROOT
for mv in 4 64 256 1024; do
  P264AMD_BENCH_SYNTH_EXTRA="--mvmax $mv" python bench.py --no-cpu-baseline --no-extras --steps 20 > gpurun_out/mv.json 2>/dev/null
  python - <<PY
import json
b=json.load(open("gpurun_out/mv.json"))
k={k:v["avg_ms"] for k,v in b["kernels"].items()}
print("mvmax $mv", b["value"], b["ms_per_step"], k)
PY
done
