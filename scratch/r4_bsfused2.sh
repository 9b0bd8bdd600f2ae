#!/bin/bash
# A/B with repetitions: edge info in its own launch (0) against one extra workgroup per picture in the k_intra_sparse launch (1)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in 0 1; do
  P264AMD_BS_FUSED=$v python bench.py --no-cpu-baseline --no-extras --steps 40 > gpurun_out/bsf_$v.json 2>/dev/null
  python - <<PY
import json
b=json.load(open("gpurun_out/bsf_$v.json"))
k={k:v["avg_ms"] for k,v in b["kernels"].items()}
print("BS_FUSED=$v", b["value"], b["ms_per_step"], k, round(k["intra"]+k["deblock"],3))
PY
done; done
