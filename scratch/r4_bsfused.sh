#!/bin/bash
# round 4: the loop filter's edge-info pass as extra workgroups of the k_intra_sparse launch (P264AMD_BS_FUSED = workgroups per
# picture, 0 = its own launch as before)
cd $GRAFT_REPO_ROOT
for v in 0 1 2 4 8; do
  P264AMD_BS_FUSED=$v python bench.py --no-cpu-baseline --no-extras > gpurun_out/bsf_$v.json 2>/dev/null
  python - <<PY
import json
b=json.load(open("gpurun_out/bsf_$v.json"))
print("BS_FUSED=$v", b["value"], b["ms_per_step"], {k:v["avg_ms"] for k,v in b["kernels"].items()}, b["golden_check"]["checked"])
PY
done
