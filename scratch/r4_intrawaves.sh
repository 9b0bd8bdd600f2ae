#!/bin/bash
# round 4: wavefronts per workgroup of the k_intra_sparse launch (with the edge-info role inside) at 2048 pictures per launch
cd $GRAFT_REPO_ROOT
for v in 2 4 8 16; do for f in 1 2; do
  P264AMD_INTRA_WAVES=$v P264AMD_BS_FUSED=$f python bench.py --no-cpu-baseline --no-extras --steps 30 > gpurun_out/iw.json 2>/dev/null
  python - <<PY
import json
b=json.load(open("gpurun_out/iw.json"))
k={k:v["avg_ms"] for k,v in b["kernels"].items()}
print("INTRA_WAVES=$v BS_FUSED=$f", b["value"], b["ms_per_step"], k, round(k["intra"]+k["deblock"],3))
PY
done; done
