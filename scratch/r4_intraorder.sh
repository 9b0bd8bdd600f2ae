#!/bin/bash
# round 4: dispatch order of the k_intra_sparse launch's workgroups: 0 = a picture's roles side by side (luma, chroma, edge info of
# picture 0, then of picture 1 ...), 1 = role-major (all luma, all chroma, all edge info), 2 = role-major with the edge info first
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in 0 1 2; do
  P264AMD_INTRA_ORDER=$v python bench.py --no-cpu-baseline --no-extras --steps 30 > gpurun_out/or.json 2>/dev/null
  python - <<PY
import json
b=json.load(open("gpurun_out/or.json"))
k={k:v["avg_ms"] for k,v in b["kernels"].items()}
print("ORDER=$v", b["value"], b["ms_per_step"], k, round(k["intra"]+k["deblock"],3), b["golden_check"]["checked"])
PY
done; done
