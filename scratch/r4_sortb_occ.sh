#!/bin/bash
# round 4: k_mc_sort_b at 4 (78 registers, one workgroup per CU) / 8 wavefronts per SIMD (64 registers, 9 spilled, two workgroups per CU)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for w in sb4 sb8; do
  P264AMD_LIB=$GRAFT_REPO_ROOT/scratch/lib_$w.so python bench.py --no-cpu-baseline --steps 5 > gpurun_out/sb.json 2>/dev/null
  python - <<PY
import json
b=json.load(open("gpurun_out/sb.json"))
e=b["extras"]["config4_1080p_main_cabac_ipb"]
print("$w", e["value"], {k:v["inter"] for k,v in e["stage_ms_per_launch"].items()})
PY
done; done
