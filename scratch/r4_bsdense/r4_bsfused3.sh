#!/bin/bash
# round 4: the edge-info role in the dense k_intra launch too (I pictures): config 2 and config 3 I + P with P264AMD_BS_FUSED = 0 / 1
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in 0 1; do
  P264AMD_BS_FUSED=$v python bench.py --no-cpu-baseline --steps 10 > gpurun_out/bf.json 2>/dev/null
  python - <<PY
import json
b=json.load(open("gpurun_out/bf.json"))
e=b["extras"]
print("BS_FUSED=$v", b["value"], "cfg2", e["config2_720p_intra_only"]["value"], "cfg3ip", e["config3_1080p_i_plus_p_gop30"]["value"], "cfg4", e["config4_1080p_main_cabac_ipb"]["value"])
PY
done; done
