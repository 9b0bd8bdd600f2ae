#!/bin/bash
# round 4: wavefronts per workgroup of the dense k_intra launch (config 2: 1024 720p I pictures per launch; config 3 I + P), 64-register build
cd $GRAFT_REPO_ROOT
for w in 2 4 8 16; do
  P264AMD_INTRA_WAVES=$w python bench.py --no-cpu-baseline --steps 5 > gpurun_out/iw2.json 2>/dev/null
  python - <<PY
import json
b=json.load(open("gpurun_out/iw2.json"))
e=b["extras"]
print("INTRA_WAVES=$w", b["value"], "cfg2", e["config2_720p_intra_only"]["value"], "cfg3ip", e["config3_1080p_i_plus_p_gop30"]["value"], "cfg4", e["config4_1080p_main_cabac_ipb"]["value"])
PY
done
