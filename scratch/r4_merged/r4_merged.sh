#!/bin/bash
# round 4: luma and chroma of a picture's intra macroblocks in ONE workgroup, one after the other (P264AMD_INTRA_MERGED=1: the band
# walk's book-keeping once instead of twice) against two workgroups side by side
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in 0 1; do
  P264AMD_INTRA_MERGED=$v python bench.py --no-cpu-baseline --steps 30 > gpurun_out/mg.json 2>/dev/null
  python - <<PY
import json
b=json.load(open("gpurun_out/mg.json"))
k={k:v["avg_ms"] for k,v in b["kernels"].items()}
e=b["extras"]
print("MERGED=$v", b["value"], b["ms_per_step"], k, "cfg2", e["config2_720p_intra_only"]["value"], "cfg3ip", e["config3_1080p_i_plus_p_gop30"]["value"], "cfg4", e["config4_1080p_main_cabac_ipb"]["value"])
PY
done; done
