#!/bin/bash
# round 4: the dense k_intra build at 5 / 6 / 7 / 8 wavefronts per SIMD (-DINTRA_WAVES_PER_EU: 96 / 80 / 72 / 64 registers; since the band
# walk's state is scalar the kernel needs 75 and does not spill at 64): scratch/variant.sh iw<N> -DINTRA_WAVES_PER_EU=<N>
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for w in 5 6 7 8; do
  P264AMD_LIB=$GRAFT_REPO_ROOT/scratch/lib_iw$w.so python bench.py --no-cpu-baseline --steps 10 > gpurun_out/io.json 2>/dev/null
  python - <<PY
import json
b=json.load(open("gpurun_out/io.json"))
e=b["extras"]
print("waves/EU $w", b["value"], "cfg2", e["config2_720p_intra_only"]["value"], "cfg3ip", e["config3_1080p_i_plus_p_gop30"]["value"], "cfg4", e["config4_1080p_main_cabac_ipb"]["value"])
PY
done; done
