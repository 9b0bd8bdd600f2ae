#!/bin/bash
# round 6: k_deblock's split launches (one workgroup of one wavefront per band, small batches) on and off
cd $GRAFT_REPO_ROOT
for sp in 0 1; do
  export P264AMD_DEBLOCK_SPLIT=$sp
  echo "split=$sp"; python scratch/r6_single.py 2>&1 | grep submit
done
unset P264AMD_DEBLOCK_SPLIT
python - <<'PY'
import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
from p264decoder_amd import HipReconstructor, Parser, _native
from tests import synth_cases
lib = _native.load()
pics = Parser(quiet=True, lib=lib).parse_stream(synth_cases.stream_bytes("cfg3_1080p_allp"), limit=12)
for sp in ("0", "1"):
    os.environ["P264AMD_DEBLOCK_SPLIT"] = sp
    for S in (1, 4, 16, 28):
        hip = HipReconstructor(120, 68, n_streams=S, slots=2, max_pictures=S * 12, lib=lib)
        hip.upload(0, pics)
        for s in range(1, S):
            for t in range(12): hip.clone_picture(s * 12 + t, t)
        hip.reconstruct([s * 12 for s in range(S)], list(range(S))); hip.sync()
        hip.timing_enable(True); hip.timing_reset()
        for t in range(1, 12): hip.reconstruct([s * 12 + t for s in range(S)], list(range(S)))
        hip.sync()
        tm = hip.timing_read()
        print("split", sp, "pictures", S, {k: round(v[0] / max(v[1], 1), 4) for k, v in tm.items()}, hip.last_launch()["deblock_wgs"])
        hip.close()
PY
