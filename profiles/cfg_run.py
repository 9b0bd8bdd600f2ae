#!/usr/bin/env python3
"""One of the non-metric workloads of bench.py's extras as a program of its own, for rocprofv3 (profiles/collect_cfg.sh):
   python3 profiles/cfg_run.py config4 | config2 [streams] [pictures]
config4: BASELINE config 4, 1920x1088 Main profile CABAC I + P + B (tests/synth_cases.py ORACLE_CASES main_1080p_cabac_ipb),
config2: BASELINE config 2, 1280x720 Baseline CAVLC intra only.  Every picture of the stream on `streams` private clones per
launch, inputs resident; two passes, the second one timed per picture with HIP events (printed)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from p264decoder_amd import HipReconstructor, Parser       # noqa: E402
from tests import synth_cases                               # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "config4"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
if which == "config4":
    data = open(synth_cases.generate(synth_cases.ORACLE_CASES["main_1080p_cabac_ipb"]), "rb").read()
else:
    data = synth_cases.stream_bytes("cfg2_720p_intra")
parser = Parser(quiet=True)
pics = parser.parse_stream(data)[:int(sys.argv[3]) if len(sys.argv) > 3 else 13]
T = len(pics)
hip = HipReconstructor(pics[0].mb_w, pics[0].mb_h, n_streams=S, slots=parser.slots, max_pictures=S * T)
hip.upload(0, pics)
for s in range(1, S):
    for t in range(T):
        hip.clone_picture(s * T + t, t)
hip.sync()
streams = list(range(S))
for t in range(T):
    hip.reconstruct([s * T + t for s in streams], streams)
hip.sync()
hip.timing_enable(True)
for t in range(T):
    hip.timing_reset()
    hip.reconstruct([s * T + t for s in streams], streams)
    hip.sync()
    tm = hip.timing_read()
    print("picture %2d slice type %d:" % (t, pics[t].desc.slice_type), {k: round(v[0] / max(v[1], 1), 3) for k, v in tm.items()}, flush=True)
hip.close()
