import sys, time, os
sys.path.insert(0, os.getcwd())
from p264decoder_amd import HipReconstructor, Parser
from tests import synth_cases
name = "main_1080p_cabac_ipb"
data = open(synth_cases.generate(synth_cases.ORACLE_CASES[name]), "rb").read()
parser = Parser(quiet=True)
pics = parser.parse_stream(data)
T = len(pics); S = 1024
hip = HipReconstructor(pics[0].mb_w, pics[0].mb_h, n_streams=S, slots=parser.slots, max_pictures=S * T)
hip.upload(0, pics)
for s in range(1, S):
    for t in range(T): hip.clone_picture(s * T + t, t)
hip.sync()
streams = list(range(S))
for t in range(T): hip.reconstruct([s * T + t for s in streams], streams)
hip.sync()
hip.timing_enable(True)
for t in range(T):
    hip.timing_reset()
    hip.reconstruct([s * T + t for s in streams], streams); hip.sync()
    tm = hip.timing_read()
    print("picture %2d type %d:" % (t, pics[t].desc.slice_type), {k: round(v[0] / max(v[1], 1), 3) for k, v in tm.items()}, flush=True)
hip.close()
