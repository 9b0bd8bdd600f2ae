import sys, time, os
sys.path.insert(0, os.getcwd())
from p264decoder_amd import HipReconstructor, Parser
from tests import synth_cases
pics = Parser(quiet=True).parse_stream(synth_cases.stream_bytes("cfg3_1080p_allp"))[:6]
T = len(pics); S = 1024
hip = HipReconstructor(120, 68, n_streams=S, slots=2, max_pictures=S * T)
hip.upload(0, pics)
for s in range(1, S):
    for t in range(T): hip.clone_picture(s * T + t, t)
hip.sync()
streams = list(range(S))
hip.reconstruct([s * T for s in streams], streams); hip.sync()
hip.timing_enable(True); hip.timing_reset()
for rep in range(2):
    for t in range(1, T): hip.reconstruct([s * T + t for s in streams], streams)
hip.sync()
tm = hip.timing_read()
print("P launches: intra %.3f ms" % (tm["intra"][0] / max(tm["intra"][1], 1)), flush=True)
hip.close()
