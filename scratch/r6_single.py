#!/usr/bin/env python3
"""round 6: where a single 1080p picture's time goes on the drop-in road (one stream, one picture per launch)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from p264decoder_amd import HipReconstructor, Parser, _native
from tests import synth_cases
lib = _native.load()
data = synth_cases.stream_bytes("cfg3_1080p_ip")
t0 = time.perf_counter()
pics = Parser(quiet=True, lib=lib).parse_stream(data)
t1 = time.perf_counter()
print("parse (python wrapper incl. copies) %.2f ms per picture" % ((t1 - t0) / len(pics) * 1e3))
hip = HipReconstructor(120, 68, n_streams=1, slots=2, max_pictures=1, lib=lib)
for p in pics[:3]:
    hip.submit(0, p); hip.sync()
hip.timing_enable(True); hip.timing_reset()
lat = []
for p in pics[3:33]:
    a = time.perf_counter(); hip.submit(0, p); hip.sync(); lat.append(time.perf_counter() - a)
tm = hip.timing_read()
print("submit+sync %.3f ms median; stages:" % (np.median(lat) * 1e3), {k: round(v[0] / max(v[1], 1), 4) for k, v in tm.items()})
a = time.perf_counter()
for p in pics[3:33]:
    hip.read_frame(0, p.desc.dst_slot)
print("read_frame %.3f ms" % ((time.perf_counter() - a) / 30 * 1e3))
hip.close()
