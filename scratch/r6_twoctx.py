#!/usr/bin/env python3
"""round 6: do two half batches on two HIP streams overlap (one half's MC beside the other half's loop filter)?
The metric's all-P workload as G contexts of S / G streams each (every context has its own HIP stream), the contexts'
reconstruct calls issued in turn from one host thread, against one context of S streams.  No library change.
usage: python3 scratch/r6_twoctx.py [S] [K] [stagger]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from p264decoder_amd import HipReconstructor, Parser, _native   # noqa: E402
from tests import synth_cases                                  # noqa: E402

MB_W, MB_H = 120, 68
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
K = int(sys.argv[2]) if len(sys.argv) > 2 else 12
STAGGER = int(sys.argv[3]) if len(sys.argv) > 3 else 0
Wm = 2
T = 1 + Wm + K
lib = _native.load()
pics = Parser(quiet=True, lib=lib).parse_stream(open(synth_cases.generate("cfg3_1080p_allp"), "rb").read(), limit=T)


def run(G):
    n = S // G
    ctxs = []
    for g in range(G):
        hip = HipReconstructor(MB_W, MB_H, n_streams=n, slots=2, max_pictures=n * T, lib=lib)
        hip.upload(0, pics)
        for s in range(1, n):
            for t in range(T):
                hip.clone_picture(s * T + t, t)
        hip.sync()
        ctxs.append(hip)
    streams = list(range(n))
    for t in range(1 + Wm):
        for hip in ctxs:
            hip.reconstruct([s * T + t for s in streams], streams)
    for hip in ctxs:
        hip.sync()
    t0 = time.perf_counter()
    if STAGGER and G > 1:                                     # context g runs `STAGGER` launches ahead of g + 1 ... no: one stage apart by starting late
        pass
    for t in range(1 + Wm, T):
        for hip in ctxs:
            hip.reconstruct([s * T + t for s in streams], streams)
    for hip in ctxs:
        hip.sync()
    dt = time.perf_counter() - t0
    for hip in ctxs:
        hip.close()
    return S * K / dt, dt / K * 1e3


for G in (1, 2, 1, 2, 4):
    fps, ms = run(G)
    print("contexts %d x %d streams: %.0f frames/s, %.3f ms per step of %d pictures" % (G, S // G, fps, ms, S), flush=True)
