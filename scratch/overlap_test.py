"""Does running two half-batches on two HIP streams (two contexts) overlap the issue-bound and the latency-bound kernels?"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from p264decoder_amd import HipReconstructor, Parser, _native
from tests import synth_cases
lib = _native.load()
NCTX = int(sys.argv[1]) if len(sys.argv) > 1 else 2
S_TOTAL = int(sys.argv[2]) if len(sys.argv) > 2 else 512
K, Wm = 10, 2
T = 1 + Wm + K
S = S_TOTAL // NCTX
parsed = []
for g in range(bench.DISTINCT):
    path = synth_cases.generate(bench.synth_args(T, 1000 + g))
    parsed.append(Parser(quiet=True, lib=lib).parse_stream(open(path, "rb").read()))
ctxs = []
for c in range(NCTX):
    hip = HipReconstructor(bench.MB_W, bench.MB_H, n_streams=S, slots=2, max_pictures=S * T, device=0, lib=lib)
    for s in range(min(S, bench.DISTINCT)):
        hip.upload(s * T, parsed[s])
    for s in range(bench.DISTINCT, S):
        for t in range(T):
            hip.clone_picture(s * T + t, (s % bench.DISTINCT) * T + t)
    hip.sync()
    ctxs.append(hip)
streams = list(range(S))
def step(t):
    for hip in ctxs:
        hip.reconstruct([s * T + t for s in streams], streams)
for t in range(1 + Wm):
    step(t)
for hip in ctxs: hip.sync()
t0 = time.perf_counter()
for t in range(1 + Wm, T):
    step(t)
for hip in ctxs: hip.sync()
dt = time.perf_counter() - t0
print("contexts %d x %d streams: %.3f ms/step, %.0f frames/s" % (NCTX, S, dt / K * 1e3, S_TOTAL * K / dt))
