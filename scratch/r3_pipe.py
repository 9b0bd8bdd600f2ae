# parse-only and end-to-end pipeline rates at several thread counts (run on the GPU box)
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from p264decoder_amd import Pipeline, _native
from tests import synth_cases
import bench
lib = _native.load()
distinct = [open(synth_cases.generate(bench.synth_args(24, 1000 + g)), "rb").read() for g in range(4)]
out = {}
for dev in (-1, 0):
    for th in (1, 8, 16, 32, 64):
        S = max(64, th)
        pipe = Pipeline([distinct[i % 4] for i in range(S)], threads=th, device=dev, lib=lib)
        if dev >= 0:
            pipe.run(max_pictures=2); pipe.close()
            pipe = Pipeline([distinct[i % 4] for i in range(S)], threads=th, device=dev, lib=lib)
        st = pipe.run(); pipe.close()
        key = "%s_%d" % ("parse" if dev < 0 else "e2e", th)
        out[key] = dict(fps=round(st["pictures"] / st["seconds"], 1), per_thread_parse_fps=round(st["pictures"] / st["parse_seconds"], 1), submit_s=round(st["submit_seconds"], 3), s=round(st["seconds"], 3))
        print(key, out[key], flush=True)
