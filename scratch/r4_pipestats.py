"""round 4: where the end-to-end pipeline's time goes (parse threads vs the submit side): PipeStats of 128 1080p streams."""
import sys, os
sys.path.insert(0, os.getcwd())
from p264decoder_amd import Pipeline, _native
from tests import synth_cases
import bench
lib = _native.load()
distinct = [open(synth_cases.generate(bench.synth_args(24, 1000 + g)), "rb").read() for g in range(4)]
import sys as _s
combos = [(int(t), 0) for t in _s.argv[1].split(',')] if len(_s.argv) > 1 else [(16, 0), (16, 0), (16, -1), (32, 0)]
for threads, dev in combos:
    pipe = Pipeline([distinct[i % 4] for i in range(128)], threads=threads, device=dev, lib=lib)
    st = pipe.run(); pipe.close()
    print(threads, dev, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()}, "fps", round(st["pictures"] / st["seconds"], 1), flush=True)
