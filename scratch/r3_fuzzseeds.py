"""Extra seeds through the seam fuzz generator (tests/seam_fuzz.py): HIP against the oracle, picture by picture."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from p264decoder_amd import HipReconstructor, _native
from tests import oracle_bind, seam_fuzz, test_gpu_seam_fuzz as T
lib = _native.load(); oracle = oracle_bind.load()
bad = 0; n = 0
for seed in range(1, int(sys.argv[1]) + 1 if len(sys.argv) > 1 else 13):
    for cfg in T.CONFIGS:
        name, mb_w, mb_h, n_pics, kw = cfg
        rng = np.random.default_rng(seed * 1000003 + sum(map(ord, name)))
        slots = kw["slots"]
        store = oracle_bind.FrameStore(mb_w, mb_h, slots)
        hip = HipReconstructor(mb_w, mb_h, n_streams=1, slots=slots, max_pictures=1, lib=lib)
        for s in range(slots):
            f = seam_fuzz.random_frame(rng, mb_w, mb_h, "smooth" if "smooth" in name else "noise")
            for dst, src in zip(store[s], f): dst[:] = src
            hip.write_frame(0, s, *f)
        for i in range(n_pics):
            pic = seam_fuzz.make_picture(rng, mb_w, mb_h, p_picture=(i != 2), dst_slot=i % slots, **kw)
            want = oracle_bind.reconstruct(oracle, store, pic)
            hip.submit(0, pic)
            got = hip.read_frame(0, pic.desc.dst_slot)
            n += 1
            if not all(np.array_equal(a, b) for a, b in zip(got, want)):
                bad += 1; print("MISMATCH seed %d config %s picture %d" % (seed, name, i), flush=True)
                break
        hip.close()
print("%d pictures, %d mismatches" % (n, bad))
