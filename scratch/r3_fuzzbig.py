"""1080p random pictures with a large share of intra macroblocks (P and B): the free lists overflow, the second round and the
band walk get real work; HIP against the oracle."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from p264decoder_amd import HipReconstructor, _native
from tests import oracle_bind, seam_fuzz
lib = _native.load(); oracle = oracle_bind.load()
mb_w, mb_h = 120, 68
bad = 0
for seed, share, b in ((1, 0.3, False), (2, 0.6, False), (3, 0.9, False), (4, 0.45, True), (5, 0.05, True), (6, 0.02, False)):
    rng = np.random.default_rng(777 + seed)
    slots = 3
    store = oracle_bind.FrameStore(mb_w, mb_h, slots)
    hip = HipReconstructor(mb_w, mb_h, n_streams=1, slots=slots, max_pictures=1, lib=lib)
    for s in range(slots):
        f = seam_fuzz.random_frame(rng, mb_w, mb_h, "noise")
        for dst, src in zip(store[s], f): dst[:] = src
        hip.write_frame(0, s, *f)
    for i in range(3):
        pic = seam_fuzz.make_picture(rng, mb_w, mb_h, p_picture=True, dst_slot=i % slots, level_style="small", qp_mode="random", n_ref=2, slots=slots,
                                     intra_share=share, b_picture=b, n_ref_l1=2)
        want = oracle_bind.reconstruct(oracle, store, pic)
        hip.submit(0, pic)
        got = hip.read_frame(0, pic.desc.dst_slot)
        ok = all(np.array_equal(a, b2) for a, b2 in zip(got, want))
        print("intra share %.2f B=%d picture %d: %s" % (share, b, i, "ok" if ok else "MISMATCH"), flush=True)
        bad += not ok
    hip.close()
print("mismatches:", bad)
