import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from p264decoder_amd import HipReconstructor, _native
from tests import oracle_bind, seam_fuzz, test_gpu_seam_fuzz as T
lib = _native.load(); oracle = oracle_bind.load()
name = sys.argv[1] if len(sys.argv) > 1 else "typical"
cfg = [c for c in T.CONFIGS if c[0] == name][0]
_, mb_w, mb_h, n_pics, kw = cfg
rng = np.random.default_rng(sum(map(ord, name)) * 7919)
slots = kw["slots"]
store = oracle_bind.FrameStore(mb_w, mb_h, slots)
hip = HipReconstructor(mb_w, mb_h, n_streams=1, slots=slots, max_pictures=1, lib=lib)
for s in range(slots):
    f = seam_fuzz.random_frame(rng, mb_w, mb_h, "smooth" if "smooth" in name else "noise")
    for dst, src in zip(store[s], f): dst[:] = src
    hip.write_frame(0, s, *f)
nodb = len(sys.argv) > 2
for i in range(n_pics):
    pic = seam_fuzz.make_picture(rng, mb_w, mb_h, p_picture=(i != 2), dst_slot=i % slots, **kw)
    if nodb: pic.desc.deblock = 0
    want = oracle_bind.reconstruct(oracle, store, pic)
    hip.submit(0, pic)
    got = hip.read_frame(0, pic.desc.dst_slot)
    bad = False
    for plane, (a, b) in enumerate(zip(got, want)):
        if not np.array_equal(a, b):
            bad = True
            sz = 16 if plane == 0 else 8
            d = a != b
            mbs = sorted({(y // sz, x // sz) for y, x in zip(*np.nonzero(d))})
            print("picture %d plane %d: %d samples in MBs %s" % (i, plane, d.sum(), mbs[:12]))
            for (my, mx) in mbs[:3]:
                r = pic.rec[my * mb_w + mx]
                print("  MB (%d,%d) type %d qp %d cbp %#x modes %#x mask %#x avail %d" % (mx, my, r["mb_type"], r["qp"], r["cbp"], r["intra_modes"], r["coef_mask"], r["avail"]))
                if r["mb_type"] == 0: print("   i4modes", pic.i4modes.reshape(-1, 16)[my * mb_w + mx].tolist())
                print("   got\n", a[my * sz:(my + 1) * sz, mx * sz:(mx + 1) * sz]); print("   want\n", b[my * sz:(my + 1) * sz, mx * sz:(mx + 1) * sz])
                if plane > 0:
                    Y0, X0 = my * 8, mx * 8
                    top = b[Y0 - 1, X0 - 1:X0 + 8].astype(int); left = b[Y0 - 1:Y0 + 8, X0 - 1].astype(int)   # index 0 = corner
                    H = sum((i + 1) * (top[1 + 4 + i] - top[1 + 2 - i]) for i in range(4)); V = sum((i + 1) * (left[1 + 4 + i] - left[1 + 2 - i]) for i in range(4))
                    A = 16 * (left[8] + top[8]); B = (17 * H + 16) >> 5; Cc = (17 * V + 16) >> 5
                    print("   plane H %d V %d a %d b %d c %d top %s left %s" % (H, V, A, B, Cc, top.tolist(), left.tolist()))
                    print(np.array([[(A - 3 * B - 3 * Cc + 16 + Cc * y + B * x) >> 5 for x in range(8)] for y in range(8)]))
                    mask = int(r["coef_mask"]); ci = int(r["coef_index"]); co = pic.coefs.reshape(-1, 16)
                    def slot(blk): return ((mask >> 24) & 1) + ((mask >> 25) & 1) + bin(mask & ((1 << blk) - 1) & 0xffffff).count("1")
                    pl = plane - 1
                    dcs = co[ci + ((mask >> 24) & 1)][pl * 4:pl * 4 + 4].astype(int) if (mask >> 25) & 1 else np.zeros(4, int)
                    print("   chroma DC levels", dcs.tolist(), "chroma_qp_offset", pic.desc.chroma_qp_offset)
                    for j in range(4):
                        blk = 16 + 4 * pl + j
                        print("   block", blk, "coded", (mask >> blk) & 1, co[ci + slot(blk)].tolist() if (mask >> blk) & 1 else None)
    if bad: break
else: print("all match")
hip.close()
# plane prediction of the first bad MB's plane from the (matching) neighbours
