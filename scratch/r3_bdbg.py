import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from p264decoder_amd import HipReconstructor, Parser, _native
from tests import oracle_bind, synth_cases
lib = _native.load()
oracle = oracle_bind.load()
args = sys.argv[1] if len(sys.argv) > 1 else "--mbw 12 --mbh 9 --frames 7 --seed 91 --refs 2 --bframes 2 --implicit --d8inf --coded 12 --maxlevel 12"
parser = Parser(quiet=True, lib=lib)
pics = parser.parse_stream(open(synth_cases.generate(args), "rb").read())
mb_w, mb_h = pics[0].mb_w, pics[0].mb_h
store = oracle_bind.FrameStore(mb_w, mb_h, parser.slots)
hip = HipReconstructor(mb_w, mb_h, n_streams=1, slots=parser.slots, max_pictures=1, lib=lib)
for i, p in enumerate(pics):
    want = oracle_bind.reconstruct(oracle, store, p)
    hip.submit(0, p)
    got = hip.read_frame(0, p.desc.dst_slot)
    bad = False
    for plane, (a, b) in enumerate(zip(got, want)):
        if not np.array_equal(a, b):
            bad = True
            sz = 16 if plane == 0 else 8
            d = (a != b)
            mbs = sorted({(y // sz, x // sz) for y, x in zip(*np.nonzero(d))})
            print("picture %d type %d plane %d: %d samples differ in %d MBs" % (i, p.desc.slice_type, plane, d.sum(), len(mbs)))
            for (my, mx) in mbs[:6]:
                mbi = my * mb_w + mx
                r0 = np.ctypeslib.as_array(p.desc.ref_idx, (mb_w * mb_h * 4,))[mbi * 4:mbi * 4 + 4]
                r1 = np.ctypeslib.as_array(p.desc.ref_idx_l1, (mb_w * mb_h * 4,))[mbi * 4:mbi * 4 + 4] if p.desc.slice_type == 1 else None
                mv0 = np.ctypeslib.as_array(p.desc.mv, (mb_w * mb_h * 32,))[mbi * 32:mbi * 32 + 32].reshape(16, 2)
                mv1 = np.ctypeslib.as_array(p.desc.mv_l1, (mb_w * mb_h * 32,))[mbi * 32:mbi * 32 + 32].reshape(16, 2) if p.desc.slice_type == 1 else None
                blk = d[my * sz:(my + 1) * sz, mx * sz:(mx + 1) * sz]
                print("  MB (%d,%d) refs0 %s refs1 %s" % (mx, my, list(r0), None if r1 is None else list(r1)))
                print("   mv0", mv0.tolist()); 
                if mv1 is not None: print("   mv1", mv1.tolist())
                print("   diff map rows:", ["".join("x" if v else "." for v in row) for row in blk])
                print("   got ", a[my * sz:(my) * sz + 2, mx * sz:(mx + 1) * sz].tolist()); print("   want", b[my * sz:(my) * sz + 2, mx * sz:(mx + 1) * sz].tolist())
    if bad: break
else:
    print("all %d pictures match" % len(pics))
hip.close()
