"""Debug helper: decode a synthetic case on the GPU and the oracle, print where they differ."""
import sys
import numpy as np
from p264decoder_amd import HipReconstructor, Parser, _native
from tests import oracle_bind, synth_cases
lib = _native.load(); oracle = oracle_bind.load()
case = sys.argv[1] if len(sys.argv) > 1 else "cif_ip"
parser = Parser(quiet=True, lib=lib)
pics = parser.parse_stream(synth_cases.stream_bytes(case) if case in synth_cases.CASES else open(synth_cases.generate(case), "rb").read())[:4]
mb_w, mb_h = pics[0].mb_w, pics[0].mb_h
store = oracle_bind.FrameStore(mb_w, mb_h, parser.slots)
hip = HipReconstructor(mb_w, mb_h, n_streams=1, slots=parser.slots, max_pictures=1, lib=lib)
for i, p in enumerate(pics):
    want = oracle_bind.reconstruct(oracle, store, p, deblock=False)
    save = p.desc.deblock; p.desc.deblock = 0
    hip.submit(0, p)
    got = hip.read_frame(0, p.desc.dst_slot)
    p.desc.deblock = save
    rec = p.mb_records()
    for plane, (a, b) in enumerate(zip(got, want)):
        if not np.array_equal(a, b):
            s = 16 if plane == 0 else 8
            ys, xs = np.nonzero(a != b)
            mbs = sorted(set(((ys // s) * mb_w + xs // s).tolist()))
            print("picture %d plane %d: %d samples differ in %d MBs" % (i, plane, len(ys), len(mbs)))
            for m in mbs[:12]:
                mv = p.mv.reshape(-1, 16, 2)[m]
                sel = (ys // s) * mb_w + xs // s == m
                print("  MB %d (%d,%d) type %d cbp %#x mask %#x mv0 %s uniform %s  rows %s cols %s" % (m, m % mb_w, m // mb_w, rec["mb_type"][m], rec["cbp"][m], rec["coef_mask"][m],
                      mv[0].tolist(), bool((mv == mv[0]).all()), sorted(set((ys[sel] % s).tolist())), sorted(set((xs[sel] % s).tolist()))))
    # feed the oracle's result back so that later pictures are judged on their own
    hip.write_frame(0, p.desc.dst_slot, *want)
    if p.desc.deblock:
        want2 = oracle_bind.reconstruct(oracle, store, p)    # redo with the loop filter for the next reference
        hip.write_frame(0, p.desc.dst_slot, *want2)
