import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from p264decoder_amd import HipReconstructor, Parser, _native
from tests import oracle_bind
lib = _native.load(); ora = oracle_bind.load()
data = open('tests/golden/f26.264','rb').read()
parser = Parser(lib=lib); pics = parser.parse_stream(data, limit=3)
mb_w, mb_h = pics[0].mb_w, pics[0].mb_h
store = oracle_bind.FrameStore(mb_w, mb_h, parser.slots)
hip = HipReconstructor(mb_w, mb_h, 1, parser.slots, 1, lib=lib)
for i,p in enumerate(pics):
    p.desc.deblock = 0
    want = oracle_bind.reconstruct(ora, store, p, deblock=False)
    hip.submit(0, p); got = hip.read_frame(0, p.desc.dst_slot)
    # keep GPU and oracle in sync for the next picture: overwrite GPU frame with oracle's
    d = got[0] != want[0]
    print('pic', i, 'luma diffs', d.sum(), 'chroma', (got[1]!=want[1]).sum(), (got[2]!=want[2]).sum())
    if d.any():
        rec = p.mb_records(); mv = p.mv.reshape(-1,16,2)
        stats = collections.Counter(); tot = collections.Counter()
        for mby in range(mb_h):
            for mbx in range(mb_w):
                mi = mby*mb_w+mbx
                if rec['mb_type'][mi] <= 2: continue
                for b in range(16):
                    bx, by = b & 3, b >> 2
                    fx, fy = mv[mi,b,0] & 3, mv[mi,b,1] & 3
                    blk = d[mby*16+by*4:mby*16+by*4+4, mbx*16+bx*4:mbx*16+bx*4+4]
                    tot[(fx,fy)] += 1
                    if blk.any(): stats[(fx,fy)] += 1
        for k in sorted(tot): print(' phase', k, 'bad blocks', stats[k], '/', tot[k])
        ys,xs = np.nonzero(d); y,x = ys[0],xs[0]
        print(' first', y, x, 'got', got[0][y, x:x+8], 'want', want[0][y, x:x+8])
    hip.write_frame(0, p.desc.dst_slot, *want)
