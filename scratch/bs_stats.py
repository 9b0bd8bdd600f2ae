"""How many edges of the bench stream have a non-zero boundary strength (tuning aid, CPU only)."""
import sys, numpy as np
sys.path.insert(0, ".")
from tests import synth_cases
from p264decoder_amd import Parser
pics = Parser(quiet=True).parse_stream(synth_cases.stream_bytes("cfg3_1080p_allp"), limit=6)
p = pics[4]
W, H = p.mb_w, p.mb_h
r = p.mb_records().reshape(H, W)
intra = r["mb_type"] <= 2
mask = r["coef_mask"]
mv = p.mv.reshape(H, W, 4, 4, 2).astype(np.int32)          # [mby][mbx][y][x][c]
ref = p.ref_idx.reshape(H, W, 2, 2).astype(np.int32)
# per-4x4-block planes over the whole picture
coded = np.zeros((H * 4, W * 4), bool)
for y in range(4):
    for x in range(4):
        b = (x & 1) | ((y & 1) << 1) | ((x & 2) << 1) | ((y & 2) << 2)
        coded[y::4, x::4] = (mask >> b) & 1
MV = mv.transpose(0, 2, 1, 3, 4).reshape(H * 4, W * 4, 2)
REF = np.repeat(np.repeat(ref.transpose(0, 2, 1, 3).reshape(H * 2, W * 2), 2, 0), 2, 1)
INTRA = np.repeat(np.repeat(intra, 4, 0), 4, 1)
def bs(axis):
    a = lambda A: np.roll(A, 1, axis=axis)
    d = np.abs(MV - a(MV)).max(-1) >= 4
    nz = d | (REF != a(REF)) | coded | a(coded) | INTRA | a(INTRA)
    if axis == 1: nz[:, 0] = False
    else: nz[0, :] = False
    return nz
v, h = bs(1), bs(0)      # [4H][4W]: segment (block) of the edge on its left / top side
print("blocks' vertical-edge segments nonzero: %.3f, horizontal %.3f" % (v.mean(), h.mean()))
# per MB and edge index: any of the 4 segments
for name, nz, ax in (("vertical", v, 1), ("horizontal", h, 0)):
    if ax == 1:
        e = nz.reshape(H, 4, W, 4).any(1)            # [H][W][edge]
    else:
        e = nz.reshape(H, 4, W, 4).transpose(0, 2, 3, 1).any(2)   # [H][W][edge]
    print(name, "per-MB edge has any bS: per edge index", e.mean((0, 1)).round(3), "all", e.mean().round(3))
    # wave = 8 octets: 2 consecutive rows x 4 pictures ~ independent; approximate with 8 random MBs
    rng = np.random.default_rng(1)
    flat = e.reshape(-1, 4)
    pick = flat[rng.integers(0, len(flat), (20000, 8))]
    print("   wave of 8 MBs executes edge: ", pick.any(1).mean(0).round(3))
print("intra MBs %.3f, skip %.3f, MBs with no edge at all %.3f" % (intra.mean(), (r["mb_type"] == 5).mean(),
      1 - (v.reshape(H, 4, W, 4).any((1, 3)) | h.reshape(H, 4, W, 4).any((1, 3))).mean()))
t = r["mb_type"]
print("I4x4 %.4f  I16x16 %.4f of all MBs" % ((t == 0).mean(), (t == 1).mean()))
up = np.zeros_like(intra); up[1:] = intra[:-1]
ul = np.zeros_like(intra); ul[1:, 1:] = intra[:-1, :-1]
ur = np.zeros_like(intra); ur[1:, :-1] = intra[:-1, 1:]
le = np.zeros_like(intra); le[:, 1:] = intra[:, :-1]
dep = intra & (up | ul | ur | le)
print("intra MBs with an intra neighbour (L/UL/U/UR): %.3f of intra; left only %.3f" % (dep.sum() / intra.sum(), (intra & le).sum() / intra.sum()))
print("intra per row: mean %.2f max %d" % (intra.sum(1).mean(), intra.sum(1).max()))
cb = np.array([bin(int(m) & 0xffff).count("1") for m in mask[intra]])
print("coded luma blocks per intra MB: mean %.2f" % cb.mean())
