#!/usr/bin/env python3
"""Records kernel-level known-answer vectors by calling the REAL reference's function tables
(oracle/_ref/libp264ref_kat.so = reference objects + oracle/ref_kat.c).  Output:
tests/golden/kat_hotpath.npz (inputs and the reference's outputs only).  Run in the build
container; tests/test_oracle_kat.py replays the vectors through the CPU oracle anywhere."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libp264ref_kat.so"))
assert lib.refk_init() == 0
rng = np.random.default_rng(264)


def P(a):
    return a.ctypes.data_as(C.c_void_p)


out = {}

# ---- a2 + a6: dequant_4x4 -> add4x4_idct, all QPs, all four lists, incl. int16 overflow cases ----
n = 900
coef = np.where(rng.random((n, 16)) < 0.5, rng.integers(-40, 41, (n, 16)), 0).astype(np.int16)
coef[600:] = rng.integers(-32768, 32768, (n - 600, 16)).astype(np.int16)      # wrap-around territory (A-Q8)
qp = rng.integers(0, 52, n).astype(np.int32)
qp[:52] = np.arange(52)
lst = rng.integers(0, 4, n).astype(np.int32)
dst = rng.integers(0, 256, (n, 16)).astype(np.uint8)
deq, rec = coef.copy(), dst.copy()
for i in range(n):
    lib.refk_dequant_idct_add(P(deq[i]), int(qp[i]), int(lst[i]), P(rec[i]), 4)
out.update(di_coef=coef, di_qp=qp, di_list=lst, di_dst=dst, di_deq=deq, di_rec=rec)

# ---- a5 + a3 / a4: DC transforms + dequant ----
n = 400
d16 = rng.integers(-300, 301, (n, 16)).astype(np.int16)
d16[300:] = rng.integers(-32768, 32768, (n - 300, 16)).astype(np.int16)
q16 = rng.integers(0, 52, n).astype(np.int32)
q16[:52] = np.arange(52)
r16 = d16.copy()
for i in range(n):
    lib.refk_luma_dc(P(r16[i]), int(q16[i]))
d4 = rng.integers(-300, 301, (n, 4)).astype(np.int16)
d4[300:] = rng.integers(-32768, 32768, (n - 300, 4)).astype(np.int16)
q4 = rng.integers(0, 52, n).astype(np.int32)
q4[:52] = np.arange(52)
r4 = d4.copy()
for i in range(n):
    lib.refk_chroma_dc(P(r4[i]), int(q4[i]))
out.update(ldc_in=d16, ldc_qp=q16, ldc_out=r16, cdc_in=d4, cdc_qp=q4, cdc_out=r4)


# ---- a7-a9: intra predictors, every mode; tiles carry a 1-sample top/left border (+4 top-right for 4x4) ----
def pred_cases(size, nmodes, fn, reps):
    S = 32
    tiles, modes, res = [], [], []
    for m in range(nmodes):
        for r in range(reps):
            t = rng.integers(0, 256, (size + 1, S)).astype(np.uint8)
            if r == 0:
                t[:] = 128
            if r == 1:
                t[0, :] = 255
                t[:, 0] = 0
            o = t.copy()
            fn(C.c_void_p(o.ctypes.data + S + 1), S, m)
            tiles.append(t)
            modes.append(m)
            res.append(o)
    return np.array(tiles), np.array(modes, np.int32), np.array(res)


t, m, r = pred_cases(16, 7, lib.refk_pred16x16, 8)
out.update(p16_in=t, p16_mode=m, p16_out=r)
t, m, r = pred_cases(8, 7, lib.refk_pred8x8c, 8)
out.update(p8_in=t, p8_mode=m, p8_out=r)
t, m, r = pred_cases(4, 12, lib.refk_pred4x4, 10)
out.update(p4_in=t, p4_mode=m, p4_out=r)

# ---- a10-a13: motion compensation on a real reference frame (pads, half-pel planes, qpel average) ----
W, H = 64, 48
yy, xx = np.mgrid[0:H, 0:W]
Y = np.clip(96 + 60 * np.sin(xx / 3.1) * np.cos(yy / 2.3) + rng.integers(-40, 41, (H, W)), 0, 255).astype(np.uint8)
U = rng.integers(0, 256, (H // 2, W // 2)).astype(np.uint8)
V = rng.integers(0, 256, (H // 2, W // 2)).astype(np.uint8)
lib.refk_frame_set(P(Y), P(U), P(V), W, H)
cases, oy, ou, ov = [], [], [], []
shapes = [(4, 4), (4, 2), (2, 4), (2, 2), (2, 1), (1, 2), (1, 1)]
for k in range(1400):
    bw, bh = shapes[k % len(shapes)]
    mbx, mby = int(rng.integers(0, W // 16)), int(rng.integers(0, H // 16))
    if k % 5 == 0:
        mbx, mby = [(0, 0), (W // 16 - 1, 0), (0, H // 16 - 1), (W // 16 - 1, H // 16 - 1)][(k // 5) % 4]
    x, y = int(rng.integers(0, 5 - bw)), int(rng.integers(0, 5 - bh))
    mvx, mvy = int(rng.integers(-80, 81)), int(rng.integers(-80, 81))          # up to 20 px: inside the 24-px safe zone (A-Q9)
    if k < 256:
        mvx, mvy = (k % 16) - 8 + 4 * ((k // 16) % 4 - 2), (k // 16) - 8       # every phase pair
    a = np.zeros((bh * 4, bw * 4), np.uint8)
    b = np.zeros((bh * 2, bw * 2), np.uint8)
    c = b.copy()
    lib.refk_mc_block(mbx, mby, x, y, bw, bh, mvx, mvy, P(a), P(b), P(c))
    cases.append((mbx, mby, x, y, bw, bh, mvx, mvy))
    oy.append(a.ravel())
    ou.append(b.ravel())
    ov.append(c.ravel())
out.update(mc_y=Y, mc_u=U, mc_v=V, mc_cases=np.array(cases, np.int32),
           mc_oy=np.concatenate(oy), mc_ou=np.concatenate(ou), mc_ov=np.concatenate(ov))

# ---- a15: the eight deblocking sample filters ----
n = 1600
S = 24
bufs, afters, params = [], [], []
for i in range(n):
    which = i % 8
    base = rng.integers(0, 256)
    step = int(rng.integers(-20, 21)) if i % 3 else 0
    horiz_edge = which % 2 == 0                                                 # v_* filters work across rows
    ramp = (np.arange(S)[:, None] >= 8) if horiz_edge else (np.arange(S)[None, :] >= 8)
    t = np.clip(base + rng.integers(-12, 13, (S, S)) + step * ramp, 0, 255).astype(np.uint8)
    alpha, beta = int(rng.integers(0, 256)), int(rng.integers(0, 19))
    if i % 4 == 0:
        alpha, beta = 255, 18
    tc = rng.integers(-1, 14, 4).astype(np.int8)
    o = t.copy()
    off = (8 * S + 4) if horiz_edge else (4 * S + 8)                            # edge above row 8 / left of col 8
    lib.refk_deblock(which, C.c_void_p(o.ctypes.data + off), S, alpha, beta, P(tc))
    bufs.append(t)
    afters.append(o)
    params.append((which, alpha, beta, tc[0], tc[1], tc[2], tc[3]))
out.update(db_in=np.array(bufs), db_out=np.array(afters), db_par=np.array(params, np.int32))

np.savez_compressed(os.path.join(HERE, "kat_hotpath.npz"), **out)

# ---- 8f rank 4: bi-prediction average / implicit-weight average through pf->avg[] / pf->avg_weight[] (core/mc.c:76-155) ----
# (its own file and its own generator: the vectors above stay byte-identical)
rb = np.random.default_rng(264004)
SIZES = [(16, 16), (16, 8), (8, 16), (8, 8), (8, 4), (4, 8), (4, 4), (4, 2), (2, 4), (2, 2)]      # PIXEL_16x16 .. PIXEL_2x2 (core/pixel.h)
lib.refk_avg.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
lib.refk_avg_weight.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
bi_a, bi_b, bi_par, bi_out = [], [], [], []
for k in range(600):
    which = k % 10
    w, h = SIZES[which]
    a = rb.integers(0, 256, (16, 24)).astype(np.uint8)
    b = rb.integers(0, 256, (16, 24)).astype(np.uint8)
    if k % 7 == 0:
        a[:] = 255; b[:] = rb.integers(250, 256)
    if k % 11 == 0:
        a[:] = 0; b[:] = rb.integers(0, 4)
    weighted = k % 2
    w1 = int(rb.integers(-64, 129)) if k % 6 else [32, 0, 64, -64, 128, 21][(k // 6) % 6]
    o = a.copy()
    if weighted:
        lib.refk_avg_weight(which, P(o), 24, P(b), 24, w1)
    else:
        lib.refk_avg(which, P(o), 24, P(b), 24)
    bi_a.append(a); bi_b.append(b); bi_out.append(o); bi_par.append((which, w, h, weighted, w1))
np.savez_compressed(os.path.join(HERE, "kat_bipred.npz"), a=np.array(bi_a), b=np.array(bi_b), out=np.array(bi_out), par=np.array(bi_par, np.int32))
print("bipred cases:", len(bi_par))
print({k: v.shape for k, v in out.items()})
chg = (out["db_in"] != out["db_out"]).reshape(n, -1).any(1)
print("deblock cases that modified samples:", [int(chg[out["db_par"][:, 0] == w].sum()) for w in range(8)])
