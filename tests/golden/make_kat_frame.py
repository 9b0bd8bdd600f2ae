#!/usr/bin/env python3
"""Records frame-level known answers of the loop filter by driving the REAL reference's whole deblocking driver
(p264_frame_deblocking_filter, core/frame.c:490-643, through oracle/ref_kat.c: refk_deblock_frame) on hand-made state:
boundary-strength derivation (:535-581), edge QPs (:593-601), alpha / beta / tc0 tables and the chroma tc0 + 1 (:262-291,
472-488), the eight sample filters (:302-470) and the raster order they run in.  Output: tests/golden/kat_deblock_frame.npz.
Build container only (needs oracle/_ref); tests/test_gpu_kat_frame.py drives the same pictures through k_deblock_bs /
k_deblock (and the edge-info role of k_intra_sparse), tests/test_oracle_kat.py through the CPU oracle.

Why not tests/golden/kat_hotpath.npz's db_* vectors: they call the sample filters with FREE alpha / beta / tc0 (and, for the
strong filters, free samples on both sides).  The kernels take those parameters out of the tables by QP and offsets, and a
strength-4 edge has an intra macroblock on one side whose samples are a prediction - of the 1 600 db_* cases one inter case has
parameters the tables can produce.  Here every case is something a picture can contain:

  * a 4x3-macroblock P picture, every macroblock with its own QP; chroma_qp_index_offset, alpha / beta offsets per case;
  * inter macroblocks whose samples are arbitrary (the test gets them there by motion compensation from a reference frame:
    whole-macroblock integer vectors, macroblocks reading their own position or swapped in pairs - so vectors differ by >= 4
    across some edges), a reference index per 8x8 (two list entries holding the same frame: strength 1 without touching the
    samples), any subset of 4x4 blocks marked as coded (with all-zero levels at the seam: strength 2 without touching them);
  * intra 16x16 macroblocks without residual (strength 3 inside, 4 on their edges) whose samples are what the reference's
    own predict_16x16[] / predict_8x8c[] make of their neighbours.
Stored per case: the pre-filter planes, the macroblock arrays, the reference's filtered planes (as a difference)."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libp264ref_kat.so"))
assert lib.refk_init() == 0
rng = np.random.default_rng(26406)
MBW, MBH, N = 4, 3, 160
n = MBW * MBH


def P(a):
    return a.ctypes.data_as(C.c_void_p)


out = {k: [] for k in ("y", "u", "v", "dy", "du", "dv", "intra", "qp", "mask", "ref8", "src", "modes", "par")}
changed = 0
for case in range(N):
    # ---- parameters
    lo = [0, 14, 14, 20, 28, 36, 44][case % 7]
    hi = [51, 30, 51, 40, 51, 51, 51][case % 7]
    base_qp = int(rng.integers(lo, hi + 1))
    spread = int(rng.integers(0, 7))
    qp = np.clip(base_qp + rng.integers(-spread, spread + 1, n), 0, 51).astype(np.uint8)
    cqo, a_off, b_off = int(rng.integers(-12, 13)), int(rng.integers(-6, 7)), int(rng.integers(-6, 7))
    if case % 5 == 0:
        a_off = b_off = 0
    intra = (rng.random(n) < 0.2).astype(np.uint8)
    if case % 11 == 0:
        intra[:] = 0
    # ---- samples: a smooth ramp + a level per macroblock + sometimes a level per 4x4 block + mild noise (so that the
    #      filters' conditions hold on some lines and fail on others)
    H, W = MBH * 16, MBW * 16
    yy, xx = np.mgrid[0:H, 0:W]
    amp = int(rng.integers(1, 5)) * (1 + base_qp // 12)
    Y = 40 + int(rng.integers(0, 120)) + (xx * int(rng.integers(-2, 3)) + yy * int(rng.integers(-2, 3))) // 2
    Y = Y + np.kron(rng.integers(-3 * amp, 3 * amp + 1, (MBH, MBW)), np.ones((16, 16), np.int64))
    if case % 2:
        Y = Y + np.kron(rng.integers(-amp, amp + 1, (MBH * 4, MBW * 4)), np.ones((4, 4), np.int64))
    Y = np.clip(Y + rng.integers(-amp, amp + 1, (H, W)), 0, 255).astype(np.uint8)
    UV = []
    for _ in range(2):
        c = 60 + int(rng.integers(0, 120)) + np.kron(rng.integers(-3 * amp, 3 * amp + 1, (MBH, MBW)), np.ones((8, 8), np.int64))
        if case % 2:
            c = c + np.kron(rng.integers(-amp, amp + 1, (MBH * 2, MBW * 2)), np.ones((4, 4), np.int64))
        UV.append(np.clip(c + rng.integers(-amp, amp + 1, (H // 2, W // 2)), 0, 255).astype(np.uint8))
    U, V = UV
    # ---- inter macroblocks: where each reads its samples from (src = own position, or a partner's), references, coded blocks
    src = np.arange(n, dtype=np.int32)
    inter = [m for m in range(n) if not intra[m]]
    rng.shuffle(inter)
    for k in range(0, len(inter) - 1, 2):
        if rng.random() < 0.35:
            a, b = inter[k], inter[k + 1]
            src[a], src[b] = b, a
    ref8 = np.zeros((n, 4), np.int8)
    mask = np.zeros(n, np.uint32)
    for m in range(n):
        if intra[m]:
            ref8[m] = -1
            continue
        r = rng.random()
        ref8[m] = int(rng.integers(0, 2)) if r < 0.6 else rng.integers(0, 2, 4)
        if r >= 0.6 and r < 0.75:                                   # 16x8 / 8x16 shapes
            a, b = rng.integers(0, 2, 2)
            ref8[m] = [a, a, b, b] if r < 0.68 else [a, b, a, b]
        if rng.random() < 0.45:
            bits = rng.random(16) < rng.choice([0.15, 0.5, 0.9])
            mask[m] = int(sum(1 << b for b in range(16) if bits[b]))
    mv = np.zeros((n, 16, 2), np.int16)
    for m in range(n):
        if not intra[m]:
            mv[m, :, 0] = 64 * (src[m] % MBW - m % MBW)
            mv[m, :, 1] = 64 * (src[m] // MBW - m // MBW)
    # ---- intra macroblocks: Intra16x16 / chroma modes legal for the position, samples by the reference's predictors
    modes = np.zeros((n, 2), np.uint8)                               # as coded: luma 0 V 1 H 2 DC 3 plane; chroma 0 DC 1 H 2 V 3 plane
    for m in range(n):
        if not intra[m]:
            continue
        mx, my = m % MBW, m // MBW
        L, T = mx > 0, my > 0
        lm = int(rng.choice(([0] if T else []) + ([1] if L else []) + [2] + ([3] if L and T else [])))
        cm = int(rng.choice([0] + ([1] if L else []) + ([2] if T else []) + ([3] if L and T else [])))
        modes[m] = (lm, cm)
        dc_variant = 2 if (L and T) else 4 if L else 5 if T else 6  # function-table index of the DC the reference's fix-up picks (decoder/macroblock.c:635-667)
        f16 = lm if lm != 2 else dc_variant
        lib.refk_pred16x16(C.c_void_p(Y.ctypes.data + my * 16 * W + mx * 16), W, f16)
        dc_variant_c = 0 if (L and T) else 4 if L else 5 if T else 6
        f8 = cm if cm != 0 else dc_variant_c
        for pl in (U, V):
            lib.refk_pred8x8c(C.c_void_p(pl.ctypes.data + my * 8 * (W // 2) + mx * 8), W // 2, f8)
    fy, fu, fv = Y.copy(), U.copy(), V.copy()
    assert lib.refk_deblock_frame(MBW, MBH, P(fy), P(fu), P(fv), P(intra), P(qp), P(mask), P(ref8), P(mv), cqo, a_off, b_off) == 0
    changed += int((fy != Y).sum() + (fu != U).sum() + (fv != V).sum())
    for k, a in (("y", Y), ("u", U), ("v", V), ("dy", fy.astype(np.int16) - Y), ("du", fu.astype(np.int16) - U), ("dv", fv.astype(np.int16) - V),
                 ("intra", intra), ("qp", qp), ("mask", mask), ("ref8", ref8), ("src", src), ("modes", modes), ("par", np.array([cqo, a_off, b_off], np.int32))):
        out[k].append(a)
res = {"dbf_" + k: np.array(v) for k, v in out.items()}
for k in ("dbf_dy", "dbf_du", "dbf_dv"):
    assert np.abs(res[k]).max() < 128
    res[k] = res[k].astype(np.int8)
np.savez_compressed(os.path.join(HERE, "kat_deblock_frame.npz"), **res)
print({k: v.shape for k, v in res.items()})
print("samples changed by the filter: %d of %d" % (changed, N * MBW * MBH * 384))
