#!/usr/bin/env python3
"""Records known-answer vectors of the B-picture derivations from the REAL reference (oracle/_ref/libp264ref_kat.so = reference
objects + oracle/ref_kat.c): p264_macroblock_bipred_init (core/macroblock.c:1400-1430: distance scale factors and implicit
weights) and p264_mb_predict_mv_direct16x16 (core/macroblock.c:254-413: spatial and temporal direct prediction) - encoder-side
functions the reference's decoder never reaches, but non-static and callable.  Output: tests/golden/kat_direct.npz (inputs and
the reference's outputs only).  Run in the build container; tests/test_direct_kat.py replays the vectors through the parser's
own derivations anywhere."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libp264ref_kat.so"))
assert lib.refk_init() == 0
rng = np.random.default_rng(2640)


def P(a):
    return a.ctypes.data_as(C.c_void_p)


out = {}

# ---- implicit weights: lists of 1..4 pictures, order counts around the current picture (incl. equal counts in both lists,
#      far apart pictures that leave the weight range, pictures on the same side) ----
n = 300
bw_n0 = rng.integers(1, 5, n).astype(np.int32)
bw_n1 = rng.integers(1, 5, n).astype(np.int32)
bw_poc0 = np.zeros((n, 8), np.int32)
bw_poc1 = np.zeros((n, 8), np.int32)
bw_cur = np.zeros(n, np.int32)
bw_dsf = np.zeros((n, 256), np.int32)
bw_w = np.zeros((n, 256), np.int32)
for i in range(n):
    span = int(rng.choice([4, 16, 60, 300]))
    bw_cur[i] = int(rng.integers(-50, 400)) * 2
    bw_poc0[i, :bw_n0[i]] = bw_cur[i] + 2 * rng.integers(-span, span + 1, bw_n0[i])
    bw_poc1[i, :bw_n1[i]] = bw_cur[i] + 2 * rng.integers(-span, span + 1, bw_n1[i])
    if rng.random() < 0.2:
        bw_poc1[i, 0] = bw_poc0[i, 0]                                          # the same picture heads both lists
    lib.refk_bipred_init(int(bw_n0[i]), P(bw_poc0[i]), int(bw_n1[i]), P(bw_poc1[i]), int(bw_cur[i]), 1, P(bw_dsf[i]), P(bw_w[i]))
out.update(bw_n0=bw_n0, bw_n1=bw_n1, bw_poc0=bw_poc0, bw_poc1=bw_poc1, bw_cur=bw_cur, bw_dsf=bw_dsf, bw_w=bw_w)


# ---- direct prediction ----
def draw_neighbours(n_ref):
    ref = np.full((2, 4), -1, np.int8)
    mv = np.zeros((2, 4, 2), np.int16)
    style = rng.random()
    for k in range(4):                                                          # A, B, C, D
        if rng.random() < 0.15:
            ref[:, k] = -2                                                      # macroblock not available
            continue
        if rng.random() < 0.12:
            continue                                                            # intra
        d = int(rng.integers(0, 3))                                             # list 0, list 1, both
        for l in range(2):
            if d == 2 or d == l:
                ref[l, k] = int(rng.integers(0, n_ref))
                mv[l, k] = rng.integers(-3, 4, 2) if style < 0.3 else rng.integers(-200, 201, 2)
    return ref, mv


def draw_col(n_col):
    intra = int(rng.random() < 0.1)
    ref = np.full((2, 4), -1, np.int8)
    mv = np.zeros((2, 16, 2), np.int16)
    still = rng.random() < 0.5
    for q in range(4):
        d = int(rng.choice([0, 0, 0, 2, 1]))
        for l in range(2):
            if d == 2 or d == l:
                ref[l, q] = 0 if rng.random() < 0.6 else int(rng.integers(0, n_col))
    for l in range(2):
        mv[l] = rng.integers(-2, 3, (16, 2)) if still else rng.integers(-120, 121, (16, 2))
        for q in range(4):
            if ref[l, q] < 0:
                for k in range(4):
                    mv[l, (q >> 1) * 8 + (q & 1) * 2 + (k >> 1) * 4 + (k & 1)] = 0
    return intra, ref, mv


n = 1200
keys = dict(spatial=np.zeros(n, np.int32), nb_ref=np.zeros((n, 2, 4), np.int8), nb_mv=np.zeros((n, 2, 4, 2), np.int16),
            col_intra=np.zeros(n, np.int32), col_ref=np.zeros((n, 2, 4), np.int8), col_mv=np.zeros((n, 2, 16, 2), np.int16),
            n0=np.zeros(n, np.int32), poc0=np.zeros((n, 8), np.int32), poc1_0=np.zeros(n, np.int32), cur=np.zeros(n, np.int32),
            n_col=np.zeros(n, np.int32), col_poc=np.zeros((n, 8), np.int32), map_col=np.zeros((n, 16), np.int32), dsf0=np.zeros((n, 16), np.int32),
            ok=np.zeros(n, np.int32), out_ref=np.zeros((n, 2, 4), np.int8), out_mv=np.zeros((n, 2, 16, 2), np.int16))
for i in range(n):
    k = {name: a[i:i + 1] for name, a in keys.items()}
    spatial = int(i % 2 == 0)
    n0 = int(rng.integers(1, 4))
    cur = int(rng.integers(10, 200)) * 2
    poc0 = np.zeros(8, np.int32)
    past = sorted({cur - 2 * int(d) for d in rng.integers(1, 9, n0)}, reverse=True)    # list 0: pictures before the current one, nearest first
    n0 = len(past)
    poc0[:n0] = past
    poc1_0 = cur + 2 * int(rng.integers(1, 6))                                   # RefPicList1[0]: the next picture after it
    # the list 0 the co-located picture was decoded with: mostly pictures of the current list 0, sometimes one that has left it
    n_col = int(rng.integers(1, 4))
    col_poc = np.zeros(8, np.int32)
    col_poc[:n_col] = [int(poc0[int(rng.integers(0, n0))]) if rng.random() < 0.85 else cur - 40 - 2 * c for c in range(n_col)]
    map_col = np.full(16, -2, np.int32)
    for c in range(n_col):
        for jj in range(n0):
            if poc0[jj] == col_poc[c]:
                map_col[c] = jj
                break
    dsf_all, w_all = np.zeros(256, np.int32), np.zeros(256, np.int32)
    one = np.array([poc1_0] + [0] * 7, np.int32)
    lib.refk_bipred_init(n0, P(poc0), 1, P(one), cur, 1, P(dsf_all), P(w_all))
    dsf0 = dsf_all.reshape(16, 16)[:, 0].copy()
    nb_ref, nb_mv = draw_neighbours(n0)
    col_intra, col_ref, col_mv = draw_col(n_col)
    o_ref, o_mv = np.zeros((2, 4), np.int8), np.zeros((2, 16, 2), np.int16)
    ok = lib.refk_direct(spatial, P(nb_ref), P(nb_mv), col_intra, P(col_ref), P(col_mv), P(map_col), P(dsf0), P(o_ref), P(o_mv))
    k["spatial"][0] = spatial; k["nb_ref"][0] = nb_ref; k["nb_mv"][0] = nb_mv; k["col_intra"][0] = col_intra; k["col_ref"][0] = col_ref
    k["col_mv"][0] = col_mv; k["n0"][0] = n0; k["poc0"][0] = poc0; k["poc1_0"][0] = poc1_0; k["cur"][0] = cur; k["n_col"][0] = n_col
    k["col_poc"][0] = col_poc; k["map_col"][0] = map_col; k["dsf0"][0] = dsf0; k["ok"][0] = ok; k["out_ref"][0] = o_ref; k["out_mv"][0] = o_mv
out.update({"d_" + name: a for name, a in keys.items()})

np.savez_compressed(os.path.join(HERE, "kat_direct.npz"), **out)
print("kat_direct.npz:", {k: v.shape for k, v in out.items()})
print("direct: available %d of %d (spatial %d, temporal %d)" % (int(keys["ok"].sum()), n, int(keys["ok"][0::2].sum()), int(keys["ok"][1::2].sum())))
