#!/usr/bin/env python3
"""Records the CABAC known-answer vectors: random bin sequences ENCODED by the real reference's arithmetic coder
(oracle/_ref/libp264ref_kat.so: refk_cabac_encode -> p264_cabac_encode_*, core/cabac.c:907-1018, contexts initialised by
p264_cabac_context_init, core/cabac.c:819-837) for I and P/B context tables, every cabac_init_idc, slice QPs over the whole
range.  Output: tests/golden/kat_cabac.npz = (slice kind, idc, qp, ops, bins, bytes) per case - data only.  Run in the
build container; tests/test_cabac_kat.py makes the product's engine decode the bytes anywhere."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libp264ref_kat.so"))
rng = np.random.default_rng(9331)
N_CASES, N_OPS, CAP = 96, 3000, 8192
ops = np.zeros((N_CASES, N_OPS), np.int16)
bins = np.zeros((N_CASES, N_OPS), np.uint8)
par = np.zeros((N_CASES, 4), np.int32)           # is_i_slice, cabac_init_idc, slice_qp, bytes
data = np.zeros((N_CASES, CAP), np.uint8)
T_I, T_P = lib.refk_slice_type_i(), lib.refk_slice_type_p()
for k in range(N_CASES):
    is_i = int(k % 4 == 0)
    idc = int(rng.integers(0, 3))
    qp = int(k % 52) if k < 52 else int(rng.integers(0, 52))
    style = k % 3
    o = np.where(rng.random(N_OPS) < (0.15, 0.5, 0.02)[style], -1, rng.integers(0, 436, N_OPS)).astype(np.int16)
    if style == 0:                                # few contexts used over and over: states walk to their ends
        o[o >= 0] = rng.choice(rng.integers(0, 436, 6), size=int((o >= 0).sum()))
    o[rng.random(N_OPS) < 0.01] = -2              # terminate bins, value 0 (a 1 would end the slice)
    b = (rng.random(N_OPS) < (0.5, 0.2, 0.9)[style]).astype(np.uint8)   # skewed bins: long MPS runs and long LPS runs
    b[o == -2] = 0
    o[-1], b[-1] = -2, 1                          # the slice ends with a terminate bin of 1, then the flush
    n = lib.refk_cabac_encode(T_I if is_i else T_P, qp, idc, o.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), N_OPS,
                              data[k].ctypes.data_as(C.c_void_p), CAP)
    assert 0 < n < CAP - 8
    ops[k], bins[k], par[k] = o, b, (is_i, idc, qp, n)
np.savez_compressed(os.path.join(HERE, "kat_cabac.npz"), ops=ops, bins=bins, par=par, data=data[:, :int(par[:, 3].max()) + 8])
print("cases", N_CASES, "bytes", int(par[:, 3].sum()))
