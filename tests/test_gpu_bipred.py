"""SURVEY 8f rank 4, first piece: the two bi-prediction combines of the reference (pf->avg[] / pf->avg_weight[],
core/mc.c:76-155) as device functions, through the C ABI (p264hip_bipred_frames) on whole frames, against (1) the
known-answer vectors recorded from the reference's own function tables and (2) the CPU oracle on random frames."""
import ctypes as C
import os

import numpy as np
import pytest

from p264decoder_amd import HipReconstructor

pytestmark = pytest.mark.gpu


def run(hip, lib, a, b, weighted, w1):
    hip.write_frame(0, 0, *a)
    hip.write_frame(0, 1, *b)
    lib.p264hip_bipred_frames.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    assert lib.p264hip_bipred_frames(hip.h, 0, 0, 1, weighted, w1) == 0
    return hip.read_frame(0, 0)


def test_bipred_against_reference_vectors(lib):
    kat = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_bipred.npz"))
    hip = HipReconstructor(2, 1, n_streams=1, slots=2, max_pictures=1, lib=lib)           # 32x16 luma: a 16x24 case fits
    for a, b, want, (which, w, h, weighted, w1) in zip(kat["a"], kat["b"], kat["out"], kat["par"].tolist()):
        ya, yb = np.zeros((16, 32), np.uint8), np.zeros((16, 32), np.uint8)
        ya[:, :24] = a; yb[:, :24] = b
        ca, cb = np.zeros((8, 16), np.uint8), np.zeros((8, 16), np.uint8)
        ca[:, :12] = a[:8, :12]; cb[:, :12] = b[:8, :12]                                    # the chroma planes go through the same code
        y, u, v = run(hip, lib, (ya, ca, ca), (yb, cb, cb), weighted, w1)
        assert np.array_equal(y[:h, :w], want[:h, :w]), "size %dx%d weighted %d w1 %d" % (w, h, weighted, w1)
    hip.close()


@pytest.mark.parametrize("weighted,w1", [(0, 0), (1, 32), (1, -64), (1, 128), (1, 17), (1, 0), (1, 64), (1, 99)])
def test_bipred_frames_against_oracle(lib, oracle, weighted, w1):
    rng = np.random.default_rng(1007 + w1)
    mb_w, mb_h = 9, 5
    hip = HipReconstructor(mb_w, mb_h, n_streams=1, slots=2, max_pictures=1, lib=lib)
    shapes = [(mb_h * 16, mb_w * 16), (mb_h * 8, mb_w * 8), (mb_h * 8, mb_w * 8)]
    a = [rng.integers(0, 256, s, dtype=np.uint8) for s in shapes]
    b = [rng.integers(0, 256, s, dtype=np.uint8) for s in shapes]
    got = run(hip, lib, a, b, weighted, w1)
    oracle.oracle_bipred_avg.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
    oracle.oracle_bipred_weight.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    for p in range(3):
        o = a[p].copy()
        hh, ww = o.shape
        if weighted:
            oracle.oracle_bipred_weight(o.ctypes.data, ww, b[p].ctypes.data, ww, ww, hh, w1)
        else:
            oracle.oracle_bipred_avg(o.ctypes.data, ww, b[p].ctypes.data, ww, ww, hh)
        assert np.array_equal(got[p], o), "plane %d" % p
    hip.close()
