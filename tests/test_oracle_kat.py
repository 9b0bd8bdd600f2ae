"""Pins the CPU oracle, kernel by kernel, to known answers recorded from the REAL reference's
function tables (tests/golden/kat_hotpath.npz, made by tests/golden/make_kat.py from oracle/_ref):
dequant + inverse transforms at every QP incl. int16 wrap, all intra predictors
(a7-a9), motion compensation for every quarter-pel phase / block size at picture corners
(a10-a13) and the eight deblocking sample filters (a15).  Bit-exact or fail."""
import ctypes as C
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def kat():
    return np.load(os.path.join(GOLDEN, "kat_hotpath.npz"))


def P(a):
    return a.ctypes.data_as(C.c_void_p)


def test_dequant_idct_add(oracle, kat):
    coef, qp, dst = kat["di_coef"], kat["di_qp"], kat["di_dst"]
    for i in range(len(qp)):
        d = coef[i].copy()
        r = dst[i].copy()
        oracle.oracle_dequant4x4(P(d), int(qp[i]))
        assert np.array_equal(d, kat["di_deq"][i]), "dequant case %d qp %d" % (i, qp[i])
        oracle.oracle_add4x4_idct(P(r), 4, P(d))
        assert np.array_equal(r, kat["di_rec"][i]), "idct case %d qp %d" % (i, qp[i])


def test_dc_transforms(oracle, kat):
    for i in range(len(kat["ldc_qp"])):
        d = kat["ldc_in"][i].copy()
        oracle.oracle_idct4x4dc(P(d))
        oracle.oracle_dequant4x4_dc(P(d), int(kat["ldc_qp"][i]))
        assert np.array_equal(d, kat["ldc_out"][i]), "luma DC case %d" % i
    for i in range(len(kat["cdc_qp"])):
        d = kat["cdc_in"][i].copy()
        oracle.oracle_idct2x2dc(P(d))
        oracle.oracle_dequant2x2_dc(P(d), int(kat["cdc_qp"][i]))
        assert np.array_equal(d, kat["cdc_out"][i]), "chroma DC case %d" % i


@pytest.mark.parametrize("name,fn", [("p16", "oracle_pred16x16"), ("p8", "oracle_pred8x8c"), ("p4", "oracle_pred4x4")])
def test_intra_predictors(oracle, kat, name, fn):
    tiles, modes, want = kat[name + "_in"], kat[name + "_mode"], kat[name + "_out"]
    S = tiles.shape[2]
    for i in range(len(modes)):
        t = tiles[i].copy()
        getattr(oracle, fn)(C.c_void_p(t.ctypes.data + S + 1), S, int(modes[i]))
        assert np.array_equal(t, want[i]), "%s mode %d case %d" % (name, modes[i], i)


def test_motion_compensation(oracle, kat):
    Y, U, V = (np.ascontiguousarray(kat[k]) for k in ("mc_y", "mc_u", "mc_v"))
    H, W = Y.shape
    oy = ou = 0
    for (mbx, mby, x, y, bw, bh, mvx, mvy) in kat["mc_cases"].tolist():
        ly, lc = 16 * bw * bh, 4 * bw * bh
        a = np.zeros(ly, np.uint8)
        b = np.zeros(lc, np.uint8)
        c = np.zeros(lc, np.uint8)
        oracle.oracle_mc_luma(P(Y), W, H, mbx * 16 + 4 * x, mby * 16 + 4 * y, mvx, mvy, 4 * bw, 4 * bh, P(a), 4 * bw)
        oracle.oracle_mc_chroma(P(U), W // 2, H // 2, mbx * 8 + 2 * x, mby * 8 + 2 * y, mvx, mvy, 2 * bw, 2 * bh, P(b), 2 * bw)
        oracle.oracle_mc_chroma(P(V), W // 2, H // 2, mbx * 8 + 2 * x, mby * 8 + 2 * y, mvx, mvy, 2 * bw, 2 * bh, P(c), 2 * bw)
        assert np.array_equal(a, kat["mc_oy"][oy:oy + ly]), "luma mv (%d,%d) at MB (%d,%d)+(%d,%d) %dx%d" % (mvx, mvy, mbx, mby, x, y, bw, bh)
        assert np.array_equal(b, kat["mc_ou"][ou:ou + lc]) and np.array_equal(c, kat["mc_ov"][ou:ou + lc]), "chroma mv (%d,%d)" % (mvx, mvy)
        oy += ly
        ou += lc


def test_deblock_sample_filters(oracle, kat):
    S = 24
    for i, (which, alpha, beta, *tc) in enumerate(kat["db_par"].tolist()):
        t = kat["db_in"][i].copy()
        horiz_edge = which % 2 == 0
        pix = C.c_void_p(t.ctypes.data + ((8 * S + 4) if horiz_edge else (4 * S + 8)))
        xs, ys = (S, 1) if horiz_edge else (1, S)           # deblock_v_* filter across rows (core/frame.c:342-349)
        tcv = np.array(tc, np.int8)
        kind = which & ~1
        if kind == 0:
            oracle.oracle_deblock_luma(pix, xs, ys, alpha, beta, P(tcv))
        elif kind == 2:
            oracle.oracle_deblock_chroma(pix, xs, ys, alpha, beta, P(tcv))
        elif kind == 4:
            oracle.oracle_deblock_luma_intra(pix, xs, ys, alpha, beta)
        else:
            oracle.oracle_deblock_chroma_intra(pix, xs, ys, alpha, beta)
        assert np.array_equal(t, kat["db_out"][i]), "deblock filter %d case %d" % (which, i)


def test_bipred_average_and_weight(oracle):
    """pf->avg[] / pf->avg_weight[] of the reference (core/mc.c:76-155), all ten block sizes, weights over
    the whole implicit range incl. negative ones (tests/golden/kat_bipred.npz, recorded from the real function tables)."""
    kat = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_bipred.npz"))
    oracle.oracle_bipred_avg.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
    oracle.oracle_bipred_weight.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    for a, b, want, (which, w, h, weighted, w1) in zip(kat["a"], kat["b"], kat["out"], kat["par"].tolist()):
        o = a.copy()
        if weighted:
            oracle.oracle_bipred_weight(P(o), 24, P(np.ascontiguousarray(b)), 24, w, h, w1)
        else:
            oracle.oracle_bipred_avg(P(o), 24, P(np.ascontiguousarray(b)), 24, w, h)
        assert np.array_equal(o, want), "size %dx%d weighted %d w1 %d" % (w, h, weighted, w1)


def test_loop_filter_against_the_reference_frame_driver(oracle):
    """tests/golden/kat_deblock_frame.npz (p264_frame_deblocking_filter on hand-made state, make_kat_frame.py): the oracle's
    boundary strengths, table look-ups, filters and their order on 160 pictures; the same pictures go through the HIP kernels in
    tests/test_gpu_kat_frame.py."""
    from tests import kat_seam as K, oracle_bind
    kat = K.deblock_frames()
    for i in range(len(kat["dbf_par"])):
        pic, ref, want = K.deblock_frame_case(kat, i)
        store = oracle_bind.FrameStore(pic.mb_w, pic.mb_h, 3)
        for s in (1, 2):
            for dst, src in zip(store[s], ref):
                dst[:] = src
        for name, a, b in zip("yuv", oracle_bind.reconstruct(oracle, store, pic), want):
            assert np.array_equal(a, b), "case %d plane %s" % (i, name)


def test_seam_pictures_of_the_intra_and_dc_vectors(oracle):
    """The pictures tests/test_gpu_kat_intra.py feeds the HIP kernels (tests/kat_seam.py), through the oracle: the builders put
    every case where the kernels will look for it."""
    from tests import kat_seam as K, oracle_bind
    kat = K.hotpath()

    def run(pic, ref):
        store = oracle_bind.FrameStore(pic.mb_w, pic.mb_h, 2)
        for dst, src in zip(store[1], ref):
            dst[:] = src
        return oracle_bind.reconstruct(oracle, store, pic)
    for i in range(len(kat["p16_mode"])):
        y, u, v = run(*K.pred16_case(kat["p16_in"][i], kat["p16_mode"][i], i))
        assert np.array_equal(y[16:32, 16:32], kat["p16_out"][i][1:17, 1:17]), i
    for i in range(0, len(kat["p8_mode"]), 2):
        y, u, v = run(*K.pred8_case(kat["p8_in"][i], kat["p8_in"][i + 1], kat["p8_mode"][i], i))
        assert np.array_equal(u[8:16, 8:16], kat["p8_out"][i][1:9, 1:9]) and np.array_equal(v[8:16, 8:16], kat["p8_out"][i + 1][1:9, 1:9]), i
    for i in range(len(kat["p4_mode"])):
        y, u, v = run(*K.pred4_case(kat["p4_in"][i], kat["p4_mode"][i], i))
        assert np.array_equal(y[16:20, 16:20], kat["p4_out"][i][1:5, 1:5]), i
    cases = list(range(0, 400, 7))
    y, u, v = run(*K.luma_dc_picture(cases, kat["ldc_in"], kat["ldc_qp"], 128))
    for k, i in enumerate(cases):
        assert np.array_equal(y[16:32, k * 16:k * 16 + 16], np.kron(K.dc_only(128, kat["ldc_out"][i]).reshape(4, 4), np.ones((4, 4), np.uint8))), i
    cases = [i for i in range(0, 400, 5) if int(kat["cdc_qp"][i]) in K.LUMA_QP_FOR_CHROMA]
    for intra in (False, True):
        y, u, v = run(*K.chroma_dc_picture(cases, kat["cdc_in"], kat["cdc_qp"], 128, intra))
        for k, i in enumerate(cases):
            want = np.kron(K.dc_only(128, kat["cdc_out"][i]).reshape(2, 2), np.ones((4, 4), np.uint8))
            assert np.array_equal(u[8:16, k * 8:k * 8 + 8], want) and np.array_equal(v[8:16, k * 8:k * 8 + 8], want), (i, intra)
