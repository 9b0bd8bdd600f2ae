"""The reference's kernel-level known answers (tests/golden/kat_hotpath.npz, recorded from its own function tables by
tests/golden/make_kat.py) driven straight through the HIP kernels - not through the oracle: a mismatch here names the
kernel and the case.  Each case becomes a small picture at the CPU->GPU seam (p264hip_write_frame + p264hip_submit):
  * motion compensation: the KAT's 64x48 reference frame, one macroblock whose partition carries the
    case's vector, nothing coded, loop filter off - every quarter-pel phase, block size and picture corner of the KAT;
  * dequant_4x4 + add4x4_idct (a2, a6): the case's prediction block in the reference frame (zero vector), its coefficients
    as the block's coded levels at the case's QP - every QP and the int16-wrap cases (A-Q8)."""
import os

import numpy as np
import pytest

from p264decoder_amd import HipReconstructor, _native as N
from tests import seam_fuzz

pytestmark = pytest.mark.gpu
ZIGZAG = [0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15]          # scan position -> raster position


@pytest.fixture(scope="module")
def kat():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_hotpath.npz"))


def p_picture(mb_w, mb_h):
    pic = seam_fuzz.SeamPicture(mb_w, mb_h)
    d = pic.desc
    d.slice_type, d.dst_slot, d.n_ref, d.deblock = N.SLICE_P, 0, 1, 0
    d.ref_slot[0] = 1
    pic.rec["mb_type"] = N.MB_P_L0
    pic.rec["qp"] = 26
    return pic


def test_motion_compensation_kernels_on_reference_vectors(lib, kat):
    Y, U, V = (np.ascontiguousarray(kat[k]) for k in ("mc_y", "mc_u", "mc_v"))
    mb_h, mb_w = Y.shape[0] // 16, Y.shape[1] // 16
    hip = HipReconstructor(mb_w, mb_h, n_streams=1, slots=2, max_pictures=1, lib=lib)
    hip.write_frame(0, 1, Y, U, V)
    oy = ou = 0
    for (mbx, mby, x, y, bw, bh, mvx, mvy) in kat["mc_cases"].tolist():
        pic = p_picture(mb_w, mb_h)
        mv = pic.mv.reshape(mb_w * mb_h, 4, 4, 2)
        mv[mby * mb_w + mbx, y:y + bh, x:x + bw] = (mvx, mvy)
        if bw < 2 or bh < 2:
            pic.rec["mb_type"][mby * mb_w + mbx] = N.MB_P_8x8
        hip.submit(0, pic.seal())
        gy, gu, gv = hip.read_frame(0, 0)
        ly, lc = 16 * bw * bh, 4 * bw * bh
        Y0, X0 = mby * 16 + 4 * y, mbx * 16 + 4 * x
        what = "mv (%d,%d) at MB (%d,%d)+(%d,%d) %dx%d" % (mvx, mvy, mbx, mby, x, y, bw, bh)
        assert np.array_equal(gy[Y0:Y0 + 4 * bh, X0:X0 + 4 * bw].reshape(-1), kat["mc_oy"][oy:oy + ly]), "luma " + what
        assert np.array_equal(gu[Y0 // 2:Y0 // 2 + 2 * bh, X0 // 2:X0 // 2 + 2 * bw].reshape(-1), kat["mc_ou"][ou:ou + lc]), "Cb " + what
        assert np.array_equal(gv[Y0 // 2:Y0 // 2 + 2 * bh, X0 // 2:X0 // 2 + 2 * bw].reshape(-1), kat["mc_ov"][ou:ou + lc]), "Cr " + what
        oy += ly
        ou += lc
    hip.close()


def test_residual_in_the_mc_kernels_on_reference_vectors(lib, kat):
    """One case per macroblock (its block 0; the macroblock's QP is the case's QP), 12 cases per 4x3-macroblock picture.
    The luma cases run through the luma kernels; the reference's chroma lists use the same flat matrix, so every case is
    valid for luma."""
    coef, qp, dst, want = kat["di_coef"], kat["di_qp"], kat["di_dst"], kat["di_rec"]
    mb_w, mb_h = 4, 3
    n_mb = mb_w * mb_h
    hip = HipReconstructor(mb_w, mb_h, n_streams=1, slots=2, max_pictures=1, lib=lib)
    for first in range(0, len(qp), n_mb):
        cases = list(range(first, min(first + n_mb, len(qp))))
        ref = np.full((mb_h * 16, mb_w * 16), 128, np.uint8)
        pic = p_picture(mb_w, mb_h)
        blocks = []
        for m, i in enumerate(cases):
            my, mx = divmod(m, mb_w)
            ref[my * 16:my * 16 + 4, mx * 16:mx * 16 + 4] = dst[i].reshape(4, 4)
            lv = np.zeros(16, np.int16)
            lv[:] = coef[i][ZIGZAG]                                       # levels in scan order: level k sits at raster ZIGZAG[k]
            r = pic.rec[m]
            r["qp"], r["cbp"], r["coef_mask"], r["coef_index"] = int(qp[i]), 1, 1, len(blocks)
            blocks.append(lv)
        for m in range(len(cases), n_mb):
            pic.rec[m]["coef_index"] = len(blocks)
        pic.desc.n_coef_blocks = len(blocks)
        pic.coefs = np.concatenate(blocks).astype(np.int16)
        c = np.full((mb_h * 8, mb_w * 8), 128, np.uint8)
        hip.write_frame(0, 1, ref, c, c)
        hip.submit(0, pic.seal())
        gy, _, _ = hip.read_frame(0, 0)
        for m, i in enumerate(cases):
            my, mx = divmod(m, mb_w)
            got = gy[my * 16:my * 16 + 4, mx * 16:mx * 16 + 4].reshape(-1)
            assert np.array_equal(got, want[i]), "dequant + inverse transform case %d (QP %d): got %s want %s" % (i, qp[i], got.tolist(), want[i].tolist())
    hip.close()
