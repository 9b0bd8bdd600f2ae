"""The reference's known answers as pictures at the CPU->GPU seam (include/p264hip.h): builders shared by
tests/test_oracle_kat.py (through the CPU oracle, anywhere) and tests/test_gpu_kat_intra.py / test_gpu_kat_frame.py (through
the HIP kernels).  The answers themselves are data recorded from the real reference (tests/golden/kat_hotpath.npz,
kat_deblock_frame.npz); nothing here computes an expected sample except the DC-only inverse transform below."""
import os

import numpy as np

from p264decoder_amd import _native as N
from tests import seam_fuzz

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ZIGZAG = [0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15]          # scan position -> raster position (decoder/macroblock.c:602-603)
# QP_C of a luma QP (core/macroblock.h:210-218), and the smallest luma QP that gives a chroma QP
CHROMA_QP = list(range(30)) + [29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39]
LUMA_QP_FOR_CHROMA = {c: CHROMA_QP.index(c) for c in sorted(set(CHROMA_QP))}


def hotpath():
    return np.load(os.path.join(GOLDEN, "kat_hotpath.npz"))


def deblock_frames():
    return np.load(os.path.join(GOLDEN, "kat_deblock_frame.npz"))


def base_picture(mb_w, mb_h, n_ref=1, deblock=0):
    """A P picture of inter macroblocks that copy reference slot 1 (zero vectors, nothing coded) into slot 0."""
    pic = seam_fuzz.SeamPicture(mb_w, mb_h)
    d = pic.desc
    d.slice_type, d.dst_slot, d.n_ref, d.deblock = N.SLICE_P, 0, n_ref, deblock
    for i in range(n_ref):
        d.ref_slot[i] = 1 + i
    pic.rec["mb_type"] = N.MB_P_L0
    pic.rec["qp"] = 26
    for m in range(mb_w * mb_h):
        mx, my = m % mb_w, m // mb_w
        pic.rec["avail"][m] = ((N.AVAIL_LEFT if mx else 0) | (N.AVAIL_TOP if my else 0) | (N.AVAIL_TOPRIGHT if my and mx + 1 < mb_w else 0)
                               | (N.AVAIL_TOPLEFT if mx and my else 0))
        pic.rec["edges"][m] = N.EDGE_INNER | (N.EDGE_LEFT if mx else 0) | (N.EDGE_TOP if my else 0)
    return pic


def set_blocks(pic, blocks_per_mb):
    """blocks_per_mb[m] = list of int16[16] blocks of macroblock m in the stream's order; fills coef_index / n_coef_blocks / coefs."""
    flat = []
    for m, blocks in enumerate(blocks_per_mb):
        pic.rec["coef_index"][m] = len(flat)
        flat += blocks
    pic.desc.n_coef_blocks = len(flat)
    pic.coefs = np.concatenate(flat).astype(np.int16) if flat else np.zeros(16, np.int16)
    return pic


# ---------------------------------------------------------------------------------------------------------------------
# intra predictors: a 3x2-macroblock picture, the macroblock under test in the middle of the bottom row,
# the tile's border samples in its (inter) neighbours, the availability flags chosen so that the decoder's mode fix-up
# (decoder/macroblock.c:635-753) lands on the function-table entry the case was recorded with
AV_ALL = N.AVAIL_LEFT | N.AVAIL_TOP | N.AVAIL_TOPRIGHT | N.AVAIL_TOPLEFT
PRED_SHAPE = (3, 2)                                   # mb_w, mb_h
TARGET = 4                                            # macroblock (1, 1)


def _noise_frame(rng, mb_w, mb_h):
    return [rng.randint(0, 256, (mb_h * 16, mb_w * 16)).astype(np.uint8), rng.randint(0, 256, (mb_h * 8, mb_w * 8)).astype(np.uint8),
            rng.randint(0, 256, (mb_h * 8, mb_w * 8)).astype(np.uint8)]


def pred16_case(tile, fn_index, seed):
    """tile: (17, 32) with row 0 = corner + top row, column 0 = left column.  fn_index: predict_16x16[] index (0 V, 1 H, 2 DC,
    3 plane, 4 DC_LEFT, 5 DC_TOP, 6 DC_128).  Returns (picture, reference frame [y, u, v])."""
    mb_w, mb_h = PRED_SHAPE
    ref = _noise_frame(np.random.RandomState(seed), mb_w, mb_h)
    ref[0][15, 15:47] = tile[0, :32]                      # corner, the sixteen samples above, and what the tile holds to their right
    ref[0][16:32, 15] = tile[1:17, 0]
    pic = base_picture(mb_w, mb_h)
    r = pic.rec[TARGET]
    r["mb_type"] = N.MB_I16x16
    coded, avail = {0: (0, AV_ALL), 1: (1, AV_ALL), 2: (2, AV_ALL), 3: (3, AV_ALL), 4: (2, N.AVAIL_LEFT), 5: (2, N.AVAIL_TOP | N.AVAIL_TOPRIGHT), 6: (2, 0)}[int(fn_index)]
    r["intra_modes"], r["avail"] = coded, avail
    pic.ref_idx.reshape(-1, 4)[TARGET] = -1
    return set_blocks(pic, [[] for _ in range(pic.n_mb)]).seal(), ref


def pred8_case(tile_u, tile_v, fn_index, seed):
    """predict_8x8c[] index: 0 DC, 1 H, 2 V, 3 plane, 4 DC_LEFT, 5 DC_TOP, 6 DC_128; tiles (9, 32), one per chroma plane."""
    mb_w, mb_h = PRED_SHAPE
    ref = _noise_frame(np.random.RandomState(seed), mb_w, mb_h)
    for plane, tile in ((1, tile_u), (2, tile_v)):
        ref[plane][7, 7:24] = tile[0, :17]
        ref[plane][8:16, 7] = tile[1:9, 0]
    pic = base_picture(mb_w, mb_h)
    r = pic.rec[TARGET]
    r["mb_type"] = N.MB_I16x16
    coded, avail = {0: (0, AV_ALL), 1: (1, AV_ALL), 2: (2, AV_ALL), 3: (3, AV_ALL), 4: (0, N.AVAIL_LEFT), 5: (0, N.AVAIL_TOP | N.AVAIL_TOPRIGHT), 6: (0, 0)}[int(fn_index)]
    # (the luma mode must be legal for the same flags: DC is, whatever they are)
    r["intra_modes"], r["avail"] = 2 | (coded << 4), avail
    pic.ref_idx.reshape(-1, 4)[TARGET] = -1
    return set_blocks(pic, [[] for _ in range(pic.n_mb)]).seal(), ref


def pred4_case(tile, fn_index, seed):
    """predict_4x4[] index: 0 V 1 H 2 DC 3 DDL 4 DDR 5 VR 6 HD 7 VL 8 HU 9 DC_LEFT 10 DC_TOP 11 DC_128; tile (5, 32): row 0 =
    corner, 4 top, 4 top-right samples; column 0 = left.  Block 0 of the macroblock under test carries the case."""
    mb_w, mb_h = PRED_SHAPE
    ref = _noise_frame(np.random.RandomState(seed), mb_w, mb_h)
    ref[0][15, 15:24] = tile[0, :9]
    ref[0][16:20, 15] = tile[1:5, 0]
    pic = base_picture(mb_w, mb_h)
    r = pic.rec[TARGET]
    r["mb_type"] = N.MB_I4x4
    coded, avail = (int(fn_index), AV_ALL) if fn_index < 9 else {9: (2, N.AVAIL_LEFT), 10: (2, N.AVAIL_TOP | N.AVAIL_TOPRIGHT), 11: (2, 0)}[int(fn_index)]
    r["intra_modes"], r["avail"] = 0, avail                          # chroma DC
    i4 = pic.i4modes.reshape(-1, 16)
    i4[TARGET] = 2                                                   # the other fifteen blocks: DC (legal everywhere)
    i4[TARGET, 0] = coded
    pic.ref_idx.reshape(-1, 4)[TARGET] = -1
    return set_blocks(pic, [[] for _ in range(pic.n_mb)]).seal(), ref


# ---------------------------------------------------------------------------------------------------------------------
# DC transforms + DC dequantisation (a3 - a5): a row of inter macroblocks holding a flat prediction value, below it a row of
# macroblocks under test - one case each - that predict vertically from it and carry ONLY DC levels.  A block with nothing
# but its DC coefficient d reconstructs as clip(pred + ((d + 32) >> 6)) (core/dct.c:212-236 with fifteen zeros), so every
# dequantised DC value of the reference shows in a 4x4 block - through the shift and the clip: three prediction values
# (0, 128, 255) between them show every value in [-16 416, 16 351] to within the shift.
def dc_only(pred, d):
    return np.clip(int(pred) + ((np.asarray(d, np.int64) + 32) >> 6), 0, 255).astype(np.uint8)


def luma_dc_picture(cases, ldc_in, ldc_qp, pred):
    """I16x16 macroblocks (vertical prediction from a flat row) with only the luma DC block coded."""
    mb_w = len(cases)
    pic = base_picture(mb_w, 2)
    blocks = [[] for _ in range(2 * mb_w)]
    for k, i in enumerate(cases):
        m = mb_w + k
        r = pic.rec[m]
        r["mb_type"], r["qp"], r["intra_modes"], r["cbp"], r["coef_mask"] = N.MB_I16x16, int(ldc_qp[i]), 0 | (2 << 4), 0, N.COEF_LUMA_DC
        pic.ref_idx.reshape(-1, 4)[m] = -1
        lv = np.zeros(16, np.int16)
        lv[:] = ldc_in[i][ZIGZAG]                                    # scan order: level k sits at raster position ZIGZAG[k]
        blocks[m] = [lv]
    ref = [np.full((32, mb_w * 16), pred, np.uint8), np.full((16, mb_w * 8), 128, np.uint8), np.full((16, mb_w * 8), 128, np.uint8)]
    return set_blocks(pic, blocks).seal(), ref


def chroma_dc_picture(cases, cdc_in, cdc_qp, pred, intra):
    """Macroblocks with only the chroma DC block coded (Cb and Cr both carry the case); intra: I16x16 with vertical chroma
    prediction from the flat row above (kernel_intra.h), else inter macroblocks copying the flat reference (kernel_mc.h's
    chroma roles).  Only cases whose chroma QP a luma QP can produce (<= 39, core/macroblock.h:210-218)."""
    mb_w = len(cases)
    pic = base_picture(mb_w, 2)
    blocks = [[] for _ in range(2 * mb_w)]
    for k, i in enumerate(cases):
        m = mb_w + k
        r = pic.rec[m]
        r["qp"], r["cbp"], r["coef_mask"] = LUMA_QP_FOR_CHROMA[int(cdc_qp[i])], 1 << 4, N.COEF_CHROMA_DC
        if intra:
            r["mb_type"], r["intra_modes"] = N.MB_I16x16, 0 | (2 << 4)
            pic.ref_idx.reshape(-1, 4)[m] = -1
        dc = np.zeros(16, np.int16)
        dc[0:4] = cdc_in[i]
        dc[4:8] = cdc_in[i]
        blocks[m] = [dc]
    ref = [np.full((32, mb_w * 16), 128, np.uint8), np.full((16, mb_w * 8), pred, np.uint8), np.full((16, mb_w * 8), pred, np.uint8)]
    return set_blocks(pic, blocks).seal(), ref


# ---------------------------------------------------------------------------------------------------------------------
# the loop filter at frame level (a14, a15): tests/golden/make_kat_frame.py
def deblock_frame_case(kat, i):
    """Returns (picture, reference frame, expected [y, u, v]).  Reference slots 1 and 2 hold the same frame."""
    Y, U, V = kat["dbf_y"][i], kat["dbf_u"][i], kat["dbf_v"][i]
    mb_h, mb_w = Y.shape[0] // 16, Y.shape[1] // 16
    n = mb_w * mb_h
    intra, qp, mask, ref8, src, modes = (kat["dbf_" + k][i] for k in ("intra", "qp", "mask", "ref8", "src", "modes"))
    cqo, a_off, b_off = (int(x) for x in kat["dbf_par"][i])
    rng = np.random.RandomState(1000 + i)
    ref = _noise_frame(rng, mb_w, mb_h)
    pic = base_picture(mb_w, mb_h, n_ref=2, deblock=1)
    d = pic.desc
    d.chroma_qp_offset, d.alpha_c0_offset, d.beta_offset = cqo, a_off, b_off
    mv = pic.mv.reshape(n, 16, 2)
    blocks = [[] for _ in range(n)]
    for m in range(n):
        r = pic.rec[m]
        r["qp"] = int(qp[m])
        mx, my = m % mb_w, m // mb_w
        if intra[m]:
            r["mb_type"], r["intra_modes"] = N.MB_I16x16, int(modes[m][0]) | (int(modes[m][1]) << 4)
            pic.ref_idx.reshape(-1, 4)[m] = -1
            continue
        sx, sy = int(src[m]) % mb_w, int(src[m]) // mb_w
        ref[0][sy * 16:sy * 16 + 16, sx * 16:sx * 16 + 16] = Y[my * 16:my * 16 + 16, mx * 16:mx * 16 + 16]
        ref[1][sy * 8:sy * 8 + 8, sx * 8:sx * 8 + 8] = U[my * 8:my * 8 + 8, mx * 8:mx * 8 + 8]
        ref[2][sy * 8:sy * 8 + 8, sx * 8:sx * 8 + 8] = V[my * 8:my * 8 + 8, mx * 8:mx * 8 + 8]
        mv[m, :, 0], mv[m, :, 1] = 64 * (sx - mx), 64 * (sy - my)
        q = [int(x) for x in ref8[m]]
        pic.ref_idx.reshape(-1, 4)[m] = q
        whole = q[0] == q[1] == q[2] == q[3] or (q[0] == q[1] and q[2] == q[3]) or (q[0] == q[2] and q[1] == q[3])
        r["mb_type"] = N.MB_P_L0 if whole else N.MB_P_8x8
        mk = int(mask[m])
        r["coef_mask"] = mk
        r["cbp"] = sum(1 << g for g in range(4) if (mk >> (4 * g)) & 15)
        blocks[m] = [np.zeros(16, np.int16) for b in range(16) if (mk >> b) & 1]     # coded, all levels zero: strength 2, samples untouched
    want = [(a.astype(np.int16) + kat["dbf_d" + k][i]).astype(np.uint8) for k, a in (("y", Y), ("u", U), ("v", V))]
    return set_blocks(pic, blocks).seal(), ref, want
