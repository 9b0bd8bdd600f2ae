"""Random pictures built DIRECTLY at the CPU->GPU seam (include/p264hip.h), no bitstream involved.

The stream writer can only produce what a conformant CAVLC stream can carry and what the reference decodes; the seam is
wider: any macroblock type mix, every partition shape down to 4x4 with its own vector, a QP per macroblock, levels up to
the int16 limits (storage wrap, SURVEY A-Q8), several reference frames, every neighbour-availability / edge pattern a
slice structure can produce, deblocking offsets.  `make_picture` draws such a picture; the GPU test feeds the same
arrays to p264hip_submit and to the CPU oracle and compares bytes (tests/test_gpu_seam_fuzz.py).
"""
import ctypes as C

import numpy as np

from p264decoder_amd import _native as N

BLK_X = [0, 1, 0, 1, 2, 3, 2, 3, 0, 1, 0, 1, 2, 3, 2, 3]      # luma 4x4 block (decode order) -> position, core/macroblock.h:194-201
BLK_Y = [0, 0, 1, 1, 0, 0, 1, 1, 2, 2, 3, 3, 2, 2, 3, 3]
MB_DT = np.dtype([("mb_type", "u1"), ("qp", "u1"), ("cbp", "u1"), ("intra_modes", "u1"),
                  ("coef_mask", "<u4"), ("coef_index", "<u4"), ("avail", "u1"), ("edges", "u1"), ("flags", "<u2")])


class SeamPicture:
    """Owns the arrays a p264hip_picture_t points to (same interface as recon.ParsedPicture)."""

    def __init__(self, mb_w, mb_h):
        n = mb_w * mb_h
        self.rec = np.zeros(n, MB_DT)
        self.mv = np.zeros(n * 32, np.int16)
        self.ref_idx = np.zeros(n * 4, np.int8)
        self.i4modes = np.full(n * 16, 2, np.uint8)
        self.coefs = np.zeros(16, np.int16)
        self.mv_l1 = np.zeros(n * 32, np.int16)        # list 1 (B pictures)
        self.ref_idx_l1 = np.full(n * 4, -1, np.int8)
        self.desc = N.Picture()
        self.desc.mb_w, self.desc.mb_h = mb_w, mb_h

    def seal(self):
        d = self.desc
        self.mb = self.rec.view(np.uint8)
        d.mb = C.cast(self.rec.ctypes.data, C.POINTER(N.MbInfo))
        d.mv = C.cast(self.mv.ctypes.data, C.POINTER(C.c_int16))
        d.ref_idx = C.cast(self.ref_idx.ctypes.data, C.POINTER(C.c_int8))
        d.i4modes = C.cast(self.i4modes.ctypes.data, C.POINTER(C.c_uint8))
        d.coefs = C.cast(self.coefs.ctypes.data, C.POINTER(C.c_int16))
        d.mv_l1 = C.cast(self.mv_l1.ctypes.data, C.POINTER(C.c_int16))
        d.ref_idx_l1 = C.cast(self.ref_idx_l1.ctypes.data, C.POINTER(C.c_int8))
        return self

    @property
    def mb_w(self):
        return self.desc.mb_w

    @property
    def mb_h(self):
        return self.desc.mb_h

    @property
    def n_mb(self):
        return self.desc.mb_w * self.desc.mb_h

    def mb_records(self):
        return self.rec


def _levels(rng, n, style):
    """n levels in scan order.  style: 'small' typical, 'large' up to +-2000, 'wrap' up to the int16 limits."""
    lv = np.zeros(16, np.int64)
    tc = 1 + int(rng.integers(0, n)) if rng.random() < 0.5 else 1 + int(rng.integers(0, min(n, 4)))
    pos = rng.choice(n, size=min(tc, n), replace=False)
    if style == "small":
        mag = rng.integers(1, 13, size=len(pos))
    elif style == "large":
        mag = rng.integers(1, 2001, size=len(pos))
    else:
        mag = np.where(rng.random(len(pos)) < 0.5, rng.integers(20000, 32768, size=len(pos)), rng.integers(1, 4000, size=len(pos)))
    lv[pos] = mag * rng.choice([-1, 1], size=len(pos))
    if style == "wrap" and rng.random() < 0.2:
        lv[pos[0]] = -32768
    return lv.astype(np.int16)


def make_picture(rng, mb_w, mb_h, *, p_picture=True, n_ref=1, slots=2, dst_slot=0, level_style="small", qp_mode="random",
                 mv_range=80, sub8x8=True, intra_share=0.15, slices=1, deblock_offsets=True, b_picture=False, n_ref_l1=1, weighted=None, mirror_l1=0.0):
    """Draw one picture.  qp_mode: 'random' (0..51 per macroblock), 'two' (two values), or an int (constant).
    b_picture: a B picture - every inter macroblock is P264_MB_B, each 8x8 quadrant predicts from list 0, list 1 or both
    (a negative index = list unused, its vectors 0); weighted: None = drawn, else weighted_bipred on / off; mirror_l1: share of
    the B macroblocks whose list-1 vectors repeat their list-0 vectors (with the same frames in both lists: blocks that read the
    same picture through different lists with equal vectors - the case where H.264 8.7.2.1's boundary strength by PICTURE and a
    list-by-list comparison of indices differ)."""
    pic = SeamPicture(mb_w, mb_h)
    d = pic.desc
    n = mb_w * mb_h
    rec = pic.rec
    mv = pic.mv.reshape(n, 16, 2)
    ref = pic.ref_idx.reshape(n, 4)
    i4 = pic.i4modes.reshape(n, 16)
    d.slice_type = N.SLICE_B if (b_picture and p_picture) else N.SLICE_P if p_picture else N.SLICE_I
    mv1 = pic.mv_l1.reshape(n, 16, 2)
    ref1 = pic.ref_idx_l1.reshape(n, 4)
    d.chroma_qp_offset = int(rng.integers(-12, 13))
    d.alpha_c0_offset = int(rng.integers(-6, 7)) if deblock_offsets else 0
    d.beta_offset = int(rng.integers(-6, 7)) if deblock_offsets else 0
    d.dst_slot = dst_slot
    d.n_ref = n_ref if p_picture else 0
    others = [s for s in range(slots) if s != dst_slot]
    for i in range(d.n_ref):
        d.ref_slot[i] = others[i % len(others)]
    if d.slice_type == N.SLICE_B:
        d.n_ref_l1 = n_ref_l1
        for i in range(n_ref_l1):
            d.ref_slot_l1[i] = others[(len(others) - 1 - i) % len(others)]      # list 1 walks the store the other way round
        d.weighted_bipred = int(rng.random() < 0.5) if weighted is None else int(bool(weighted))
        for i in range(N.MAX_REFS * N.MAX_REFS):                           # implicit weights: 64 - dist_scale_factor, -64 .. 128; often 32
            d.bipred_weight[i] = 32 if rng.random() < 0.3 else int(rng.integers(-64, 129))
    # slice structure: first macroblock of every slice and its deblocking idc (0 all edges, 1 none, 2 not across slices)
    starts = sorted(set([0] + [int(x) for x in rng.integers(1, max(n, 2), size=slices - 1)])) if slices > 1 and n > 1 else [0]
    idcs = [int(rng.choice([0, 0, 2, 1])) for _ in starts]
    slice_of = np.zeros(n, np.int32)
    for k, s in enumerate(starts):
        slice_of[s:] = k
    d.deblock = 1 if any(i != 1 for i in idcs) else 0
    two = rng.integers(0, 52, size=2)
    blocks = []
    for m in range(n):
        mbx, mby = m % mb_w, m // mb_w
        r = rec[m]
        first = starts[slice_of[m]]

        def ok(x, y):
            return 0 <= x < mb_w and 0 <= y < mb_h and first <= y * mb_w + x < m
        L, T, TR, TL = ok(mbx - 1, mby), ok(mbx, mby - 1), ok(mbx + 1, mby - 1), ok(mbx - 1, mby - 1)
        r["avail"] = (N.AVAIL_LEFT if L else 0) | (N.AVAIL_TOP if T else 0) | (N.AVAIL_TOPRIGHT if TR else 0) | (N.AVAIL_TOPLEFT if TL else 0)
        idc = idcs[slice_of[m]]
        e = 0
        if idc != 1:
            e = N.EDGE_INNER
            if mbx > 0 and (idc == 0 or L):
                e |= N.EDGE_LEFT
            if mby > 0 and (idc == 0 or T):
                e |= N.EDGE_TOP
        r["edges"] = e
        r["qp"] = int(rng.integers(0, 52)) if qp_mode == "random" else int(two[rng.integers(0, 2)]) if qp_mode == "two" else int(qp_mode)
        style = level_style if level_style != "mixed" else str(rng.choice(["small", "small", "large", "wrap"]))
        intra = (not p_picture) or rng.random() < intra_share
        mask = 0
        mb_blocks = []
        if intra:
            ref[m] = -1
            is16 = rng.random() < 0.5
            legal_c = [0] + ([1] if L else []) + ([2] if T else []) + ([3] if L and T and TL else [])
            cmode = int(rng.choice(legal_c))
            if is16:
                r["mb_type"] = N.MB_I16x16
                legal = ([0] if T else []) + ([1] if L else []) + [2] + ([3] if L and T and TL else [])
                r["intra_modes"] = int(rng.choice(legal)) | (cmode << 4)
                cbp_l = 15 if rng.random() < 0.4 else 0
                if rng.random() < 0.7:
                    mask |= N.COEF_LUMA_DC
                    mb_blocks.append(("ldc", _levels(rng, 16, style)))
                for b in range(16):
                    if cbp_l and rng.random() < 0.4:
                        mask |= 1 << b
                n_lv = 15
            else:
                r["mb_type"] = N.MB_I4x4
                r["intra_modes"] = cmode << 4
                for b in range(16):
                    bx, by = BLK_X[b], BLK_Y[b]
                    l, t = bx > 0 or L, by > 0 or T
                    tl = True if (bx > 0 and by > 0) else T if bx > 0 else L if by > 0 else TL
                    legal = ([0] if t else []) + ([1] if l else []) + [2] + ([3, 7] if t else []) + ([4, 5, 6] if l and t and tl else []) + ([8] if l else [])
                    i4[m, b] = int(rng.choice(legal))
                cbp_l = int(rng.integers(0, 16)) if rng.random() < 0.7 else 0
                for b in range(16):
                    if (cbp_l >> (b >> 2)) & 1 and rng.random() < 0.6:
                        mask |= 1 << b
                n_lv = 16
        else:
            shape = rng.random()
            skip = shape < 0.15
            cells = np.zeros((4, 4, 2), np.int64)

            def vec():
                return rng.integers(-mv_range, mv_range + 1, size=2)
            if skip or shape < 0.45:
                cells[:] = vec()
                r["mb_type"] = N.MB_P_SKIP if skip else N.MB_P_L0
                ref[m] = 0 if skip else int(rng.integers(0, max(d.n_ref, 1)))
            elif shape < 0.55:
                cells[:2] = vec(); cells[2:] = vec()
                r["mb_type"] = N.MB_P_L0
                ra, rb = rng.integers(0, max(d.n_ref, 1), size=2)
                ref[m] = [ra, ra, rb, rb]
            elif shape < 0.65:
                cells[:, :2] = vec(); cells[:, 2:] = vec()
                r["mb_type"] = N.MB_P_L0
                ra, rb = rng.integers(0, max(d.n_ref, 1), size=2)
                ref[m] = [ra, rb, ra, rb]
            else:
                r["mb_type"] = N.MB_P_8x8
                ref[m] = rng.integers(0, max(d.n_ref, 1), size=4)
                for q in range(4):
                    qy, qx = (q >> 1) * 2, (q & 1) * 2
                    sub = int(rng.integers(0, 4)) if sub8x8 else 0
                    if sub == 0:
                        cells[qy:qy + 2, qx:qx + 2] = vec()
                    elif sub == 1:
                        cells[qy, qx:qx + 2] = vec(); cells[qy + 1, qx:qx + 2] = vec()
                    elif sub == 2:
                        cells[qy:qy + 2, qx] = vec(); cells[qy:qy + 2, qx + 1] = vec()
                    else:
                        for yy in range(2):
                            for xx in range(2):
                                cells[qy + yy, qx + xx] = vec() if rng.random() < 0.8 else cells[qy, qx]
            if rng.random() < 0.1:          # quarter-pel phase sweep: keep the integer part, force a phase
                cells = (cells & ~3) | rng.integers(0, 4, size=2)
            mv[m] = cells.reshape(16, 2)
            if d.slice_type == N.SLICE_B:
                # direction per quadrant (0 list 0, 1 list 1, 2 both); whole-macroblock directions are as likely as mixed ones
                r["mb_type"] = N.MB_B
                dirs = [int(rng.integers(0, 3))] * 4 if rng.random() < 0.5 else [int(x) for x in rng.integers(0, 3, size=4)]
                c1 = np.zeros((4, 4, 2), np.int64)
                style1 = rng.random()
                if style1 < 0.4:
                    c1[:] = vec()
                elif style1 < 0.7:
                    for q in range(4):
                        c1[(q >> 1) * 2:(q >> 1) * 2 + 2, (q & 1) * 2:(q & 1) * 2 + 2] = vec()
                else:
                    c1 = rng.integers(-mv_range, mv_range + 1, size=(4, 4, 2))
                if rng.random() < mirror_l1:
                    c1 = cells.copy()
                if skip:                            # (B_Skip / direct: whatever the derivation gave - here: both lists, one vector each)
                    dirs = [2] * 4
                for q in range(4):
                    qy, qx = (q >> 1) * 2, (q & 1) * 2
                    if dirs[q] == 1:
                        ref[m, q] = -1
                        mv[m].reshape(4, 4, 2)[qy:qy + 2, qx:qx + 2] = 0
                    if dirs[q] == 0:
                        c1[qy:qy + 2, qx:qx + 2] = 0
                    else:
                        ref1[m, q] = int(rng.integers(0, n_ref_l1))
                mv1[m] = c1.reshape(16, 2)
            cbp_l = 0 if skip else (int(rng.integers(0, 16)) if rng.random() < 0.6 else 0)
            for b in range(16):
                if (cbp_l >> (b >> 2)) & 1 and rng.random() < 0.6:
                    mask |= 1 << b
            n_lv = 16
        # chroma
        cc = 0 if (not intra and r["mb_type"] == N.MB_P_SKIP) else int(rng.choice([0, 0, 1, 2]))
        if cc and rng.random() < 0.8:
            mask |= N.COEF_CHROMA_DC
            dc = np.zeros(16, np.int16)
            dc[:8] = np.concatenate([_levels(rng, 4, style)[:4], _levels(rng, 4, style)[:4]])
            mb_blocks.append(("cdc", dc))
        if cc == 2:
            for b in range(16, 24):
                if rng.random() < 0.5:
                    mask |= 1 << b
        for b in range(24):
            if (mask >> b) & 1:
                nl = 15 if (b >= 16 or (intra and r["mb_type"] == N.MB_I16x16)) else n_lv
                mb_blocks.append((b, _levels(rng, nl, style)))
        r["cbp"] = (cbp_l if (intra or not (r["mb_type"] == N.MB_P_SKIP)) else 0) | (cc << 4)
        r["coef_mask"] = mask
        r["coef_index"] = len(blocks)
        order = {"ldc": -2, "cdc": -1}
        mb_blocks.sort(key=lambda kv: order.get(kv[0], kv[0]) if isinstance(kv[0], str) else kv[0])
        blocks += [b for _, b in mb_blocks]
    d.n_coef_blocks = len(blocks)
    if blocks:
        pic.coefs = np.concatenate(blocks).astype(np.int16)
    return pic.seal()


def random_frame(rng, mb_w, mb_h, kind="noise"):
    h, w = mb_h * 16, mb_w * 16
    if kind == "noise":
        return [rng.integers(0, 256, size=(h, w), dtype=np.uint8), rng.integers(0, 256, size=(h // 2, w // 2), dtype=np.uint8),
                rng.integers(0, 256, size=(h // 2, w // 2), dtype=np.uint8)]
    # smooth gradients + mild noise: the loop filter's |p0 - q0| < alpha conditions actually hold
    yy, xx = np.mgrid[0:h, 0:w]
    base = (xx * int(rng.integers(1, 4)) + yy * int(rng.integers(1, 4))) // 3 + int(rng.integers(0, 100))
    y = np.clip(base + rng.integers(-3, 4, size=(h, w)), 0, 255).astype(np.uint8)
    u = np.clip(base[::2, ::2] // 2 + 60 + rng.integers(-2, 3, size=(h // 2, w // 2)), 0, 255).astype(np.uint8)
    v = np.clip(200 - base[::2, ::2] // 2 + rng.integers(-2, 3, size=(h // 2, w // 2)), 0, 255).astype(np.uint8)
    return [y, u, v]
