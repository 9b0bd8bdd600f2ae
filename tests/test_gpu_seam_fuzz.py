"""Seam-level fuzz (SURVEY 8b "internal CPU->GPU seam"): random p264hip_picture_t built directly, no bitstream -
p264hip_submit against the CPU oracle, byte for byte.  Covers what no decodable stream of the reference's safe subset can:
every partition shape down to 4x4 at every quarter-pel phase, a QP per macroblock, levels up to the int16 limits (storage
wrap, A-Q8), three reference frames with an index per quadrant, slice-shaped availability / edge patterns, deblocking
offsets over their whole range.  The coverage assertions check that the drawn pictures really contain those cases."""
import ctypes as C
import numpy as np
import pytest

from p264decoder_amd import HipReconstructor, _native as N
from tests import oracle_bind, seam_fuzz

pytestmark = pytest.mark.gpu

CONFIGS = [
    # name, mb_w, mb_h, pictures, make_picture keywords
    ("typical", 9, 7, 6, dict(level_style="small", qp_mode="random", n_ref=1, slots=2)),
    ("int16_wrap", 7, 5, 6, dict(level_style="wrap", qp_mode="random", n_ref=1, slots=2)),
    ("mixed_levels_3refs", 8, 6, 8, dict(level_style="mixed", qp_mode="random", n_ref=3, slots=4, slices=3)),
    ("two_qps_smooth", 10, 6, 6, dict(level_style="small", qp_mode="two", n_ref=2, slots=3, mv_range=12)),
    ("far_vectors", 6, 5, 5, dict(level_style="small", qp_mode=30, n_ref=1, slots=2, mv_range=600)),
    ("quadrant_partitions_only", 9, 6, 6, dict(level_style="large", qp_mode="random", n_ref=2, slots=3, sub8x8=False)),
    ("sliced_single_column", 1, 9, 5, dict(level_style="mixed", qp_mode="random", n_ref=1, slots=2, slices=4)),
    ("single_row", 11, 1, 5, dict(level_style="mixed", qp_mode="random", n_ref=1, slots=2, slices=3)),
    ("wide_picture", 67, 3, 4, dict(level_style="small", qp_mode="random", n_ref=2, slots=3)),
    # B pictures: two lists in the seam, every 8x8 quadrant from list 0, list 1 or both, plain and
    # implicit-weight averages (core/macroblock.c:525-583, core/mc.c:76-132).  The reference cannot decode B slices: these
    # are pinned to the oracle, whose two combines are pinned to the reference's function tables (kat_bipred.npz).
    ("b_pictures", 9, 7, 8, dict(level_style="small", qp_mode="random", n_ref=2, slots=4, b_picture=True, n_ref_l1=2)),
    ("b_pictures_far_wrap", 7, 6, 6, dict(level_style="mixed", qp_mode="random", n_ref=1, slots=3, b_picture=True, n_ref_l1=2, mv_range=500, slices=2)),
    ("b_pictures_weighted_smooth", 10, 6, 6, dict(level_style="small", qp_mode="two", n_ref=3, slots=4, b_picture=True, n_ref_l1=1, weighted=True, mv_range=12)),
    # the same two frames in both lists, in opposite order, and list-1 vectors that often repeat the list-0 vectors: neighbouring
    # blocks reach one picture through different lists (or crossed) with equal vectors - boundary strength 0 by H.264 8.7.2.1,
    # 1 by a list-by-list comparison of the indices (what core/frame.c:565-577 does)
    ("b_same_frames_swapped_lists", 10, 7, 8, dict(level_style="small", qp_mode="two", n_ref=2, slots=3, b_picture=True, n_ref_l1=2, mv_range=6, mirror_l1=0.7, sub8x8=False)),
]


def compare(got, want, what, pic):
    for plane, (a, b) in enumerate(zip(got, want)):
        if not np.array_equal(a, b):
            ys, xs = np.nonzero(a != b)
            s = 16 if plane == 0 else 8
            m = (ys[0] // s) * pic.mb_w + xs[0] // s
            r = pic.rec[m]
            pytest.fail("%s plane %d: %d samples differ, first (y=%d,x=%d) MB %d type %d qp %d mask %#x: got %d want %d" % (
                what, plane, len(ys), ys[0], xs[0], m, r["mb_type"], r["qp"], r["coef_mask"], a[ys[0], xs[0]], b[ys[0], xs[0]]))


@pytest.mark.parametrize("name,mb_w,mb_h,n_pics,kw", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_seam_fuzz(lib, oracle, name, mb_w, mb_h, n_pics, kw):
    rng = np.random.default_rng(sum(map(ord, name)) * 7919)
    slots = kw["slots"]
    store = oracle_bind.FrameStore(mb_w, mb_h, slots)
    hip = HipReconstructor(mb_w, mb_h, n_streams=1, slots=slots, max_pictures=1, lib=lib)
    for s in range(slots):
        f = seam_fuzz.random_frame(rng, mb_w, mb_h, "smooth" if "smooth" in name else "noise")
        for dst, src in zip(store[s], f):
            dst[:] = src
        hip.write_frame(0, s, *f)
    seen = dict(sub4x4=0, qp_edges=0, wrap=0, phases=set(), multi_ref=0, types=set(), avail=set(), intra4_modes=set(), weighted=set(), dirs=set(), b_roads=set())
    oracle.oracle_stats_reset()
    for i in range(n_pics):
        pic = seam_fuzz.make_picture(rng, mb_w, mb_h, p_picture=(i != 2), dst_slot=i % slots, **kw)
        want = oracle_bind.reconstruct(oracle, store, pic)
        hip.submit(0, pic)
        got = hip.read_frame(0, pic.desc.dst_slot)
        compare(got, want, "%s picture %d" % (name, i), pic)
        # ---- what did this picture exercise?
        rec = pic.rec
        n = pic.n_mb
        mv = pic.mv.reshape(n, 4, 4, 2)
        inter = rec["mb_type"] > N.MB_IPCM
        seen["types"] |= set(rec["mb_type"].tolist())
        if pic.desc.slice_type == N.SLICE_B:
            seen["weighted"].add(int(pic.desc.weighted_bipred))
            r0, r1 = pic.ref_idx.reshape(n, 4)[inter], pic.ref_idx_l1.reshape(n, 4)[inter]
            seen["dirs"] |= set(((r0 >= 0).astype(int) + 2 * (r1 >= 0).astype(int)).reshape(-1).tolist())
            # which road through the motion-compensation stage (kernel_mc.h, mc_classify): one list -> the P road; both lists
            # with one vector per quadrant and list -> two passes (whole-macroblock items, quadrant items, quadrants that the
            # second pass only carries through); vectors differing inside a quadrant -> the generic two-list class
            mv1 = pic.mv_l1.reshape(n, 4, 4, 2)
            for m, a0, a1 in zip(np.nonzero(inter)[0], r0, r1):
                if not (a1 >= 0).any():
                    seen["b_roads"].add("list0 only"); continue
                uni = True
                for q in range(4):
                    qy, qx = (q >> 1) * 2, (q & 1) * 2
                    for used, vv in ((a0[q] >= 0 or a1[q] < 0, mv[m]), (a1[q] >= 0, mv1[m])):
                        if used and len({tuple(x) for x in vv[qy:qy + 2, qx:qx + 2].reshape(4, 2).tolist()}) > 1:
                            uni = False
                if not uni:
                    seen["b_roads"].add("generic"); continue
                bi = (a0 >= 0) & (a1 >= 0)
                if not bi.any():
                    seen["b_roads"].add("list1 only")
                elif bi.all() and len({tuple(x) for x in mv1[m].reshape(16, 2).tolist()}) == 1 and len(set(a1.tolist())) == 1 and len(set(a0.tolist())) == 1:
                    seen["b_roads"].add("second pass whole")
                elif bi.all():
                    seen["b_roads"].add("second pass quadrants")
                else:
                    seen["b_roads"].add("second pass with carried quadrants")
        seen["avail"] |= set(rec["avail"].tolist())
        for m in np.nonzero(inter)[0]:
            v = mv[m]
            for qy in (0, 2):
                for qx in (0, 2):
                    q = v[qy:qy + 2, qx:qx + 2].reshape(4, 2)
                    if len({tuple(x) for x in q.tolist()}) >= 3:
                        seen["sub4x4"] += 1
            seen["phases"] |= {(int(x) & 3, int(y) & 3) for x, y in v.reshape(16, 2).tolist()}
            if len(set(pic.ref_idx.reshape(n, 4)[m].tolist())) > 1:
                seen["multi_ref"] += 1
        qp = rec["qp"].reshape(mb_h, mb_w).astype(int)
        seen["qp_edges"] += int((qp[:, 1:] != qp[:, :-1]).sum() + (qp[1:] != qp[:-1]).sum())
        if pic.desc.n_coef_blocks:
            seen["wrap"] += int((np.abs(pic.coefs.astype(np.int32)) > 16000).sum())
        i4 = pic.i4modes.reshape(n, 16)[rec["mb_type"] == N.MB_I4x4]
        seen["intra4_modes"] |= set(i4.reshape(-1).tolist())
    hip.close()
    # what the arithmetic really went through, counted by the oracle on the same inputs (oracle/cpu_recon.c, g_stats)
    st = (C.c_longlong * 8)()
    oracle.oracle_stats_get(st)
    assert st[2] > 0 and st[3] > 0 and st[4] > 0 and st[5] > 0, "the loop filter changed nothing: %s" % list(st)
    if kw["qp_mode"] in ("random", "two") and mb_w * mb_h >= 9:
        assert st[6] > 0, "no edge was filtered with the mean of two different QPs"
    if kw["level_style"] in ("wrap", "mixed"):
        assert st[7] > 0, "no dequantised coefficient wrapped its int16 store (A-Q8)"
    if kw.get("b_picture"):
        oracle.oracle_bipred_blocks.restype = C.c_longlong
        assert oracle.oracle_bipred_blocks() > 100, "hardly any bi-predicted block"
        assert seen["dirs"] == {1, 2, 3}, "not every prediction direction (list 0, list 1, both) occurred: %s" % seen["dirs"]
        assert seen["weighted"] == ({1} if kw.get("weighted") else {0, 1}), seen["weighted"]
        if kw.get("sub8x8", True):
            assert seen["b_roads"] >= {"list0 only", "list1 only", "generic", "second pass whole", "second pass quadrants", "second pass with carried quadrants"}, seen["b_roads"]
        if kw.get("mirror_l1"):
            oracle.oracle_bs_by_picture.restype = C.c_longlong
            assert oracle.oracle_bs_by_picture() > 20, "no edge segment whose strength by picture differs from the one by list index"

        assert {N.MB_I4x4, N.MB_I16x16, N.MB_B} <= seen["types"]
    else:
        assert {N.MB_I4x4, N.MB_I16x16, N.MB_P_L0, N.MB_P_8x8, N.MB_P_SKIP} <= seen["types"]
    if kw.get("sub8x8", True) and mb_w * mb_h >= 30:
        assert seen["sub4x4"] > 0, "no quadrant with three or more different vectors was drawn"
    if kw.get("mv_range", 80) >= 12 and mb_w * mb_h >= 30:
        assert len(seen["phases"]) == 16, "not every quarter-pel phase occurred"
    if kw["qp_mode"] in ("random", "two") and mb_w * mb_h >= 9:
        assert seen["qp_edges"] > 0
    if kw["level_style"] in ("wrap", "mixed"):
        assert seen["wrap"] > 0
    if kw["n_ref"] > 1:
        assert seen["multi_ref"] > 0
    if mb_w * mb_h >= 30:
        assert seen["intra4_modes"] == set(range(9))


def test_seam_fuzz_1080p_batch(lib, oracle):
    """One large picture pair through the batch entry point (several streams at once, different pictures per stream)."""
    rng = np.random.default_rng(20261004)
    mb_w, mb_h, S = 120, 68, 3
    stores = [oracle_bind.FrameStore(mb_w, mb_h, 2) for _ in range(S)]
    hip = HipReconstructor(mb_w, mb_h, n_streams=S, slots=2, max_pictures=S, lib=lib)
    for s in range(S):
        f = seam_fuzz.random_frame(rng, mb_w, mb_h, "smooth" if s == 1 else "noise")
        for slot in range(2):
            for dst, src in zip(stores[s][slot], f):
                dst[:] = src
            hip.write_frame(s, slot, *f)
    for i in range(2):
        pics = [seam_fuzz.make_picture(rng, mb_w, mb_h, p_picture=True, dst_slot=i % 2, level_style="mixed" if s == 0 else "small",
                                       qp_mode="random" if s != 1 else "two", intra_share=0.05, sub8x8=(s != 2))
                for s in range(S)]
        hip.upload(0, pics)
        hip.reconstruct(list(range(S)), list(range(S)))
        for s in range(S):
            want = oracle_bind.reconstruct(oracle, stores[s], pics[s])
            got = hip.read_frame(s, pics[s].desc.dst_slot)
            compare(got, want, "1080p picture %d stream %d" % (i, s), pics[s])
    hip.close()


@pytest.mark.parametrize("share,b_picture", [(0.02, False), (0.3, False), (0.9, False), (0.45, True)])
def test_seam_fuzz_1080p_intra_shares(lib, oracle, share, b_picture):
    """P / B pictures of 1080p with few to nearly only intra macroblocks: k_intra's lists of ready macroblocks overflow (256 per
    type and round), the second round and the ordered band walk behind it get real work (in the bench stream they see 15 % of
    the 4 % intra macroblocks)."""
    rng = np.random.default_rng(777 + int(share * 100) + 1000 * b_picture)
    mb_w, mb_h, slots = 120, 68, 3
    store = oracle_bind.FrameStore(mb_w, mb_h, slots)
    hip = HipReconstructor(mb_w, mb_h, n_streams=1, slots=slots, max_pictures=1, lib=lib)
    for s in range(slots):
        f = seam_fuzz.random_frame(rng, mb_w, mb_h, "noise")
        for dst, src in zip(store[s], f):
            dst[:] = src
        hip.write_frame(0, s, *f)
    for i in range(2):
        pic = seam_fuzz.make_picture(rng, mb_w, mb_h, p_picture=True, dst_slot=i % slots, level_style="small", qp_mode="random", n_ref=2, slots=slots,
                                     intra_share=share, b_picture=b_picture, n_ref_l1=2)
        n_intra = int((pic.rec["mb_type"] <= N.MB_IPCM).sum())
        assert abs(n_intra - share * mb_w * mb_h) < 0.25 * share * mb_w * mb_h + 40
        want = oracle_bind.reconstruct(oracle, store, pic)
        hip.submit(0, pic)
        compare(hip.read_frame(0, pic.desc.dst_slot), want, "intra share %.2f picture %d" % (share, i), pic)
    hip.close()
