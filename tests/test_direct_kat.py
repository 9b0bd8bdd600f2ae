"""B-picture derivations of the host parser (csrc/host/parser.c: implicit_weights, direct_spatial, direct_temporal) against
known answers recorded from the REFERENCE's own functions - p264_macroblock_bipred_init (core/macroblock.c:1400-1430) and
p264_mb_predict_mv_direct16x16 (core/macroblock.c:254-413), encoder-side code its decoder never reaches but which
oracle/ref_kat.c can call (tests/golden/make_kat_direct.py -> tests/golden/kat_direct.npz).  SURVEY 8c names these as the
partial pins the reference still offers for BASELINE configs 4-5.

Where the reference deviates from H.264 the parser follows the standard; every such case is singled out below by its
cause, counted, and everything else must match bit for bit:
  * weights: a pair of references with EQUAL picture order counts gets the default weights (32, 32) - 8.4.2.3.1 takes the
    implicit formula only when DiffPicOrderCnt(picA, picB) != 0; the reference runs its formula on a scale factor of 256
    and ends up with weight 0 for list 0 (core/macroblock.c:1413-1427);
  * temporal direct: the reference gives up (returns 0, "not available" - fine for an encoder choosing a mode) when the
    co-located block has no list-0 motion or its reference picture is not in the current list 0 (core/macroblock.c:276,
    298-306); a decoder has to derive something: 8.4.1.2.3 takes the list-1 motion of the co-located block then."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KAT = np.load(os.path.join(ROOT, "tests", "golden", "kat_direct.npz"))


def P(a):
    return a.ctypes.data_as(C.c_void_p)


def test_implicit_weights_match_the_reference(lib):
    lib.p264parse_kat_bipred.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    n = len(KAT["bw_cur"])
    equal_poc = checked = 0
    for i in range(n):
        n0, n1 = int(KAT["bw_n0"][i]), int(KAT["bw_n1"][i])
        poc0, poc1 = np.ascontiguousarray(KAT["bw_poc0"][i]), np.ascontiguousarray(KAT["bw_poc1"][i])
        w = np.zeros(256, np.int16)
        assert lib.p264parse_kat_bipred(n0, P(poc0), n1, P(poc1), int(KAT["bw_cur"][i]), P(w)) == 0
        ref_w = KAT["bw_w"][i].reshape(16, 16)
        for r0 in range(n0):
            for r1 in range(n1):
                if poc0[r0] == poc1[r1]:
                    equal_poc += 1
                    assert w[r0 * 16 + r1] == 32 and ref_w[r0, r1] == 0      # the deviation named above, and nothing else
                else:
                    checked += 1
                    assert w[r0 * 16 + r1] == ref_w[r0, r1], (i, r0, r1, int(w[r0 * 16 + r1]), int(ref_w[r0, r1]))
    assert checked > 1000 and equal_poc > 20
    # weights really vary, and pairs outside the range fall back to the plain average
    allw = KAT["bw_w"].reshape(-1)
    assert len(set(allw.tolist())) > 40


def run_direct(lib, i):
    lib.p264parse_kat_direct.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                         C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    k = {name[2:]: KAT[name][i] for name in KAT.files if name.startswith("d_")}
    k = {a: (np.ascontiguousarray(v) if np.ndim(v) else int(v)) for a, v in k.items()}
    o_ref, o_mv = np.zeros((2, 4), np.int8), np.zeros((2, 16, 2), np.int16)
    rc = lib.p264parse_kat_direct(int(k["spatial"]), P(k["nb_ref"]), P(k["nb_mv"]), int(k["col_intra"]), P(k["col_ref"]), P(k["col_mv"]),
                                  int(k["n0"]), P(k["poc0"]), int(k["poc1_0"]), int(k["cur"]), int(k["n_col"]), P(k["col_poc"]), P(o_ref), P(o_mv))
    assert rc == 0
    return k, o_ref, o_mv


def used_mv(ref, mv):
    """vectors with those of unused lists (index < 0) blanked: the reference leaves whatever its cache held there"""
    out = mv.copy()
    for l in range(2):
        for q in range(4):
            if ref[l, q] < 0:
                for b in range(4):
                    out[l, (q >> 1) * 8 + (q & 1) * 2 + (b >> 1) * 4 + (b & 1)] = 0
    return out


def test_spatial_direct_matches_the_reference(lib):
    n = len(KAT["d_ok"])
    seen = dict(cases=0, zero_pred=0, col_zero=0, one_list=0, both=0, moved=0)
    for i in range(0, n, 2):
        k, o_ref, o_mv = run_direct(lib, i)
        assert int(k["spatial"]) == 1 and int(k["ok"]) == 1
        assert np.array_equal(o_ref, k["out_ref"]), (i, o_ref.tolist(), k["out_ref"].tolist())
        assert np.array_equal(used_mv(o_ref, o_mv), used_mv(k["out_ref"], k["out_mv"])), i
        seen["cases"] += 1
        nb = k["nb_ref"]
        seen["zero_pred"] += int((nb[:, :3] < 0).all() and (nb[:, 2] != -2).all() or (nb < 0).all())
        seen["one_list"] += int((o_ref[0] < 0).all() != (o_ref[1] < 0).all())
        seen["both"] += int((o_ref >= 0).all())
        seen["moved"] += int(np.abs(o_mv).max() > 0)
        # a quadrant whose vectors were zeroed by colZeroFlag: the macroblock's predicted vector is not zero, some block's is
        for l in range(2):
            if (o_ref[l] == 0).all() and np.abs(o_mv[l]).max() > 0 and (np.abs(o_mv[l]).sum(axis=1) == 0).any():
                seen["col_zero"] += 1
    assert seen["cases"] == n // 2 and seen["zero_pred"] >= 5 and min(v for k_, v in seen.items() if k_ != "zero_pred") > 50, seen


def test_temporal_direct_matches_the_reference_where_it_answers(lib):
    n = len(KAT["d_ok"])
    answered = gave_up = scaled = intra = 0
    for i in range(1, n, 2):
        k, o_ref, o_mv = run_direct(lib, i)
        assert int(k["spatial"]) == 0
        if not int(k["ok"]):
            # the reference's "not available": some quadrant of the co-located macroblock has no list-0 motion, or the picture it
            # refers to is not in the current list 0 - exactly those cases, nothing else
            col0 = k["col_ref"][0]
            assert not int(k["col_intra"]) and any(c < 0 or k["map_col"][c] < 0 for c in col0.tolist()), i
            gave_up += 1
            assert (o_ref[1] == 0).all() and (o_ref[0] >= 0).all()               # ours still derives a prediction (8.4.1.2.3)
            continue
        answered += 1
        assert np.array_equal(o_ref, k["out_ref"]), (i, o_ref.tolist(), k["out_ref"].tolist())
        assert np.array_equal(o_mv, k["out_mv"]), i
        intra += int(k["col_intra"])
        scaled += int(not int(k["col_intra"]) and np.abs(k["col_mv"][0]).max() > 8 and not np.array_equal(o_mv[0], k["col_mv"][0]))
    assert answered > 150 and gave_up > 150 and scaled > 80 and intra > 10, (answered, gave_up, scaled, intra)
