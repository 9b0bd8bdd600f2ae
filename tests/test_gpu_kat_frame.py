"""The loop filter against the REFERENCE'S OWN frame driver (tests/golden/kat_deblock_frame.npz, recorded by
tests/golden/make_kat_frame.py from p264_frame_deblocking_filter, core/frame.c:490-643): boundary strengths from macroblock
type / coded blocks / reference indices / vectors, edge QPs with a QP per macroblock and a chroma offset, alpha / beta / tc0 by
table with non-zero offsets, the normal and the strong filters for luma and chroma, in raster order - 160 pictures of 4x3
macroblocks through k_deblock_bs / k_deblock (and the edge-info role of k_intra_sparse), every sample against what the
reference made of the same picture.  (kat_hotpath.npz's db_* vectors call the sample filters with parameters no table entry
and no picture can produce - see make_kat_frame.py - and stay with the oracle's replay in tests/test_oracle_kat.py.)"""
import numpy as np
import pytest

from p264decoder_amd import HipReconstructor, _native as N
from tests import kat_seam as K

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kat():
    return K.deblock_frames()


@pytest.mark.parametrize("launch", ["fused", "own", "with-i-picture", "one-by-one"])
def test_deblocking_on_the_reference_frame_drivers_answers(lib, kat, launch, monkeypatch):
    """fused: the edge info by extra workgroups of k_intra_sparse (the default for batches of P pictures); own: k_deblock_bs<false>;
    with-i-picture: an I picture in the batch (k_intra dense + k_deblock_bs); one-by-one: a launch per picture (one band, one
    picture per workgroup) instead of all 160 in one call (eight pictures per k_deblock workgroup)."""
    if launch == "own":
        monkeypatch.setenv("P264AMD_BS_FUSED", "0")
    n = len(kat["dbf_par"])
    jobs = [K.deblock_frame_case(kat, i) for i in range(n)]
    pic0 = jobs[0][0]
    S = n + (1 if launch == "with-i-picture" else 0)
    hip = HipReconstructor(pic0.mb_w, pic0.mb_h, n_streams=S, slots=3, max_pictures=S, lib=lib)
    for s, (pic, ref, want) in enumerate(jobs):
        hip.write_frame(s, 1, *ref)
        hip.write_frame(s, 2, *ref)
        hip.upload(s, [pic])
    if launch == "with-i-picture":
        ip = K.base_picture(pic0.mb_w, pic0.mb_h, deblock=1)
        ip.desc.slice_type, ip.desc.n_ref = N.SLICE_I, 0
        ip.rec["mb_type"], ip.rec["intra_modes"] = N.MB_I16x16, 2
        ip.ref_idx[:] = -1
        ip = K.set_blocks(ip, [[] for _ in range(ip.n_mb)]).seal()
        hip.upload(S - 1, [ip])
    if launch == "one-by-one":
        for s in range(S):
            hip.reconstruct([s], [s])
    else:
        hip.reconstruct(list(range(S)), list(range(S)))
    hip.sync()
    fused = hip.last_launch()["edge_info_fused"] > 0
    assert fused == (launch in ("fused", "one-by-one"))
    changed = 0
    for s, (pic, ref, want) in enumerate(jobs):
        got = hip.read_frame(s, 0)
        for name, a, b in zip("yuv", got, want):
            assert np.array_equal(a, b), "case %d plane %s: %d samples differ from the reference's filtered picture (offsets %s, QPs %s)" % (
                s, name, int((a != b).sum()), kat["dbf_par"][s].tolist(), kat["dbf_qp"][s].tolist())
        changed += int((want[0] != kat["dbf_y"][s]).sum())
    assert changed > 50000                                           # the filter did filter
    hip.close()
