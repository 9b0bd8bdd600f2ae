"""The reference's known answers for the intra predictors and the DC transforms (tests/golden/kat_hotpath.npz: p16_* / p8_* /
p4_* - every entry of predict_16x16[], predict_8x8c[], predict_4x4[], core/predict.c:55-638 - and ldc_* / cdc_* -
idct4x4dc + p264_mb_dequant_4x4_dc, idct2x2dc + p264_mb_dequant_2x2_dc, core/dct.c:55-68,104-136, core/quant.c:138-191, all
52 QPs and full-range int16 input) driven through the HIP kernels themselves, not through the oracle: every case is a small
picture at the CPU->GPU seam (tests/kat_seam.py) and every expected sample is the reference's recorded output.

All cases of a family go through ONE p264hip_reconstruct call (a stream per case / per group of cases).  `launch`:
  sparse - a batch of P pictures: k_intra_sparse (the two-round free lists, then the band walk);
  dense  - one more stream with an I picture in the same call: the whole batch goes through k_intra, the dense build."""
import numpy as np
import pytest

from p264decoder_amd import HipReconstructor, _native as N
from tests import kat_seam as K

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kat():
    return K.hotpath()


def run_batch(lib, jobs, launch):
    """jobs: [(picture, reference frame)] of one picture size.  Returns the reconstructed [y, u, v] per job."""
    pic0 = jobs[0][0]
    S = len(jobs) + (1 if launch == "dense" else 0)
    hip = HipReconstructor(pic0.mb_w, pic0.mb_h, n_streams=S, slots=2, max_pictures=S, lib=lib)
    keep = []
    for s, (pic, ref) in enumerate(jobs):
        hip.write_frame(s, 1, *ref)
        hip.upload(s, [pic])
    if launch == "dense":
        ip = K.base_picture(pic0.mb_w, pic0.mb_h)
        ip.desc.slice_type, ip.desc.n_ref = N.SLICE_I, 0
        ip.rec["mb_type"], ip.rec["intra_modes"] = N.MB_I16x16, 2
        ip.ref_idx[:] = -1
        keep.append(K.set_blocks(ip, [[] for _ in range(ip.n_mb)]).seal())
        hip.upload(S - 1, [keep[0]])
    hip.reconstruct(list(range(S)), list(range(S)))
    hip.sync()
    assert (hip.last_launch()["edge_info_fused"] > 0) == (launch != "dense")     # (k_intra_sparse carries the edge-info role; k_intra does not)
    out = [hip.read_frame(s, 0) for s in range(len(jobs))]
    hip.close()
    return out


@pytest.mark.parametrize("launch", ["sparse", "dense"])
def test_intra_16x16_predictors_on_reference_vectors(lib, kat, launch):
    modes = kat["p16_mode"]
    jobs = [K.pred16_case(kat["p16_in"][i], modes[i], i) for i in range(len(modes))]
    assert sorted(set(modes.tolist())) == list(range(7))
    for i, ((y, u, v), (pic, ref)) in enumerate(zip(run_batch(lib, jobs, launch), jobs)):
        assert np.array_equal(y[16:32, 16:32], kat["p16_out"][i][1:17, 1:17]), "predict_16x16[%d] case %d" % (modes[i], i)
        assert np.array_equal(y[:16], ref[0][:16]) and np.array_equal(y[16:, :16], ref[0][16:, :16])      # the neighbours are what they were


@pytest.mark.parametrize("launch", ["sparse", "dense"])
def test_intra_chroma_predictors_on_reference_vectors(lib, kat, launch):
    modes = kat["p8_mode"]
    pairs = list(range(0, len(modes), 2))                            # two cases of one mode per macroblock: Cb and Cr
    assert all(modes[i] == modes[i + 1] for i in pairs) and sorted(set(modes.tolist())) == list(range(7))
    jobs = [K.pred8_case(kat["p8_in"][i], kat["p8_in"][i + 1], modes[i], i) for i in pairs]
    for i, (y, u, v) in zip(pairs, run_batch(lib, jobs, launch)):
        assert np.array_equal(u[8:16, 8:16], kat["p8_out"][i][1:9, 1:9]), "predict_8x8c[%d] case %d (Cb)" % (modes[i], i)
        assert np.array_equal(v[8:16, 8:16], kat["p8_out"][i + 1][1:9, 1:9]), "predict_8x8c[%d] case %d (Cr)" % (modes[i], i + 1)


@pytest.mark.parametrize("launch", ["sparse", "dense"])
def test_intra_4x4_predictors_on_reference_vectors(lib, kat, launch):
    modes = kat["p4_mode"]
    assert sorted(set(modes.tolist())) == list(range(12))
    jobs = [K.pred4_case(kat["p4_in"][i], modes[i], i) for i in range(len(modes))]
    for i, (y, u, v) in enumerate(run_batch(lib, jobs, launch)):
        assert np.array_equal(y[16:20, 16:20], kat["p4_out"][i][1:5, 1:5]), "predict_4x4[%d] case %d" % (modes[i], i)


@pytest.mark.parametrize("launch", ["sparse", "dense"])
def test_luma_dc_transform_and_dequant_on_reference_vectors(lib, kat, launch):
    """idct4x4dc + p264_mb_dequant_4x4_dc: 400 cases (all QPs; the last 100 with full-range int16 input, A-Q8), twenty per
    picture, at three prediction values: 16 dequantised DC values per case, each seen through clip(pred + ((d + 32) >> 6))."""
    n, W = len(kat["ldc_qp"]), 20
    groups = [list(range(f, min(f + W, n))) for f in range(0, n, W)]
    seen = set()
    for pred in (0, 128, 255):
        jobs = [K.luma_dc_picture(g, kat["ldc_in"], kat["ldc_qp"], pred) for g in groups]
        for g, (y, u, v) in zip(groups, run_batch(lib, jobs, launch)):
            for k, i in enumerate(g):
                want = np.kron(K.dc_only(pred, kat["ldc_out"][i]).reshape(4, 4), np.ones((4, 4), np.uint8))
                assert np.array_equal(y[16:32, k * 16:k * 16 + 16], want), "luma DC case %d (QP %d) on prediction %d" % (i, kat["ldc_qp"][i], pred)
                seen.update(np.unique(want).tolist())
    assert len(seen) > 200                                            # the cases do not all saturate


@pytest.mark.parametrize("road", ["inter", "intra-sparse", "intra-dense"])
def test_chroma_dc_transform_and_dequant_on_reference_vectors(lib, kat, road):
    """idct2x2dc + p264_mb_dequant_2x2_dc (truncating shift below QP_C 30, core/quant.c:153): in inter macroblocks (the chroma
    roles of k_mc) and in intra macroblocks (k_intra_sparse / k_intra).  A luma QP reaches chroma QPs up to 39 only
    (core/macroblock.h:210-218): the cases recorded at 40 ... 51 are counted and left to the oracle's replay."""
    cases = [i for i in range(len(kat["cdc_qp"])) if int(kat["cdc_qp"][i]) in K.LUMA_QP_FOR_CHROMA]
    assert len(cases) >= 300 and {int(kat["cdc_qp"][i]) for i in cases} == set(range(40))
    W = 20
    groups = [(cases[f:f + W] + cases[:W])[:W] for f in range(0, len(cases), W)]      # (one picture width: the last group is filled up)
    for pred in (0, 128, 255):
        jobs = [K.chroma_dc_picture(g, kat["cdc_in"], kat["cdc_qp"], pred, road != "inter") for g in groups]
        for g, (y, u, v) in zip(groups, run_batch(lib, jobs, "dense" if road == "intra-dense" else "sparse")):
            for k, i in enumerate(g):
                want = np.kron(K.dc_only(pred, kat["cdc_out"][i]).reshape(2, 2), np.ones((4, 4), np.uint8))
                for name, plane in (("Cb", u), ("Cr", v)):
                    assert np.array_equal(plane[8:16, k * 8:k * 8 + 8], want), "chroma DC case %d (QP_C %d) %s on prediction %d" % (i, kat["cdc_qp"][i], name, pred)
