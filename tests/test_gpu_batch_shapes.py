"""The two row-wavefront kernels pick their workgroup shape from the batch size (pictures vs compute units):
k_deblock walks bands of 8 / 4 / 2 macroblock rows of 1 / 2 / 4 pictures per wavefront, a workgroup serving up to 16 pictures
in groups of as many as a wavefront holds (work units = band x group); k_intra uses 16 or 8 wavefronts per picture.  A test batch is far smaller than an MI355X, so every shape is forced through the
P264AMD_* knobs (read once, when the context is created) and checked against the CPU oracle, picture by picture, on every stream."""
import os

import numpy as np
import pytest

from p264decoder_amd import HipReconstructor, Parser
from tests import oracle_bind, synth_cases

pytestmark = pytest.mark.gpu

SHAPES = [  # (P264AMD_DEBLOCK_RB_LOG2, P264AMD_DEBLOCK_PICS_PER_WG, P264AMD_INTRA_WAVES)
    ("3", "1", "16"), ("2", "2", "8"), ("2", "1", "4"), ("1", "4", "8"), ("1", "3", "1"), ("1", "1", "16"),
    # more pictures per workgroup than a wavefront holds: groups (the last one partly empty with 7 streams)
    ("2", "4", "8"), ("3", "4", "8"), ("2", "7", "4"), ("3", "16", "4"), ("1", "13", "2"),
]


@pytest.mark.parametrize("rb,per_wg,intra_waves", SHAPES)
@pytest.mark.parametrize("case", ["cif_ip", "col_Nx1", "wide_70"])
def test_workgroup_shapes(lib, oracle, case, rb, per_wg, intra_waves, monkeypatch):
    monkeypatch.setenv("P264AMD_DEBLOCK_RB_LOG2", rb)
    monkeypatch.setenv("P264AMD_DEBLOCK_PICS_PER_WG", per_wg)
    monkeypatch.setenv("P264AMD_INTRA_WAVES", intra_waves)
    parser = Parser(quiet=True, lib=lib)
    pics = parser.parse_stream(synth_cases.stream_bytes(case))[:8]
    mb_w, mb_h = pics[0].mb_w, pics[0].mb_h
    S = 7                                                      # odd on purpose: the last workgroup is partly empty
    store = oracle_bind.FrameStore(mb_w, mb_h, parser.slots)
    hip = HipReconstructor(mb_w, mb_h, n_streams=S, slots=parser.slots, max_pictures=len(pics), lib=lib)
    hip.upload(0, pics)
    for i, p in enumerate(pics):
        want = oracle_bind.reconstruct(oracle, store, p)
        hip.reconstruct([i] * S, list(range(S)))
        for s in range(S):
            got = hip.read_frame(s, p.desc.dst_slot)
            for plane, (a, b) in enumerate(zip(got, want)):
                assert np.array_equal(a, b), "%s shape (%s,%s,%s): picture %d stream %d plane %d differs" % (case, rb, per_wg, intra_waves, i, s, plane)
    hip.close()


@pytest.mark.parametrize("per_wg,S", [("3", 7), ("5", 11), ("7", 7), ("13", 14)])
@pytest.mark.parametrize("case", ["cif_ip", "col_Nx1", "wide_70", "dense", "qpdelta"])
def test_odd_picture_counts_as_pairs_and_a_single(lib, oracle, case, per_wg, S, monkeypatch):
    """An odd number of pictures per k_deblock workgroup (what a batch of 3 / 5 / 7 pictures per compute unit selects by itself):
    the pictures in pairs through bands of 4 rows and the last one alone through bands of 8 (odd_single) - unit order, band
    numbering and progress counters of two shapes in one workgroup; picture heights with an odd and an even number of 4-row
    bands; the last workgroup partly empty."""
    monkeypatch.setenv("P264AMD_DEBLOCK_RB_LOG2", "2")
    monkeypatch.setenv("P264AMD_DEBLOCK_PICS_PER_WG", per_wg)
    monkeypatch.setenv("P264AMD_DEBLOCK_ODD_SINGLE", "1")
    parser = Parser(quiet=True, lib=lib)
    pics = parser.parse_stream(synth_cases.stream_bytes(case))[:6]
    mb_w, mb_h = pics[0].mb_w, pics[0].mb_h
    store = oracle_bind.FrameStore(mb_w, mb_h, parser.slots)
    hip = HipReconstructor(mb_w, mb_h, n_streams=S, slots=parser.slots, max_pictures=len(pics), lib=lib)
    hip.upload(0, pics)
    for i, p in enumerate(pics):
        want = oracle_bind.reconstruct(oracle, store, p)
        hip.reconstruct([i] * S, list(range(S)))
        li = hip.last_launch()
        assert li["deblock_odd_single"] == 1 and li["deblock_rb_log2"] == 2 and li["deblock_pics_per_wg"] == int(per_wg), li
        for s in range(S):
            got = hip.read_frame(s, p.desc.dst_slot)
            for plane, (a, b) in enumerate(zip(got, want)):
                assert np.array_equal(a, b), "%s, %s pictures per workgroup: picture %d stream %d plane %d differs" % (case, per_wg, i, s, plane)
    hip.close()


def test_odd_picture_counts_at_1080p(lib, oracle, monkeypatch):
    """The same at the bench's picture size: three and five 1080p pictures in one workgroup (17 bands of 4 rows per pair, 9 bands
    of 8 rows for the single picture: more units than wavefronts)."""
    monkeypatch.setenv("P264AMD_DEBLOCK_RB_LOG2", "2")
    monkeypatch.setenv("P264AMD_DEBLOCK_ODD_SINGLE", "1")
    for per_wg, S in (("3", 3), ("5", 6)):
        monkeypatch.setenv("P264AMD_DEBLOCK_PICS_PER_WG", per_wg)
        parser = Parser(quiet=True, lib=lib)
        pics = parser.parse_stream(synth_cases.stream_bytes("cfg3_1080p_allp"))[:3]
        mb_w, mb_h = pics[0].mb_w, pics[0].mb_h
        store = oracle_bind.FrameStore(mb_w, mb_h, parser.slots)
        hip = HipReconstructor(mb_w, mb_h, n_streams=S, slots=parser.slots, max_pictures=len(pics), lib=lib)
        hip.upload(0, pics)
        for i, p in enumerate(pics):
            want = oracle_bind.reconstruct(oracle, store, p)
            hip.reconstruct([i] * S, list(range(S)))
            assert hip.last_launch()["deblock_odd_single"] == 1
            for s in range(S):
                got = hip.read_frame(s, p.desc.dst_slot)
                for plane, (a, b) in enumerate(zip(got, want)):
                    assert np.array_equal(a, b), "%s pictures per workgroup: picture %d stream %d plane %d differs" % (per_wg, i, s, plane)
        hip.close()


@pytest.mark.parametrize("fused", ["0", "1", "3", "16"])
@pytest.mark.parametrize("case", ["cif_ip", "qpdelta", "wide_70"])
def test_edge_info_inside_the_intra_launch_or_on_its_own(lib, oracle, case, fused, monkeypatch):
    """The loop filter's edge-info pass: its own launch (P264AMD_BS_FUSED=0: k_deblock_bs, as batches with I or B pictures always
    take it) or 1 / 3 / 16 extra workgroups per picture of the k_intra_sparse launch (batches of P pictures; the default is 1)."""
    monkeypatch.setenv("P264AMD_BS_FUSED", fused)
    parser = Parser(quiet=True, lib=lib)
    pics = parser.parse_stream(synth_cases.stream_bytes(case))[:8]
    mb_w, mb_h = pics[0].mb_w, pics[0].mb_h
    S = 5
    store = oracle_bind.FrameStore(mb_w, mb_h, parser.slots)
    hip = HipReconstructor(mb_w, mb_h, n_streams=S, slots=parser.slots, max_pictures=len(pics), lib=lib)
    hip.upload(0, pics)
    for i, p in enumerate(pics):
        want = oracle_bind.reconstruct(oracle, store, p)
        hip.reconstruct([i] * S, list(range(S)))
        for s in (0, S - 1):
            got = hip.read_frame(s, p.desc.dst_slot)
            for plane, (a, b) in enumerate(zip(got, want)):
                assert np.array_equal(a, b), "%s, P264AMD_BS_FUSED=%s: picture %d stream %d plane %d differs" % (case, fused, i, s, plane)
    hip.close()


@pytest.mark.parametrize("band_log2,wgs", [("0", "4"), ("2", "7"), ("6", "200"), ("1", "16")])
def test_mc_launch_knobs(lib, oracle, band_log2, wgs, monkeypatch):
    """The other launch paths no default run takes: locality bands of the motion-compensation lists of 1 / 4 / 64 macroblock
    rows (more or fewer keys and chunks), few or many workgroups per picture in the fused launch (down to one workgroup per
    role)."""
    monkeypatch.setenv("P264AMD_MC_BAND_LOG2", band_log2)
    monkeypatch.setenv("P264AMD_MC_WGS_PER_PIC", wgs)
    parser = Parser(quiet=True, lib=lib)
    pics = parser.parse_stream(synth_cases.stream_bytes("cif_ip"))[:10]
    mb_w, mb_h = pics[0].mb_w, pics[0].mb_h
    S = 5
    store = oracle_bind.FrameStore(mb_w, mb_h, parser.slots)
    hip = HipReconstructor(mb_w, mb_h, n_streams=S, slots=parser.slots, max_pictures=len(pics), lib=lib)
    hip.upload(0, pics)
    for i, p in enumerate(pics):
        want = oracle_bind.reconstruct(oracle, store, p)
        hip.reconstruct([i] * S, list(range(S)))
        for s in (0, S - 1):
            got = hip.read_frame(s, p.desc.dst_slot)
            for plane, (a, b) in enumerate(zip(got, want)):
                assert np.array_equal(a, b), "band %s wgs %s: picture %d stream %d plane %d differs" % (band_log2, wgs, i, s, plane)
    hip.close()


def test_vectors_far_outside_the_picture(lib, oracle):
    """Windows that lie entirely outside the picture (vectors of +-100 pixels on a 160x128 picture): beyond the
    reference's 32-pixel pads its behaviour is undefined (SURVEY A-Q9), so this is HIP against the oracle's clamped
    coordinates only - it exercises the replicated-border dwords of the motion-compensation kernel on all four sides."""
    parser = Parser(quiet=True, lib=lib)
    pics = parser.parse_stream(open(synth_cases.generate("--mbw 10 --mbh 8 --frames 6 --gop 6 --seed 29 --mvmax 400 --coded 5 --maxlevel 6"), "rb").read())
    mb_w, mb_h = pics[0].mb_w, pics[0].mb_h
    store = oracle_bind.FrameStore(mb_w, mb_h, parser.slots)
    hip = HipReconstructor(mb_w, mb_h, n_streams=1, slots=parser.slots, max_pictures=1, lib=lib)
    for i, p in enumerate(pics):
        want = oracle_bind.reconstruct(oracle, store, p)
        hip.submit(0, p)
        got = hip.read_frame(0, p.desc.dst_slot)
        for plane, (a, b) in enumerate(zip(got, want)):
            assert np.array_equal(a, b), "picture %d plane %d differs" % (i, plane)
    hip.close()
