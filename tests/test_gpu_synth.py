"""GPU parity on the synthetic workloads, full length: BASELINE config 2 (1280x720 I-only, 30
pictures), config 3 (1920x1088 I+P, 60 pictures; and the all-P throughput variant) and the edge
cases.  For every picture: HIP output (through the C ABI) == CPU oracle on the same parsed
buffers, byte for byte, and == the committed SHA-256 of the real reference decoder."""
import numpy as np
import pytest

from p264decoder_amd import Decoder, HipReconstructor, Parser
from tests import oracle_bind, synth_cases
from tests.conftest import frame_sha256

pytestmark = pytest.mark.gpu


def run_case(lib, oracle, name, with_oracle=True):
    _, hashes = synth_cases.golden(name)
    parser = Parser(quiet=True, lib=lib)
    pics = parser.parse_stream(synth_cases.stream_bytes(name))
    assert len(pics) == len(hashes)
    mb_w, mb_h = pics[0].mb_w, pics[0].mb_h
    store = oracle_bind.FrameStore(mb_w, mb_h, parser.slots) if with_oracle else None
    hip = HipReconstructor(mb_w, mb_h, n_streams=1, slots=parser.slots, max_pictures=1, lib=lib)
    for i, p in enumerate(pics):
        hip.submit(0, p)
        got = hip.read_frame(0, p.desc.dst_slot)
        if with_oracle:
            want = oracle_bind.reconstruct(oracle, store, p)
            for plane, (a, b) in enumerate(zip(got, want)):
                if not np.array_equal(a, b):
                    ys, xs = np.nonzero(a != b)
                    s = 16 if plane == 0 else 8
                    pytest.fail("%s picture %d plane %d: %d samples differ from the oracle, first (y=%d,x=%d) MB (%d,%d) type %d" % (
                        name, i, plane, len(ys), ys[0], xs[0], ys[0] // s, xs[0] // s,
                        p.mb_records()["mb_type"][(ys[0] // s) * mb_w + xs[0] // s]))
        assert frame_sha256(*got) == hashes[i], "%s picture %d differs from the reference decoder" % (name, i)
    hip.close()


@pytest.mark.parametrize("name", [n for n in synth_cases.CASES if n not in synth_cases.BIG])
def test_edge_cases(lib, oracle, name):
    run_case(lib, oracle, name)


def test_config2_720p_intra(lib, oracle):
    run_case(lib, oracle, "cfg2_720p_intra")


def test_config3_1080p_ip(lib, oracle):
    run_case(lib, oracle, "cfg3_1080p_ip", with_oracle=False)      # 60 pictures: reference hashes only


def test_config3_1080p_ip_levels_32(lib, oracle):
    """config 3 with levels up to +-32 (SURVEY 8d's range) on a stream that stays inside the reference's clip table"""
    run_case(lib, oracle, "cfg3_1080p_ip_l32", with_oracle=False)


def test_config3_1080p_allp_vs_oracle(lib, oracle):
    run_case(lib, oracle, "cfg3_1080p_allp")


def test_config3_1080p_allp_300_pictures(lib, oracle):
    """The throughput stream at the length SURVEY 8d gives it - 1 IDR + 299 P pictures - every picture against the real
    reference decoder's hash."""
    run_case(lib, oracle, "cfg3_1080p_allp_300", with_oracle=False)


def test_per_macroblock_qp_1080p(lib, oracle):
    """mb_qp_delta on every coded macroblock plus non-zero deblocking offsets at 1080p (A-Q2, A-Q3), reference-pinned"""
    run_case(lib, oracle, "qpd_1080p")


def test_2160p_all_p(lib, oracle):
    """3840x2160: more macroblocks than the work-list sort keeps in registers (it classifies twice instead), nine bands of
    work lists, 135 macroblock rows for the two row-wavefront kernels - against the oracle and the reference's hashes"""
    run_case(lib, oracle, "uhd_2160p_allp")


def test_2160p_batch_of_streams(lib):
    """the same pictures as a batch of 5 streams in one call (two-picture deblocking workgroups at this size)"""
    _, hashes = synth_cases.golden("uhd_2160p_allp")
    parser = Parser(quiet=True, lib=lib)
    pics = parser.parse_stream(synth_cases.stream_bytes("uhd_2160p_allp"))
    S = 5
    hip = HipReconstructor(pics[0].mb_w, pics[0].mb_h, n_streams=S, slots=parser.slots, max_pictures=S, lib=lib)
    for i, p in enumerate(pics):
        hip.upload(0, [p])
        for s in range(1, S):
            hip.clone_picture(s, 0)
        hip.reconstruct(list(range(S)), list(range(S)))
        hip.sync()
        for s in range(S):
            assert frame_sha256(*hip.read_frame(s, p.desc.dst_slot)) == hashes[i], "picture %d stream %d differs from the reference decoder" % (i, s)
    hip.close()


def test_bench_shape_batch_against_reference_hashes(lib):
    """More pictures than 2 x compute units in ONE reconstruct call (that selects the small-band workgroup shapes of the
    row-wavefront kernels: 4 intra wavefronts, bands of 4 rows and 3 pictures per deblocking workgroup on a 256-CU device, the
    last workgroup partly empty), 1080p, every stream a private clone of the config-3 all-P pictures.  Every stream's pictures
    must hash to what the real reference decoder produced for that stream (tests/golden/synth_cfg3_1080p_allp.sha256).  The
    batch bench.py itself times is test_the_bench_s_own_batch_against_reference_hashes below."""
    _, hashes = synth_cases.golden("cfg3_1080p_allp")
    parser = Parser(quiet=True, lib=lib)
    pics = parser.parse_stream(synth_cases.stream_bytes("cfg3_1080p_allp"), limit=4)      # IDR + 3 P pictures
    mb_w, mb_h = pics[0].mb_w, pics[0].mb_h
    S = 2 * 256 + 3
    hip = HipReconstructor(mb_w, mb_h, n_streams=S, slots=parser.slots, max_pictures=S, lib=lib)
    for i, p in enumerate(pics):
        hip.upload(0, [p])
        for s in range(1, S):
            hip.clone_picture(s, 0)
        hip.reconstruct(list(range(S)), list(range(S)))
        hip.sync()
        for s in range(S):
            assert frame_sha256(*hip.read_frame(s, p.desc.dst_slot)) == hashes[i], "picture %d stream %d differs from the reference decoder" % (i, s)
    hip.close()


def test_the_bench_s_own_batch_against_reference_hashes(lib):
    """The shape bench.py's default run times, under a parity test: 8 x compute units streams (2048 on an MI355X) of the 1080p
    all-P stream in ONE p264hip_reconstruct per picture - IDR + 3 P pictures, every stream a private clone, EVERY stream of
    every picture hashed against what the real reference decoder produced (decoder/decoder.c:635-661 is what a picture goes
    through).  The launch must have selected the bench's shapes: 8 pictures per deblocking workgroup in bands of four rows, 48
    motion-compensation workgroups per picture, 4 intra wavefronts, the edge-info pass inside the intra launch of P batches."""
    _, hashes = synth_cases.golden("cfg3_1080p_allp")
    parser = Parser(quiet=True, lib=lib)
    pics = parser.parse_stream(synth_cases.stream_bytes("cfg3_1080p_allp"), limit=4)      # IDR + 3 P pictures
    mb_w, mb_h = pics[0].mb_w, pics[0].mb_h
    probe = HipReconstructor(mb_w, mb_h, n_streams=1, slots=parser.slots, max_pictures=1, lib=lib)
    n_cu = probe.last_launch()["compute_units"]
    probe.close()
    S = 8 * n_cu
    hip = HipReconstructor(mb_w, mb_h, n_streams=S, slots=parser.slots, max_pictures=S, lib=lib)
    for i, p in enumerate(pics):
        hip.upload(0, [p])
        for s in range(1, S):
            hip.clone_picture(s, 0)
        hip.reconstruct(list(range(S)), list(range(S)))
        hip.sync()
        li = hip.last_launch()
        assert li["pictures"] == S and li["deblock_pics_per_wg"] == 8 and li["deblock_rb_log2"] == 2 and li["deblock_wgs"] == n_cu and li["intra_waves"] == 4, li
        if p.desc.slice_type == 0:                     # P picture: the inter launch and the fused edge-info pass
            assert li["mc_wgs_per_picture"] == 48 and li["edge_info_fused"] == 1, li
        for s in range(S):
            assert frame_sha256(*hip.read_frame(s, p.desc.dst_slot)) == hashes[i], "picture %d stream %d differs from the reference decoder" % (i, s)
    hip.close()


def test_config3_through_dropin_api(lib):
    _, hashes = synth_cases.golden("cfg3_1080p_ip")
    dec = Decoder(lib=lib)
    n = 0
    for y, u, v in dec.decode_annexb(synth_cases.stream_bytes("cfg3_1080p_ip")):
        assert y.shape == (1088, 1920)          # MB-aligned: the crop in the SPS is not applied (A-Q1)
        assert frame_sha256(y, u, v) == hashes[n], "picture %d" % n
        n += 1
        if n == 12:
            break
    dec.close()
