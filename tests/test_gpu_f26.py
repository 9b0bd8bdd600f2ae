"""GPU parity on the reference's own clip (BASELINE config 1 stream through the MI355X path).

Every picture is reconstructed by the HIP kernels through the C ABI (p264hip_*), compared
byte-for-byte with the CPU oracle on the same parsed buffers, and against the committed
per-frame SHA-256 of the real reference decoder.  f26 exercises I4x4, I16x16, P_L0 (16x16 /
16x8 / 8x16), P_8x8, P_SKIP, all 16 quarter-pel phases, residuals and the loop filter."""
import numpy as np
import pytest

from p264decoder_amd import Decoder, HipReconstructor, Parser
from tests import oracle_bind
from tests.conftest import frame_sha256

pytestmark = pytest.mark.gpu


def test_f26_hip_vs_oracle_and_reference(lib, oracle, f26, f26_hashes):
    parser = Parser(quiet=True, lib=lib)
    pics = parser.parse_stream(f26)
    mb_w, mb_h = pics[0].mb_w, pics[0].mb_h
    store = oracle_bind.FrameStore(mb_w, mb_h, parser.slots)
    hip = HipReconstructor(mb_w, mb_h, n_streams=1, slots=parser.slots, max_pictures=1, lib=lib)
    for i, p in enumerate(pics):
        ref = oracle_bind.reconstruct(oracle, store, p)
        hip.submit(0, p)
        got = hip.read_frame(0, p.desc.dst_slot)
        for plane, (a, b) in enumerate(zip(got, ref)):
            if not np.array_equal(a, b):
                ys, xs = np.nonzero(a != b)
                pytest.fail("frame %d plane %d: %d samples differ, first at (y=%d,x=%d) MB (%d,%d)" % (
                    i, plane, len(ys), ys[0], xs[0], ys[0] // (16 if plane == 0 else 8), xs[0] // (16 if plane == 0 else 8)))
        assert frame_sha256(*got) == f26_hashes[i], "frame %d differs from the reference decoder" % i
    hip.close()


def test_f26_dropin_api(lib, f26, f26_hashes):
    """The same clip through p264_param_default / p264_nal_decode / p264_decoder_decode."""
    dec = Decoder(lib=lib)
    n = 0
    for y, u, v in dec.decode_annexb(f26):
        assert y.shape == (288, 352)
        assert frame_sha256(y, u, v) == f26_hashes[n], "frame %d" % n
        n += 1
    dec.close()
    assert n == 300


def test_f26_batched_streams(lib, oracle, f26, f26_hashes):
    """Several independent streams reconstructed by one launch per picture index."""
    parser = Parser(quiet=True, lib=lib)
    pics = parser.parse_stream(f26, limit=12)
    S = 5
    hip = HipReconstructor(pics[0].mb_w, pics[0].mb_h, n_streams=S, slots=parser.slots, max_pictures=len(pics), lib=lib)
    hip.upload(0, pics)
    for i in range(len(pics)):
        hip.reconstruct([i] * S, list(range(S)))
    for s in range(S):
        got = hip.read_frame(s, pics[-1].desc.dst_slot)
        assert frame_sha256(*got) == f26_hashes[len(pics) - 1]
    hip.close()
