"""Main-profile streams with B pictures (CAVLC; BASELINE config 4 minus its entropy coder) on the GPU: the 1080p
I + P + B workload through the HIP kernels against the committed ORACLE hashes (tests/golden/oracle_main_1080p_ipb.sha256 -
the reference cannot decode B pictures, so this pins the product to the oracle only: parity with the reference is
unpinned here), and a B stream through the drop-in API (p264_decoder_decode), whose pictures must equal the ones the
parser + p264hip path gives - decode order, like the reference's output order."""
import hashlib

import numpy as np
import pytest

from p264decoder_amd import Decoder, HipReconstructor, Parser
from tests import synth_cases
from tests.conftest import frame_sha256

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["main_1080p_ipb", "main_1080p_cabac_ipb"])
def test_main_1080p_ipb_matches_the_oracle_hashes(lib, name):
    """CAVLC, and BASELINE config 4 itself: the same pictures behind the CABAC entropy coder"""
    digest, hashes = synth_cases.oracle_golden(name)
    data = open(synth_cases.generate(synth_cases.ORACLE_CASES[name]), "rb").read()
    assert hashlib.sha256(data).hexdigest() == digest
    parser = Parser(quiet=True, lib=lib)
    pics = parser.parse_stream(data)
    assert len(pics) == len(hashes) and sum(p.desc.slice_type == 1 for p in pics) == 8
    hip = HipReconstructor(pics[0].mb_w, pics[0].mb_h, n_streams=1, slots=parser.slots, max_pictures=1, lib=lib)
    for i, p in enumerate(pics):
        hip.submit(0, p)
        assert frame_sha256(*hip.read_frame(0, p.desc.dst_slot)) == hashes[i], "picture %d (slice type %d) differs from the oracle" % (i, p.desc.slice_type)
    hip.close()


def test_b_stream_through_the_dropin_api(lib):
    args = "--mbw 9 --mbh 7 --frames 16 --seed 81 --refs 2 --bframes 2 --implicit --coded 10 --maxlevel 6"
    data = open(synth_cases.generate(args), "rb").read()
    parser = Parser(quiet=True, lib=lib)
    pics = parser.parse_stream(data)
    hip = HipReconstructor(pics[0].mb_w, pics[0].mb_h, n_streams=1, slots=parser.slots, max_pictures=1, lib=lib)
    want = []
    for p in pics:
        hip.submit(0, p)
        want.append([a.copy() for a in hip.read_frame(0, p.desc.dst_slot)])
    hip.close()
    dec = Decoder(lib=lib)
    got = [[np.array(a) for a in pic] for pic in dec.decode_annexb(data)]
    dec.close()
    assert len(got) == len(want) == 16
    for i, (g, w) in enumerate(zip(got, want)):
        for plane, (a, b) in enumerate(zip(g, w)):
            assert np.array_equal(a[:b.shape[0], :b.shape[1]], b), "picture %d plane %d" % (i, plane)
