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


def test_mixed_slice_types_in_one_batch(lib, oracle):
    """One reconstruct call with I, P and B pictures side by side (two streams of the same Main-profile sequence, the second one
    picture behind): the B-capable sort, the second motion-compensation pass and the two-list bS derivation run for the
    whole batch as soon as ONE picture is a B picture, and must leave the others as they are."""
    from tests import oracle_bind
    args = "--mbw 11 --mbh 9 --frames 13 --seed 57 --refs 2 --bframes 2 --implicit --d8inf --coded 14 --maxlevel 10"
    parser = Parser(quiet=True, lib=lib)
    pics = parser.parse_stream(open(synth_cases.generate(args), "rb").read())
    types = [p.desc.slice_type for p in pics]
    assert {0, 1, 2} <= set(types)
    mb_w, mb_h = pics[0].mb_w, pics[0].mb_h
    store = oracle_bind.FrameStore(mb_w, mb_h, parser.slots)
    want = [[a.copy() for a in oracle_bind.reconstruct(oracle, store, p)] for p in pics]
    hip = HipReconstructor(mb_w, mb_h, n_streams=2, slots=parser.slots, max_pictures=len(pics), lib=lib)
    hip.upload(0, pics)
    mixed = 0
    for i in range(len(pics) + 1):
        ids, streams = [], []
        if i < len(pics): ids.append(i); streams.append(0)
        if i >= 1: ids.append(i - 1); streams.append(1)
        mixed += len({types[k] for k in ids}) > 1
        hip.reconstruct(ids, streams)
        for k, s in zip(ids, streams):
            got = hip.read_frame(s, pics[k].desc.dst_slot)
            for plane, (a, b) in enumerate(zip(got, want[k])):
                assert np.array_equal(a, b), "batch %d: picture %d (type %d) of stream %d, plane %d" % (i, k, types[k], s, plane)
    assert mixed >= 6
    hip.close()
