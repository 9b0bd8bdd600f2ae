"""The CABAC macroblock layer (csrc/host/parser_cabac.h; SURVEY 8f rank 4 / BASELINE configs 4-5; the reference's own is a
stub, decoder/macroblock.c:594-597): the stream writer writes the SAME random syntax once with the CAVLC codes and once
through its own CABAC binarisations, context selection and arithmetic encoder (tools/synth264_cabac.h) - the parser must
read identical pictures out of both: macroblock records, vectors and reference indices of both lists, intra modes, every
coefficient level.  I, P and B slices, several slices per picture, sub-8x8 partitions, direct prediction, per-macroblock
QP.  (Two implementations written separately from the standard; the arithmetic ENGINE below them is pinned to the
reference's encoder by tests/test_cabac_kat.py.  Whole-stream parity with the reference stays unpinned: it decodes
neither.)  Then (GPU) the CABAC pictures through the HIP kernels against the oracle."""
import subprocess

import numpy as np
import pytest

from p264decoder_amd import Parser, _native as N
from tests import synth_cases

STREAMS = [
    "--mbw 9 --mbh 7 --frames 8 --gop 4 --seed 101 --coded 25 --maxlevel 40 --qp-delta 6",                      # I + P, large levels (escape codes)
    "--mbw 8 --mbh 6 --frames 10 --gop 0 --seed 102 --refs 2 --sub8x8 --slices 3 --coded 20 --maxlevel 12",     # two references, sub-8x8, slices
    "--mbw 11 --mbh 5 --frames 6 --intra-only --seed 103 --coded 45 --maxlevel 2000",                             # I only, dense, huge levels
    "--mbw 8 --mbh 6 --frames 16 --seed 104 --refs 2 --bframes 2 --sub8x8 --implicit --coded 12 --maxlevel 8",  # B, spatial direct
    "--mbw 7 --mbh 6 --frames 16 --seed 105 --refs 3 --bframes 3 --temporal --d8inf --slices 2 --coded 10 --maxlevel 8 --qp-delta 4",
    "--mbw 6 --mbh 5 --frames 8 --gop 0 --seed 106 --mvmax 600 --coded 8 --maxlevel 6",                          # long vectors: mvd escape codes
]


def make(tmp_path, args, tag):
    synth_cases.ensure_tool()
    stream = str(tmp_path / ("%s.264" % tag))
    subprocess.run([synth_cases.TOOL, stream] + args.split(), check=True)
    return open(stream, "rb").read()


@pytest.mark.parametrize("args", STREAMS)
def test_cabac_and_cavlc_forms_parse_to_the_same_pictures(lib, tmp_path, args):
    # (CABAC streams always get the standard's QP chain - the reference's own bookkeeping, SURVEY A-Q2, is kept for the Baseline
    # CAVLC streams it can decode; the CAVLC twin of a stream without B pictures is Baseline: parsed strictly for the comparison)
    a = Parser(quiet=True, strict=True, lib=lib).parse_stream(make(tmp_path, args, "cavlc"))
    c = Parser(quiet=True, lib=lib).parse_stream(make(tmp_path, args + " --cabac", "cabac"))
    assert len(a) == len(c) == int(args.split("--frames ")[1].split()[0])
    n_coef = 0
    for i, (p, q) in enumerate(zip(a, c)):
        assert p.desc.slice_type == q.desc.slice_type and p.desc.n_ref == q.desc.n_ref and p.desc.dst_slot == q.desc.dst_slot
        assert np.array_equal(p.mb, q.mb), "picture %d: macroblock records differ" % i
        assert np.array_equal(p.mv, q.mv) and np.array_equal(p.ref_idx, q.ref_idx), "picture %d: list-0 motion differs" % i
        assert np.array_equal(p.i4modes, q.i4modes), "picture %d: intra 4x4 modes differ" % i
        assert p.desc.n_coef_blocks == q.desc.n_coef_blocks and np.array_equal(p.coefs, q.coefs), "picture %d: levels differ" % i
        if p.desc.slice_type == N.SLICE_B:
            assert np.array_equal(p.mv_l1, q.mv_l1) and np.array_equal(p.ref_idx_l1, q.ref_idx_l1), "picture %d: list-1 motion differs" % i
            assert list(p.desc.bipred_weight) == list(q.desc.bipred_weight)
        n_coef += p.desc.n_coef_blocks
    assert n_coef > 100


def test_main_profile_streams_get_the_conformant_qp_chain(lib, tmp_path):
    """mb_qp_delta accumulates (QP_Y = (QP_Y,PRED + delta + 52) % 52, H.264 7.4.5) in every stream the reference cannot decode -
    CABAC, B slices, profiles other than Baseline - whether or not the strict option is set; Baseline CAVLC keeps the reference's
    rule (delta added to the slice QP, decoder/macroblock.c:568) unless it is."""
    qps = lambda pics: np.concatenate([p.mb_records()["qp"] for p in pics])
    for args in (STREAMS[0] + " --cabac", STREAMS[4], STREAMS[4] + " --cabac"):            # CABAC I + P; Main CAVLC with B; Main CABAC with B
        data = make(tmp_path, args, "m")
        d, s = Parser(quiet=True, lib=lib).parse_stream(data), Parser(quiet=True, strict=True, lib=lib).parse_stream(data)
        assert np.array_equal(qps(d), qps(s)), args
        assert len(set(qps(d).tolist())) > 8
    base = make(tmp_path, STREAMS[0], "b")                                                  # Baseline CAVLC, the same deltas
    d, s = Parser(quiet=True, lib=lib).parse_stream(base), Parser(quiet=True, strict=True, lib=lib).parse_stream(base)
    assert not np.array_equal(qps(d), qps(s))
    # the reference's rule keeps every QP within +-6 of the slice QP (the writer draws its deltas that way); the chain wanders
    assert int(np.abs(qps(d).astype(int) - 26).max()) <= 6 < int(np.abs(qps(s).astype(int) - 26).max())


def test_truncated_and_damaged_cabac_streams_do_not_crash(lib, tmp_path):
    import random
    data = make(tmp_path, STREAMS[3] + " --cabac", "c")
    random.seed(7)
    for trial in range(40):
        d = bytearray(data)
        for _ in range(random.randrange(1, 12)):
            d[random.randrange(30, len(d))] = random.randrange(256)
        try:
            Parser(quiet=True, lib=lib).parse_stream(bytes(d[:random.randrange(60, len(d))]))
        except Exception:
            pass


@pytest.mark.gpu
@pytest.mark.parametrize("args", [STREAMS[0], STREAMS[3], STREAMS[4]])
def test_cabac_streams_hip_vs_oracle(lib, oracle, tmp_path, args):
    from p264decoder_amd import HipReconstructor
    from tests import oracle_bind
    parser = Parser(quiet=True, lib=lib)
    pics = parser.parse_stream(make(tmp_path, args + " --cabac", "cabac"))
    store = oracle_bind.FrameStore(pics[0].mb_w, pics[0].mb_h, parser.slots)
    hip = HipReconstructor(pics[0].mb_w, pics[0].mb_h, n_streams=1, slots=parser.slots, max_pictures=1, lib=lib)
    for i, p in enumerate(pics):
        want = oracle_bind.reconstruct(oracle, store, p)
        hip.submit(0, p)
        for plane, (x, y) in enumerate(zip(hip.read_frame(0, p.desc.dst_slot), want)):
            assert np.array_equal(x, y), "picture %d plane %d differs" % (i, plane)
    hip.close()
