"""Host side of the multi-stream pipeline (include/p264pipe.h) without a GPU: device=-1 runs the threaded parsers only.
Checks the thread pool, the per-stream picture counts (streams of different lengths), determinism against the
single-threaded parser, and the allocator hook of p264parse."""
import ctypes as C

import pytest

from p264decoder_amd import Parser, Pipeline, _native as N
from tests import synth_cases


def test_parse_only_pipeline_counts(lib, f26):
    cif = synth_cases.stream_bytes("cif_ip")
    streams = [f26, cif, f26, cif, cif]
    pipe = Pipeline(streams, threads=3, device=-1, lib=lib)
    st = pipe.run()
    assert st["pictures"] == 300 * 2 + 24 * 3 and st["streams"] == 5 and st["threads"] == 3 and st["rounds"] == 300
    assert [pipe.pictures(i) for i in range(5)] == [300, 24, 300, 24, 24]
    assert st["bytes"] == sum(len(s) for s in streams)
    pipe.close()


def test_parse_only_pipeline_limit(lib, f26):
    pipe = Pipeline([f26] * 4, threads=8, device=-1, lib=lib)          # more threads than streams: clamped
    st = pipe.run(max_pictures=7)
    assert st["pictures"] == 28 and st["threads"] == 4 and st["rounds"] == 7
    pipe.close()


def test_gpu_pipeline_needs_a_device(lib):
    if lib.p264hip_device_count() > 0:
        pytest.skip("a HIP device is present")
    assert not lib.p264pipe_open(0, 1, 1)                                # fails loudly, no CPU fallback


def test_parser_allocator_hook(lib, f26):
    """p264parse_set_allocator: every picture array comes from the caller's allocator - since round 6 as sections of one block
    per picture buffer, laid out like an input slot of the HIP layer -, output unchanged."""
    live, total = {}, [0]
    ALLOC = C.CFUNCTYPE(C.c_void_p, C.c_size_t)
    FREE = C.CFUNCTYPE(None, C.c_void_p)
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p; libc.malloc.argtypes = [C.c_size_t]; libc.free.argtypes = [C.c_void_p]

    def alloc(n):
        p = libc.malloc(n)
        live[p] = n; total[0] += 1
        return p

    def release(p):
        if p:
            assert p in live
            del live[p]
            libc.free(p)
    a, r = ALLOC(alloc), FREE(release)
    lib.p264parse_set_allocator.argtypes = [C.c_void_p, ALLOC, FREE]
    h = lib.p264parse_open(1)
    lib.p264parse_set_allocator(h, a, r)
    ref = Parser(quiet=True, lib=lib).parse_stream(f26)[:5]
    got = 0
    for t, ridc, payload in N.split_annexb(lib, f26):
        buf = (C.c_uint8 * len(payload)).from_buffer_copy(payload)
        pic = C.POINTER(N.Picture)()
        rc = lib.p264parse_nal(h, t, ridc, buf, len(payload), C.byref(pic))
        assert rc >= 0
        if rc == 1:
            n_mb = pic.contents.mb_w * pic.contents.mb_h
            base = C.addressof(pic.contents.mb.contents)
            assert base in live and base < C.addressof(pic.contents.mv.contents) < base + live[base]
            mine = C.string_at(pic.contents.mv, n_mb * 64)
            assert mine == C.string_at(ref[got].desc.mv, n_mb * 64)
            got += 1
            if got == 5:
                break
    assert total[0] >= 2                                                # 2 picture buffers, one block each
    lib.p264parse_close(h)
    assert not live                                                     # everything given back


def test_parameter_set_switches(lib):
    """(1) A stream that alternates between two PPS ids of one SPS keeps its context and its frame store (H.264: activating
    another PPS changes nothing else; P pictures still find their reference).  (2) A stream whose picture size changes
    re-initialises the context; the buffers of the picture handed out just before must survive that (the pipeline is
    still uploading from them): a re-init only retires them.  Under tests/tools/asan_host.sh both are use-after-free checks."""
    import numpy as np
    args = "--mbw 6 --mbh 5 --frames 12 --gop 6 --seed 61 --coded 20 --maxlevel 6"
    one = open(synth_cases.generate(args), "rb").read()
    alt = open(synth_cases.generate(args + " --pps-alt"), "rb").read()
    pa = Parser(quiet=True, lib=lib).parse_stream(one)
    b = Parser(quiet=True, lib=lib)
    pb = b.parse_stream(alt)
    assert len(pa) == len(pb) == 12 and lib.p264parse_generation(b.h) == 1
    for x, y in zip(pa, pb):
        assert np.array_equal(x.mb, y.mb) and np.array_equal(x.mv, y.mv) and np.array_equal(x.coefs, y.coefs)
    # picture size change in the middle of a stream
    other = open(synth_cases.generate("--mbw 8 --mbh 6 --frames 6 --gop 3 --seed 62 --coded 20 --maxlevel 6"), "rb").read()
    c = Parser(quiet=True, lib=lib)
    held = []
    for typ, idc, rbsp in N.split_annexb(lib, one + other + one):
        pic = C.POINTER(N.Picture)()
        buf = (C.c_uint8 * max(len(rbsp), 1)).from_buffer_copy(rbsp if len(rbsp) else b"\0")
        rc = lib.p264parse_nal(c.h, typ, idc, buf, len(rbsp), C.byref(pic))
        assert rc >= 0
        if held and rc == 0:                                         # the last picture's arrays are still readable and unchanged,
            ptr, n, copy = held[-1][1:]                              # also right after a re-initialisation
            assert np.array_equal(np.ctypeslib.as_array(ptr, (n,)), copy)
        if rc == 1:
            d = pic.contents
            n = d.mb_w * d.mb_h * 32
            held.append((d.mb_w, d.mv, n, np.ctypeslib.as_array(d.mv, (n,)).copy()))
    assert len(held) == 30 and lib.p264parse_generation(c.h) == 3
    assert [h[0] for h in held[10:14]] == [6, 6, 8, 8]
    ref_other = Parser(quiet=True, lib=lib).parse_stream(other)
    assert np.array_equal(held[12 + 5][3], ref_other[5].mv)
    pipe = Pipeline([alt, one, alt], threads=3, device=-1, lib=lib)
    st = pipe.run()
    assert st["pictures"] == 36
    pipe.close()


@pytest.mark.parametrize("field,value", [("log2_max_frame_num", 40), ("poc_lsb", 60), ("mb_w", 5000), ("num_ref_frames", 99), ("mb_h", 100000)])
def test_sps_fields_out_of_range_are_rejected(lib, field, value):
    """parse_sps validates the H.264 ranges before the set can be activated (shift widths, allocation sizes)."""
    from tests.tools.bitwriter import BitWriter
    w = BitWriter()
    w.u(8, 66); w.u(8, 0xc0); w.u(8, 40)
    w.ue(0)                                                         # sps id
    w.ue(value - 4 if field == "log2_max_frame_num" else 4)
    w.ue(0)                                                         # poc type 0
    w.ue(value - 4 if field == "poc_lsb" else 4)
    w.ue(value if field == "num_ref_frames" else 1); w.u(1, 0)
    w.ue(value - 1 if field == "mb_w" else 5); w.ue(value - 1 if field == "mb_h" else 4)
    w.u(1, 1); w.u(1, 1); w.u(1, 0); w.u(1, 0)
    w.trailing()
    p = Parser(quiet=True, lib=lib)
    with pytest.raises(Exception):
        p.feed(7, 3, w.bytes())
    assert lib.p264parse_mb_width(p.h) == 0
