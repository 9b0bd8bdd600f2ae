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
    """p264parse_set_allocator: every picture array comes from the caller's allocator, output unchanged."""
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
            assert C.addressof(pic.contents.mb.contents) in live and C.addressof(pic.contents.mv.contents) in live
            mine = C.string_at(pic.contents.mv, n_mb * 64)
            assert mine == C.string_at(ref[got].desc.mv, n_mb * 64)
            got += 1
            if got == 5:
                break
    assert total[0] >= 10                                               # 2 buffers x 5 arrays
    lib.p264parse_close(h)
    assert not live                                                     # everything given back
