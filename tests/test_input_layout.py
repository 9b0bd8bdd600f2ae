"""The packed form of a picture's arrays (p264hip_input_layout_t, include/p264hip.h): what the stream fan-out sends between
ranks and what an input slot of the HIP layer holds.  Host side only: layout, pack, the unpacked view, the record check."""
import ctypes as C

import numpy as np
import pytest

from p264decoder_amd import HipReconstructor, Parser, _native as N
from tests import synth_cases


def _pictures(case, lib, limit=3):
    return Parser(quiet=True, lib=lib).parse_stream(synth_cases.stream_bytes(case), limit=limit)


B_CIF = "--mbw 22 --mbh 18 --frames 7 --seed 5 --refs 2 --bframes 2 --implicit --d8inf --coded 8 --maxlevel 8"      # I P B B P B B


@pytest.mark.parametrize("case", ["cif_ip", B_CIF])
def test_pack_and_unpacked_view_round_trip(lib, case):
    pics = _pictures(case, lib, limit=5)
    assert case == "cif_ip" or any(p.desc.slice_type == 1 for p in pics)
    for p in pics:
        lay = N.InputLayout()
        assert lib.p264hip_input_layout(C.byref(p.desc), C.byref(lay)) == 0
        n = p.desc.mb_w * p.desc.mb_h
        offs = [lay.off_mb, lay.off_mv, lay.off_ref, lay.off_i4, lay.off_coef]
        assert offs == sorted(offs) and all(o % 256 == 0 for o in offs) and lay.bytes % 256 == 0
        assert lay.off_mv - lay.off_mb >= n * 16 and lay.off_ref - lay.off_mv >= n * 64 and lay.bytes >= lay.off_coef + p.desc.n_coef_blocks * 32
        buf = HipReconstructor.pack(p, lib)
        view = N.Picture()
        assert lib.p264hip_unpack_input(C.byref(p.desc), buf.ctypes.data, buf.size, C.byref(view)) == 0
        for name, count, typ in (("mb", n * 16, C.c_uint8), ("mv", n * 32, C.c_int16), ("ref_idx", n * 4, C.c_int8), ("i4modes", n * 16, C.c_uint8),
                                 ("coefs", p.desc.n_coef_blocks * 16, C.c_int16)):
            if not count:
                continue
            a = np.ctypeslib.as_array(C.cast(getattr(p.desc, name), C.POINTER(typ)), (count,))
            b = np.ctypeslib.as_array(C.cast(getattr(view, name), C.POINTER(typ)), (count,))
            assert np.array_equal(a, b), name
        if p.desc.slice_type == 1:
            assert lay.off_mv_l1 > lay.off_coef and lay.off_weights > lay.off_ref_l1 > lay.off_mv_l1
            a = np.ctypeslib.as_array(p.desc.mv_l1, (n * 32,))
            assert np.array_equal(a, np.ctypeslib.as_array(view.mv_l1, (n * 32,)))
            w = buf[lay.off_weights:lay.off_weights + 512].view(np.int16)
            assert np.array_equal(w, np.ctypeslib.as_array(p.desc.bipred_weight))
        else:
            assert not view.mv_l1 and lay.off_mv_l1 == 0
        assert view.n_coef_blocks == p.desc.n_coef_blocks and view.dst_slot == p.desc.dst_slot


def test_pack_refuses_records_that_point_outside_the_coefficients(lib):
    p = _pictures("cif_ip", lib, limit=1)[0]
    lay = N.InputLayout()
    lib.p264hip_input_layout(C.byref(p.desc), C.byref(lay))
    buf = np.zeros(lay.bytes, np.uint8)
    assert lib.p264hip_pack_input(C.byref(p.desc), buf.ctypes.data, buf.size - 1) < 0          # too small
    coded = [i for i in range(p.desc.mb_w * p.desc.mb_h) if p.desc.mb[i].coef_mask]
    keep = p.desc.mb[coded[-1]].coef_index
    p.desc.mb[coded[-1]].coef_index = p.desc.n_coef_blocks                                       # its blocks now lie behind the end
    try:
        assert lib.p264hip_pack_input(C.byref(p.desc), buf.ctypes.data, buf.size) < 0
    finally:
        p.desc.mb[coded[-1]].coef_index = keep
    assert lib.p264hip_pack_input(C.byref(p.desc), buf.ctypes.data, buf.size) == lay.bytes


def test_every_host_entry_refuses_a_cpu_older_than_the_build(lib):
    """The host objects are built for x86-64-v3; cpu_check.c (built for plain x86-64) looks at the CPU when the library is loaded
    and every public entry of the other objects asks it - a message and an error instead of an illegal instruction.  The
    verdict is a flag of the library: flipped here."""
    import ctypes as C
    from p264decoder_amd import Parser, _native
    flag = C.c_int.in_dll(lib, "p264amd_cpu_unsupported")
    assert flag.value == 0
    pics = Parser(quiet=True, lib=lib).parse_stream(synth_cases.stream_bytes("tiny_1x1"))
    flag.value = 1
    try:
        assert not lib.p264parse_open(0)
        assert not lib.p264pipe_open(-1, 1, 1)
        buf = (C.c_uint8 * (1 << 16))()
        assert lib.p264hip_pack_input(C.byref(pics[0].desc), buf, len(buf)) < 0
        p = _native.p264_param_t() if hasattr(_native, "p264_param_t") else None
        assert not lib.p264_decoder_open(None if p is None else C.byref(p))
    finally:
        flag.value = 0
    assert lib.p264hip_pack_input(C.byref(pics[0].desc), buf, len(buf)) > 0
