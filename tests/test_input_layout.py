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


@pytest.mark.parametrize("case", ["cif_ip", "tiny_1x1", "wide_70", "dense"])
def test_the_parser_builds_its_pictures_in_the_layout_of_an_input_slot(lib, case):
    """records | vectors | reference indices | intra 4x4 modes | coded levels as sections of ONE host block at the offsets of
    p264hip_input_layout: p264hip_upload / _upload_async then need one host -> HBM copy per picture, not five."""
    import ctypes as C
    from p264decoder_amd import _native as N
    h = lib.p264parse_open(1)
    assert h
    n_pics = in_one = 0
    for typ, idc, rbsp in N.split_annexb(lib, synth_cases.stream_bytes(case)):
        pic = C.POINTER(N.Picture)()
        buf = (C.c_uint8 * max(len(rbsp), 1)).from_buffer_copy(rbsp if len(rbsp) else b"\0")
        rc = lib.p264parse_nal(h, typ, idc, buf, len(rbsp), C.byref(pic))
        assert rc >= 0
        if rc != 1:
            continue
        d = pic.contents
        lay = N.InputLayout()
        assert lib.p264hip_input_layout(C.byref(d), C.byref(lay)) == 0
        base = C.cast(d.mb, C.c_void_p).value
        assert C.cast(d.mv, C.c_void_p).value == base + lay.off_mv
        assert C.cast(d.ref_idx, C.c_void_p).value == base + lay.off_ref
        assert C.cast(d.i4modes, C.c_void_p).value == base + lay.off_i4
        # (the section holds 8 blocks per macroblock; a picture with more - up to 26 are possible - moves its levels out)
        assert d.n_coef_blocks == 0 or C.cast(d.coefs, C.c_void_p).value == base + lay.off_coef or d.n_coef_blocks > 8 * d.mb_w * d.mb_h + 64
        in_one += d.n_coef_blocks <= 8 * d.mb_w * d.mb_h + 64
        n_pics += 1
    lib.p264parse_close(h)
    assert n_pics >= 6 and in_one >= n_pics - 2


def _seam_pictures():
    """parsed pictures of several streams + seam-level random pictures with levels up to the int16 limits and sub-8x8 vectors"""
    import numpy as np
    from tests import seam_fuzz
    pics = []
    for case in ("cif_ip", "dense", "qp0", "mv_far"):
        pics += Parser(quiet=True).parse_stream(synth_cases.stream_bytes(case))[:6]
    rng = np.random.default_rng(77)
    for k in range(6):
        pics.append(seam_fuzz.make_picture(rng, 7, 5, p_picture=(k != 2), n_ref=2, slots=3, dst_slot=0, level_style="mixed" if k % 2 else "small", sub8x8=True))
    return pics


def test_compact_link_format_round_trip(lib):
    """p264hip_pack_compact -> p264hip_expand_compact gives the slot layout back byte for byte - records, both lists' vectors and
    reference indices, Intra4x4 modes, coded levels, a B picture's weights (only the padding between the sections is not
    compared); the block is smaller; a tampered block is refused by the full check."""
    import ctypes as C
    import numpy as np
    from p264decoder_amd import HipReconstructor, _native as N
    assert C.sizeof(N.CompactHdr) == 128
    shapes, n_b = set(), 0
    pics = _seam_pictures() + Parser(quiet=True).parse_stream(synth_cases.stream_bytes(B_CIF))
    rng = np.random.default_rng(78)
    from tests import seam_fuzz
    for k in range(3):
        pics.append(seam_fuzz.make_picture(rng, 7, 5, p_picture=True, b_picture=True, n_ref=2, n_ref_l1=2, slots=3, dst_slot=0, level_style="mixed", sub8x8=True))
    for p in pics:
        plain = HipReconstructor.pack(p, lib)
        comp = HipReconstructor.pack_compact(p, lib)
        assert lib.p264hip_compact_check(C.byref(p.desc), comp.ctypes.data, comp.size) == 0
        back = HipReconstructor.expand_compact(p, comp, lib)
        lay = N.InputLayout()
        lib.p264hip_input_layout(C.byref(p.desc), C.byref(lay))
        n = p.desc.mb_w * p.desc.mb_h
        nb = p.desc.n_coef_blocks
        sections = [(0, n * 16), (lay.off_mv, n * 64), (lay.off_ref, n * 4), (lay.off_i4, n * 16), (lay.off_coef, nb * 32)]
        if p.desc.slice_type == N.SLICE_B:
            n_b += 1
            sections += [(lay.off_mv_l1, n * 64), (lay.off_ref_l1, n * 4), (lay.off_weights, 512)]
        for off, size in sections:
            assert np.array_equal(back[off:off + size], plain[off:off + size]), (off, size)
        hdr = N.CompactHdr.from_buffer_copy(comp[:128].tobytes())
        assert hdr.n_lists == (2 if p.desc.slice_type == N.SLICE_B else 1)
        shape_bits = comp[hdr.list[0].off_shape:hdr.list[0].off_shape + (n + 3) // 4]
        shapes |= {int((shape_bits[i >> 2] >> (2 * (i & 3))) & 3) for i in range(n)}
        assert comp.size < plain.size
        # tampering: a shape bit, a count, the size
        for off, what in ((int(hdr.list[0].off_shape), "shape"), (N.CompactHdr.list.offset + 12, "n_vec"), (12, "bytes")):
            bad = comp.copy()
            bad[off] ^= 1
            assert lib.p264hip_compact_check(C.byref(p.desc), bad.ctypes.data, bad.size) != 0, what
    assert shapes == {0, 1, 2, 3} and n_b >= 5
