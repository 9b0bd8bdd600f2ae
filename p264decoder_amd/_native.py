"""ctypes view of the C ABI declared in include/p264hip.h, include/p264parse.h and
include/p264_dropin.h.  No torch types cross this boundary: plain pointers and sizes.

The shared library is built in-tree by ``p264decoder_amd/build.py`` (hipcc, gfx950) as
``p264decoder_amd/libp264amd.so``.  There is no CPU fallback for the reconstruction:
if the library is missing, importing the product entry points raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libp264amd.so")

MAX_REFS = 16
NKERNELS = 4

MB_I4x4, MB_I16x16, MB_IPCM, MB_P_L0, MB_P_8x8, MB_P_SKIP, MB_B = range(7)
SLICE_P, SLICE_B, SLICE_I = 0, 1, 2
COEF_LUMA_DC = 1 << 24
COEF_CHROMA_DC = 1 << 25
AVAIL_LEFT, AVAIL_TOP, AVAIL_TOPRIGHT, AVAIL_TOPLEFT = 1, 2, 4, 8
EDGE_LEFT, EDGE_TOP, EDGE_INNER = 1, 2, 4


class MbInfo(C.Structure):
    """p264hip_mb_t (16 bytes)"""
    _fields_ = [
        ("mb_type", C.c_uint8), ("qp", C.c_uint8), ("cbp", C.c_uint8), ("intra_modes", C.c_uint8),
        ("coef_mask", C.c_uint32), ("coef_index", C.c_uint32),
        ("avail", C.c_uint8), ("edges", C.c_uint8), ("flags", C.c_uint16),
    ]


class Picture(C.Structure):
    """p264hip_picture_t"""
    _fields_ = [
        ("mb_w", C.c_int32), ("mb_h", C.c_int32), ("slice_type", C.c_int32),
        ("chroma_qp_offset", C.c_int32), ("deblock", C.c_int32),
        ("alpha_c0_offset", C.c_int32), ("beta_offset", C.c_int32),
        ("dst_slot", C.c_int32), ("n_ref", C.c_int32), ("ref_slot", C.c_int32 * MAX_REFS),
        ("n_coef_blocks", C.c_uint32), ("frame_num", C.c_uint32),
        ("mb", C.POINTER(MbInfo)), ("mv", C.POINTER(C.c_int16)), ("ref_idx", C.POINTER(C.c_int8)),
        ("i4modes", C.POINTER(C.c_uint8)), ("coefs", C.POINTER(C.c_int16)),
        ("mv_l1", C.POINTER(C.c_int16)), ("ref_idx_l1", C.POINTER(C.c_int8)), ("n_ref_l1", C.c_int32), ("weighted_bipred", C.c_int32),
        ("ref_slot_l1", C.c_int32 * MAX_REFS), ("bipred_weight", C.c_int16 * (MAX_REFS * MAX_REFS)),
    ]


assert C.sizeof(MbInfo) == 16


class LaunchInfo(C.Structure):
    """p264hip_launch_info_t"""
    _fields_ = [(n, C.c_int32) for n in ("pictures", "compute_units", "mc_wgs_per_picture", "intra_waves", "edge_info_fused",
                                         "deblock_pics_per_wg", "deblock_rb_log2", "deblock_waves", "deblock_wgs", "deblock_odd_single")] + [("reserved", C.c_int32 * 6)]


BUILD_TIMING = 1


class InputLayout(C.Structure):
    """p264hip_input_layout_t"""
    _fields_ = [(n, C.c_size_t) for n in ("off_mb", "off_mv", "off_ref", "off_i4", "off_coef", "off_mv_l1", "off_ref_l1", "off_weights", "bytes")]


class CompactList(C.Structure):
    _fields_ = [("off_ref", C.c_uint32), ("off_shape", C.c_uint32), ("off_vec", C.c_uint32), ("n_vec", C.c_uint32)]


class CompactHdr(C.Structure):
    """p264hip_compact_hdr_t (128 bytes)"""
    _fields_ = [("magic", C.c_uint32), ("n_mb", C.c_uint32), ("n_coef_blocks", C.c_uint32), ("bytes", C.c_uint32), ("n_lists", C.c_uint32),
                ("off_rec", C.c_uint32), ("off_i4flag", C.c_uint32), ("off_i4", C.c_uint32), ("off_lvflag", C.c_uint32), ("off_levels", C.c_uint32),
                ("off_weights", C.c_uint32), ("n_i4", C.c_uint32), ("level_bytes", C.c_uint32), ("list", CompactList * 2), ("reserved", C.c_uint32 * 11)]


class PipeStats(C.Structure):
    """p264pipe_stats_t"""
    _fields_ = [("pictures", C.c_int64), ("bytes", C.c_int64), ("seconds", C.c_double), ("parse_seconds", C.c_double),
                ("submit_seconds", C.c_double), ("rounds", C.c_int), ("streams", C.c_int), ("threads", C.c_int), ("reserved", C.c_int),
                ("bytes_uploaded", C.c_int64), ("wait_parse_seconds", C.c_double), ("wait_device_seconds", C.c_double)]

_lib = None


def load(path=None):
    """Load libp264amd.so and declare prototypes.  Raises OSError if it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    lib = C.CDLL(path or os.environ.get("P264AMD_LIB") or LIB_PATH)
    u8p, i64p = C.POINTER(C.c_uint8), C.POINTER(C.c_int64)
    # ---- p264parse.h
    lib.p264parse_open.restype = C.c_void_p
    lib.p264parse_open.argtypes = [C.c_int]
    lib.p264parse_close.argtypes = [C.c_void_p]
    lib.p264parse_nal.restype = C.c_int
    lib.p264parse_nal.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.POINTER(Picture))]
    for f in ("p264parse_mb_width", "p264parse_mb_height", "p264parse_slots", "p264parse_generation"):
        getattr(lib, f).restype = C.c_int
        getattr(lib, f).argtypes = [C.c_void_p]
    lib.p264_annexb_next.restype = C.c_int
    lib.p264_annexb_next.argtypes = [C.c_void_p, C.c_int64, i64p, i64p, i64p]
    # ---- p264hip.h
    if hasattr(lib, "p264hip_create"):
        lib.p264hip_create.restype = C.c_int
        lib.p264hip_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        lib.p264hip_destroy.argtypes = [C.c_void_p]
        lib.p264hip_last_error.restype = C.c_char_p
        lib.p264hip_device_count.restype = C.c_int
        lib.p264hip_upload.restype = C.c_int
        lib.p264hip_upload.argtypes = [C.c_void_p, C.c_int, C.POINTER(Picture), C.c_int]
        lib.p264hip_clone_picture.restype = C.c_int
        lib.p264hip_clone_picture.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.p264hip_reconstruct.restype = C.c_int
        lib.p264hip_reconstruct.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
        lib.p264hip_submit.restype = C.c_int
        lib.p264hip_submit.argtypes = [C.c_void_p, C.c_int, C.POINTER(Picture)]
        lib.p264hip_sync.restype = C.c_int
        lib.p264hip_sync.argtypes = [C.c_void_p]
        lib.p264hip_read_frame.restype = C.c_int
        lib.p264hip_read_frame.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        lib.p264hip_write_frame.restype = C.c_int
        lib.p264hip_write_frame.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        lib.p264hip_timing_enable.restype = C.c_int
        lib.p264hip_timing_enable.argtypes = [C.c_void_p, C.c_int]
        lib.p264hip_timing_read.restype = C.c_int
        lib.p264hip_timing_read.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        lib.p264hip_timing_reset.restype = C.c_int
        lib.p264hip_timing_reset.argtypes = [C.c_void_p]
        lib.p264hip_last_launch.restype = C.c_int
        lib.p264hip_last_launch.argtypes = [C.c_void_p, C.POINTER(LaunchInfo)]
        lib.p264hip_build_info.restype = C.c_int
        lib.p264hip_build_info.argtypes = []
        lib.p264hip_upload_copies.restype = C.c_int64
        lib.p264hip_upload_copies.argtypes = [C.c_void_p]
        lib.p264hip_upload_async.restype = C.c_int
        lib.p264hip_upload_async.argtypes = [C.c_void_p, C.c_int, C.POINTER(Picture)]
        lib.p264hip_host_alloc.restype = C.c_void_p
        lib.p264hip_host_alloc.argtypes = [C.c_size_t]
        lib.p264hip_host_free.argtypes = [C.c_void_p]
        lib.p264hip_marker.restype = C.c_int
        lib.p264hip_input_layout.argtypes = [C.POINTER(Picture), C.POINTER(InputLayout)]
        lib.p264hip_compact_bound.restype = C.c_size_t
        lib.p264hip_compact_bound.argtypes = [C.POINTER(Picture)]
        lib.p264hip_pack_compact.restype = C.c_int64
        lib.p264hip_pack_compact.argtypes = [C.POINTER(Picture), C.c_void_p, C.c_size_t]
        lib.p264hip_expand_compact.restype = C.c_int
        lib.p264hip_expand_compact.argtypes = [C.POINTER(Picture), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        lib.p264hip_compact_header_ok.restype = C.c_int
        lib.p264hip_compact_header_ok.argtypes = [C.POINTER(Picture), C.c_void_p, C.c_size_t]
        lib.p264hip_compact_check.restype = C.c_int
        lib.p264hip_compact_check.argtypes = [C.POINTER(Picture), C.c_void_p, C.c_size_t]
        lib.p264hip_upload_compact.restype = C.c_int
        lib.p264hip_upload_compact.argtypes = [C.c_void_p, C.c_int, C.POINTER(Picture), C.c_void_p, C.c_size_t]
        lib.p264hip_pack_input.restype = C.c_int64
        lib.p264hip_pack_input.argtypes = [C.POINTER(Picture), C.c_void_p, C.c_size_t]
        lib.p264hip_unpack_input.argtypes = [C.POINTER(Picture), C.c_void_p, C.c_size_t, C.POINTER(Picture)]
        lib.p264hip_upload_packed.argtypes = [C.c_void_p, C.c_int, C.POINTER(Picture), C.c_void_p, C.c_size_t]
        lib.p264hip_input_reserve.argtypes = [C.c_void_p, C.c_int, C.POINTER(Picture), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        lib.p264hip_input_commit.argtypes = [C.c_void_p, C.c_int]
        lib.p264hip_frame_planar_device.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        lib.p264hip_copy_to_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.p264hip_copy_from_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.p264hip_marker.argtypes = [C.c_void_p]
        lib.p264hip_marker_wait.restype = C.c_int
        lib.p264hip_marker_wait.argtypes = [C.c_void_p, C.c_int]
    # ---- p264pipe.h
    if hasattr(lib, "p264pipe_open"):
        lib.p264pipe_open.restype = C.c_void_p
        lib.p264pipe_open.argtypes = [C.c_int, C.c_int, C.c_int]
        lib.p264pipe_set_input.restype = C.c_int
        lib.p264pipe_set_input.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
        lib.p264pipe_run.restype = C.c_int
        lib.p264pipe_run.argtypes = [C.c_void_p, C.c_int, C.POINTER(PipeStats)]
        lib.p264pipe_frame_size.restype = C.c_int
        lib.p264pipe_frame_size.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.p264pipe_read_frame.restype = C.c_int
        lib.p264pipe_read_frame.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        lib.p264pipe_stream_pictures.restype = C.c_int64
        lib.p264pipe_stream_pictures.argtypes = [C.c_void_p, C.c_int]
        lib.p264pipe_close.argtypes = [C.c_void_p]
    if path is None:
        _lib = lib
    return lib


def split_annexb(lib, data):
    """Yield (nal_type, nal_ref_idc, rbsp bytes) for every NAL unit of an Annex-B byte string,
    applying the reference's emulation-prevention strip (core/core.c:310-336, incl. quirk A-Q10)."""
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    pos, off, ln = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    while lib.p264_annexb_next(buf, len(data), C.byref(pos), C.byref(off), C.byref(ln)):
        if ln.value < 1:
            continue
        nal = data[off.value:off.value + ln.value]
        hdr = nal[0]
        yield hdr & 0x1F, (hdr >> 5) & 3, strip_emulation(nal)


def strip_emulation(nal):
    """Payload after the NAL header byte with 00 00 03 -> 00 00, except when the 03 is within the
    last three bytes (the reference's loop bound, core/core.c:323)."""
    src = nal
    n = len(src)
    out = bytearray()
    i = 1
    while i < n:
        if i < n - 3 and src[i] == 0 and src[i + 1] == 0 and src[i + 2] == 3:
            out += b"\x00\x00"
            i += 3
            continue
        out.append(src[i])
        i += 1
    return bytes(out)
