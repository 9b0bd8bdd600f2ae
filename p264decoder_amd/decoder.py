"""Python mirror of the reference's public decode API (p264.h:266,351,379-382), bound to the
symbols of the same names exported by libp264amd.so (include/p264_dropin.h).

    dec = Decoder()                      # p264_param_default + p264_decoder_open
    for y, u, v in dec.decode_annexb(open("clip.264", "rb").read()): ...
    dec.close()                          # p264_decoder_close

Names, argument meaning and error behaviour follow the reference: decode() takes one NAL
unit, returns a picture or None, raises on the reference's -1.
"""
import ctypes as C

import numpy as np

from . import _native as N
from .recon import P264Error


class _Zone(C.Structure):
    _fields_ = [("i_start", C.c_int), ("i_end", C.c_int), ("b_force_qp", C.c_int), ("i_qp", C.c_int), ("f_bitrate_factor", C.c_float)]


class _Vui(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("i_sar_height", "i_sar_width", "i_overscan", "i_vidformat", "b_fullrange",
                                       "i_colorprim", "i_transfer", "i_colmatrix", "i_chroma_loc")]


class _Analyse(C.Structure):
    _fields_ = [("intra", C.c_uint), ("inter", C.c_uint)] + [(n, C.c_int) for n in (
        "b_transform_8x8", "b_weighted_bipred", "i_direct_mv_pred", "i_chroma_qp_offset", "i_me_method", "i_me_range",
        "i_mv_range", "i_subpel_refine", "b_chroma_me", "b_bframe_rdo", "b_mixed_references", "i_trellis", "b_fast_pskip", "b_psnr")]


class _Rc(C.Structure):
    _fields_ = [("i_qp_constant", C.c_int), ("i_qp_min", C.c_int), ("i_qp_max", C.c_int), ("i_qp_step", C.c_int),
                ("b_cbr", C.c_int), ("i_bitrate", C.c_int), ("i_rf_constant", C.c_int), ("f_rate_tolerance", C.c_float),
                ("i_vbv_max_bitrate", C.c_int), ("i_vbv_buffer_size", C.c_int), ("f_vbv_buffer_init", C.c_float),
                ("f_ip_factor", C.c_float), ("f_pb_factor", C.c_float),
                ("b_stat_write", C.c_int), ("psz_stat_out", C.c_char_p), ("b_stat_read", C.c_int), ("psz_stat_in", C.c_char_p),
                ("psz_rc_eq", C.c_char_p), ("f_qcompress", C.c_float), ("f_qblur", C.c_float), ("f_complexity_blur", C.c_float),
                ("zones", C.POINTER(_Zone)), ("i_zones", C.c_int), ("psz_zones", C.c_char_p)]


class Param(C.Structure):
    """p264_param_t (p264.h:119-244)"""
    _fields_ = [("cpu", C.c_uint), ("i_threads", C.c_int),
                ("i_width", C.c_int), ("i_height", C.c_int), ("i_csp", C.c_int), ("i_level_idc", C.c_int), ("i_frame_total", C.c_int),
                ("vui", _Vui), ("i_fps_num", C.c_int), ("i_fps_den", C.c_int),
                ("i_frame_reference", C.c_int), ("i_keyint_max", C.c_int), ("i_keyint_min", C.c_int), ("i_scenecut_threshold", C.c_int),
                ("i_bframe", C.c_int), ("b_bframe_adaptive", C.c_int), ("i_bframe_bias", C.c_int), ("b_bframe_pyramid", C.c_int),
                ("b_deblocking_filter", C.c_int), ("i_deblocking_filter_alphac0", C.c_int), ("i_deblocking_filter_beta", C.c_int),
                ("b_cabac", C.c_int), ("i_cabac_init_idc", C.c_int), ("i_cqm_preset", C.c_int), ("psz_cqm_file", C.c_char_p),
                ("cqm_4iy", C.c_uint8 * 16), ("cqm_4ic", C.c_uint8 * 16), ("cqm_4py", C.c_uint8 * 16), ("cqm_4pc", C.c_uint8 * 16),
                ("cqm_8iy", C.c_uint8 * 64), ("cqm_8py", C.c_uint8 * 64),
                ("pf_log", C.c_void_p), ("p_log_private", C.c_void_p), ("i_log_level", C.c_int), ("b_visualize", C.c_int),
                ("analyse", _Analyse), ("rc", _Rc), ("b_aud", C.c_int), ("b_repeat_headers", C.c_int)]


class Image(C.Structure):
    _fields_ = [("i_csp", C.c_int), ("i_plane", C.c_int), ("i_stride", C.c_int * 4), ("plane", C.POINTER(C.c_uint8) * 4)]


class PictureOut(C.Structure):
    """p264_picture_t (p264.h:280-296)"""
    _fields_ = [("i_type", C.c_int), ("i_qpplus1", C.c_int), ("i_pts", C.c_int64), ("i_width", C.c_int), ("i_height", C.c_int), ("img", Image)]


class Nal(C.Structure):
    """p264_nal_t (p264.h:333-341)"""
    _fields_ = [("i_ref_idc", C.c_int), ("i_type", C.c_int), ("i_payload", C.c_int), ("p_payload", C.POINTER(C.c_uint8))]


def _bind(lib):
    lib.p264_param_default.argtypes = [C.POINTER(Param)]
    lib.p264_param_default.restype = None
    lib.p264_nal_decode.argtypes = [C.POINTER(Nal), C.c_void_p, C.c_int]
    lib.p264_nal_decode.restype = C.c_int
    lib.p264_decoder_open.argtypes = [C.POINTER(Param)]
    lib.p264_decoder_open.restype = C.c_void_p
    lib.p264_decoder_decode.argtypes = [C.c_void_p, C.POINTER(C.POINTER(PictureOut)), C.POINTER(Nal)]
    lib.p264_decoder_decode.restype = C.c_int
    lib.p264_decoder_close.argtypes = [C.c_void_p]
    lib.p264_decoder_close.restype = None
    return lib


def param_default(lib=None):
    """p264_param_default"""
    lib = _bind(lib or N.load())
    p = Param()
    lib.p264_param_default(C.byref(p))
    return p


class Decoder:
    """p264_decoder_open / p264_decoder_decode / p264_decoder_close."""

    def __init__(self, param=None, lib=None):
        self.lib = _bind(lib or N.load())
        self.param = param or param_default(self.lib)
        self.h = self.lib.p264_decoder_open(C.byref(self.param))
        if not self.h:
            raise P264Error("p264_decoder_open failed (no MI355X / HIP device?  there is no CPU fallback)")
        self._payload = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.p264_decoder_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def nal_decode(self, nal_bytes):
        """p264_nal_decode: header split + emulation-prevention strip into a caller-owned payload."""
        n = len(nal_bytes)
        self._payload = (C.c_uint8 * (n + 8))()
        nal = Nal()
        nal.p_payload = C.cast(self._payload, C.POINTER(C.c_uint8))
        src = (C.c_uint8 * n).from_buffer_copy(nal_bytes)
        self.lib.p264_nal_decode(C.byref(nal), src, n)
        return nal

    def decode(self, nal):
        """p264_decoder_decode: one NAL in; (y, u, v) numpy copies of the MB-aligned picture, or None."""
        pp = C.POINTER(PictureOut)()
        rc = self.lib.p264_decoder_decode(self.h, C.byref(pp), C.byref(nal))
        if rc < 0:
            raise P264Error("p264_decoder_decode returned %d" % rc)
        if not pp:
            return None
        pic = pp.contents
        out = []
        for i in range(3):
            w = pic.i_width >> (1 if i else 0)
            h = pic.i_height >> (1 if i else 0)
            st = pic.img.i_stride[i]
            a = np.ctypeslib.as_array(pic.img.plane[i], ((h - 1) * st + w,))
            out.append(np.lib.stride_tricks.as_strided(a, (h, w), (st, 1)).copy())
        return tuple(out)

    def decode_annexb(self, data):
        """The loop of the reference CLI (p264decoder.c:229-350) over an in-memory Annex-B stream."""
        buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
        pos, off, ln = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        while self.lib.p264_annexb_next(buf, len(data), C.byref(pos), C.byref(off), C.byref(ln)):
            if ln.value < 1:
                continue
            nal = self.nal_decode(data[off.value:off.value + ln.value])
            pic = self.decode(nal)
            if pic is not None:
                yield pic
