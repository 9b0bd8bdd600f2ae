"""Python mirrors of the two C-ABI layers: the host bitstream parser (p264parse_*) and the
MI355X reconstruction context (p264hip_*).  Thin ctypes plumbing - every pixel is produced by
the HIP kernels in libp264amd.so; nothing here computes."""
import ctypes as C

import numpy as np

from . import _native as N


class P264Error(RuntimeError):
    pass


class ParsedPicture:
    """An owned copy of one p264hip_picture_t (the parser reuses its buffers)."""

    def __init__(self, desc):
        n = desc.mb_w * desc.mb_h
        self.mb = np.ctypeslib.as_array(C.cast(desc.mb, C.POINTER(C.c_uint8)), (n * 16,)).copy()
        self.mv = np.ctypeslib.as_array(desc.mv, (n * 32,)).copy()
        self.ref_idx = np.ctypeslib.as_array(desc.ref_idx, (n * 4,)).copy()
        self.i4modes = np.ctypeslib.as_array(desc.i4modes, (n * 16,)).copy()
        nc = int(desc.n_coef_blocks)
        self.coefs = np.ctypeslib.as_array(desc.coefs, (nc * 16,)).copy() if nc else np.zeros(16, np.int16)
        d = N.Picture()
        C.memmove(C.byref(d), C.byref(desc), C.sizeof(N.Picture))
        if desc.slice_type == N.SLICE_B:                # list-1 arrays of a B picture
            self.mv_l1 = np.ctypeslib.as_array(desc.mv_l1, (n * 32,)).copy()
            self.ref_idx_l1 = np.ctypeslib.as_array(desc.ref_idx_l1, (n * 4,)).copy()
            d.mv_l1 = C.cast(self.mv_l1.ctypes.data, C.POINTER(C.c_int16))
            d.ref_idx_l1 = C.cast(self.ref_idx_l1.ctypes.data, C.POINTER(C.c_int8))
        else:
            d.mv_l1 = C.POINTER(C.c_int16)()
            d.ref_idx_l1 = C.POINTER(C.c_int8)()
        d.mb = C.cast(self.mb.ctypes.data, C.POINTER(N.MbInfo))
        d.mv = C.cast(self.mv.ctypes.data, C.POINTER(C.c_int16))
        d.ref_idx = C.cast(self.ref_idx.ctypes.data, C.POINTER(C.c_int8))
        d.i4modes = C.cast(self.i4modes.ctypes.data, C.POINTER(C.c_uint8))
        d.coefs = C.cast(self.coefs.ctypes.data, C.POINTER(C.c_int16))
        self.desc = d

    # convenience views -------------------------------------------------------------------
    @property
    def mb_w(self):
        return self.desc.mb_w

    @property
    def mb_h(self):
        return self.desc.mb_h

    @property
    def n_mb(self):
        return self.desc.mb_w * self.desc.mb_h

    def mb_records(self):
        dt = np.dtype([("mb_type", "u1"), ("qp", "u1"), ("cbp", "u1"), ("intra_modes", "u1"),
                       ("coef_mask", "<u4"), ("coef_index", "<u4"), ("avail", "u1"), ("edges", "u1"), ("flags", "<u2")])
        return self.mb.view(dt)


class Parser:
    """p264parse_* : NAL units in, complete parsed pictures out (CPU, serial by nature)."""

    def __init__(self, quiet=True, strict=False, lib=None):
        self.lib = lib or N.load()
        self.h = self.lib.p264parse_open((1 if quiet else 0) | (2 if strict else 0))
        if not self.h:
            raise P264Error("p264parse_open failed")

    def close(self):
        if self.h:
            self.lib.p264parse_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def feed(self, nal_type, ref_idc, rbsp):
        """Returns a ParsedPicture when this NAL completed a picture, else None."""
        pic = C.POINTER(N.Picture)()
        buf = (C.c_uint8 * max(len(rbsp), 1)).from_buffer_copy(rbsp if len(rbsp) else b"\0")
        rc = self.lib.p264parse_nal(self.h, nal_type, ref_idc, buf, len(rbsp), C.byref(pic))
        if rc < 0:
            raise P264Error("p264parse_nal failed on NAL type %d" % nal_type)
        return ParsedPicture(pic.contents) if rc == 1 else None

    @property
    def slots(self):
        return self.lib.p264parse_slots(self.h)

    def parse_stream(self, data, limit=None):
        """All pictures of an Annex-B byte string."""
        out = []
        for typ, idc, rbsp in N.split_annexb(self.lib, data):
            p = self.feed(typ, idc, rbsp)
            if p is not None:
                out.append(p)
                if limit and len(out) >= limit:
                    break
        return out


class HipReconstructor:
    """p264hip_* : frame stores and resident picture inputs on one MI355X."""

    def __init__(self, mb_w, mb_h, n_streams=1, slots=2, max_pictures=1, device=0, lib=None):
        self.lib = lib or N.load()
        if not hasattr(self.lib, "p264hip_create"):
            raise P264Error("libp264amd.so was built without the HIP layer")
        self.mb_w, self.mb_h, self.n_streams, self.slots = mb_w, mb_h, n_streams, slots
        h = C.c_void_p()
        rc = self.lib.p264hip_create(C.byref(h), device, mb_w, mb_h, n_streams, slots, max_pictures)
        if rc != 0:
            raise P264Error("p264hip_create: %s" % self.lib.p264hip_last_error().decode())
        self.h = h

    def _chk(self, rc, what):
        if rc != 0:
            raise P264Error("%s: %s" % (what, self.lib.p264hip_last_error().decode()))

    def close(self):
        if getattr(self, "h", None):
            self.lib.p264hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, first, pictures):
        arr = (N.Picture * len(pictures))(*[p.desc for p in pictures])
        self._chk(self.lib.p264hip_upload(self.h, first, arr, len(pictures)), "p264hip_upload")

    # ---- the packed form of a picture's arrays (p264hip_input_layout_t) and the roads into a slot that use it ----
    @staticmethod
    def pack(picture, lib=None):
        """The picture's arrays as one block in the layout of an input slot (numpy uint8; host side only)."""
        lib = lib or N.load()
        lay = N.InputLayout()
        if lib.p264hip_input_layout(C.byref(picture.desc), C.byref(lay)) != 0:
            raise P264Error("p264hip_input_layout failed")
        buf = np.zeros(lay.bytes, np.uint8)
        n = lib.p264hip_pack_input(C.byref(picture.desc), buf.ctypes.data, buf.size)
        if n != lay.bytes:
            raise P264Error("p264hip_pack_input: %d" % n)
        return buf

    @staticmethod
    def pack_compact(picture, lib=None):
        """The picture's arrays in the compact link format (include/p264hip.h: p264hip_compact_hdr_t); numpy uint8, host side only."""
        lib = lib or N.load()
        buf = np.zeros(lib.p264hip_compact_bound(C.byref(picture.desc)), np.uint8)
        n = lib.p264hip_pack_compact(C.byref(picture.desc), buf.ctypes.data, buf.size)
        if n < 0:
            raise P264Error("p264hip_pack_compact: %d" % n)
        return buf[:n].copy()

    @staticmethod
    def expand_compact(picture, compact, lib=None):
        """Host reference of the device expansion: the compact block back in the slot layout (what pack() gives, up to bytes no kernel reads)."""
        lib = lib or N.load()
        lay = N.InputLayout()
        if lib.p264hip_input_layout(C.byref(picture.desc), C.byref(lay)) != 0:
            raise P264Error("p264hip_input_layout failed")
        out = np.zeros(lay.bytes, np.uint8)
        rc = lib.p264hip_expand_compact(C.byref(picture.desc), compact.ctypes.data, compact.size, out.ctypes.data, out.size)
        if rc != 0:
            raise P264Error("p264hip_expand_compact: %d" % rc)
        return out

    def upload_compact(self, slot, picture, compact):
        self._chk(self.lib.p264hip_upload_compact(self.h, slot, C.byref(picture.desc), compact.ctypes.data, compact.size), "p264hip_upload_compact")

    def upload_packed(self, slot, picture, packed):
        self._chk(self.lib.p264hip_upload_packed(self.h, slot, C.byref(picture.desc), packed.ctypes.data, packed.size), "p264hip_upload_packed")

    def input_reserve(self, slot, picture):
        """(device address, bytes) of the block slot `slot` expects for this picture; fill it, then input_commit(slot)."""
        dev, n = C.c_void_p(), C.c_size_t()
        self._chk(self.lib.p264hip_input_reserve(self.h, slot, C.byref(picture.desc), C.byref(dev), C.byref(n)), "p264hip_input_reserve")
        return dev.value, n.value

    def input_commit(self, slot):
        self._chk(self.lib.p264hip_input_commit(self.h, slot), "p264hip_input_commit")

    def frame_planar_device(self, stream, slot, index=0):
        dev, n = C.c_void_p(), C.c_size_t()
        self._chk(self.lib.p264hip_frame_planar_device(self.h, stream, slot, index, C.byref(dev), C.byref(n)), "p264hip_frame_planar_device")
        return dev.value, n.value

    def clone_picture(self, dst, src):
        self._chk(self.lib.p264hip_clone_picture(self.h, dst, src), "p264hip_clone_picture")

    def reconstruct(self, pic_ids, streams):
        n = len(pic_ids)
        a = (C.c_int * n)(*pic_ids)
        b = (C.c_int * n)(*streams)
        self._chk(self.lib.p264hip_reconstruct(self.h, a, b, n), "p264hip_reconstruct")

    def submit(self, stream, picture):
        self._chk(self.lib.p264hip_submit(self.h, stream, C.byref(picture.desc)), "p264hip_submit")

    def sync(self):
        self._chk(self.lib.p264hip_sync(self.h), "p264hip_sync")

    def read_frame(self, stream, slot):
        w, h = self.mb_w * 16, self.mb_h * 16
        y = np.empty((h, w), np.uint8)
        u = np.empty((h // 2, w // 2), np.uint8)
        v = np.empty((h // 2, w // 2), np.uint8)
        self._chk(self.lib.p264hip_read_frame(self.h, stream, slot, y.ctypes.data, w, u.ctypes.data, v.ctypes.data, w // 2), "p264hip_read_frame")
        return y, u, v

    def write_frame(self, stream, slot, y, u, v):
        w = self.mb_w * 16
        y, u, v = (np.ascontiguousarray(a, np.uint8) for a in (y, u, v))
        self._chk(self.lib.p264hip_write_frame(self.h, stream, slot, y.ctypes.data, w, u.ctypes.data, v.ctypes.data, w // 2), "p264hip_write_frame")

    def timing_enable(self, on=True):
        self._chk(self.lib.p264hip_timing_enable(self.h, 1 if on else 0), "p264hip_timing_enable")

    def timing_reset(self):
        self._chk(self.lib.p264hip_timing_reset(self.h), "p264hip_timing_reset")

    def timing_read(self):
        ms = (C.c_double * N.NKERNELS)()
        cnt = (C.c_int64 * N.NKERNELS)()
        self._chk(self.lib.p264hip_timing_read(self.h, ms, cnt), "p264hip_timing_read")
        names = ("inter", "intra", "deblock", "reconstruct")
        return {names[i]: (ms[i], cnt[i]) for i in range(N.NKERNELS)}


    def last_launch(self):
        """What the last reconstruct call launched (p264hip_launch_info_t as a dict)."""
        li = N.LaunchInfo()
        self._chk(self.lib.p264hip_last_launch(self.h, C.byref(li)), "p264hip_last_launch")
        return {n: int(getattr(li, n)) for n, _ in N.LaunchInfo._fields_ if n != "reserved"}


def device_count(lib=None):
    lib = lib or N.load()
    return lib.p264hip_device_count() if hasattr(lib, "p264hip_device_count") else 0
