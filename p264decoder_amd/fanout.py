"""Python mirror of include/p264fan.h: stream fan-out (one rank owns inputs and outputs, pictures are scattered to the
ranks that own the streams and the reconstructed planes gathered back).  ctypes plumbing only."""
import ctypes as C

import numpy as np

from . import _native as N


class Transport(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("send", C.c_void_p), ("recv", C.c_void_p), ("group_begin", C.c_void_p),
                ("group_end", C.c_void_p), ("close", C.c_void_p), ("name", C.c_char_p), ("abort", C.c_void_p),
                ("send_dev", C.c_void_p), ("recv_dev", C.c_void_p)]


BK_OPEN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int)
BK_RECON = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(N.Picture), C.POINTER(C.c_uint8))
BK_CLOSE = C.CFUNCTYPE(None, C.c_void_p)


class Backend(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("open", BK_OPEN), ("reconstruct", BK_RECON), ("close", BK_CLOSE), ("sync", C.c_void_p),
                ("reserve", C.c_void_p), ("reconstruct_reserved", C.c_void_p), ("planes", C.c_void_p)]      # (the device road: NULL in test backends)


class FanStats(C.Structure):
    _fields_ = [("pictures", C.c_int64), ("pictures_remote", C.c_int64), ("bytes_scattered", C.c_int64), ("bytes_gathered", C.c_int64),
                ("seconds", C.c_double), ("parse_seconds", C.c_double), ("exchange_seconds", C.c_double), ("rounds", C.c_int), ("world", C.c_int),
                ("parse_wait_seconds", C.c_double), ("reconstruct_seconds", C.c_double), ("parse_threads", C.c_int), ("device_road_rounds", C.c_int)]


FRAME_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_uint8))


def _proto(lib):
    lib.p264fan_open.restype = C.c_void_p
    lib.p264fan_open.argtypes = [C.c_int, C.c_int, C.POINTER(Transport), C.POINTER(Backend), C.c_int]
    lib.p264fan_close.argtypes = [C.c_void_p]
    lib.p264fan_worker_run.restype = C.c_int
    lib.p264fan_worker_run.argtypes = [C.c_void_p]
    lib.p264fan_root_run.restype = C.c_int
    lib.p264fan_root_run.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int, FRAME_CB, C.c_void_p, C.POINTER(FanStats)]
    lib.p264fan_tcp_transport.restype = C.c_int
    lib.p264fan_tcp_transport.argtypes = [C.POINTER(Transport), C.c_int, C.c_int, C.c_char_p, C.c_int]
    lib.p264fan_last_error.restype = C.c_char_p
    if hasattr(lib, "p264fan_rccl_transport"):
        lib.p264fan_rccl_unique_id.restype = C.c_int
        lib.p264fan_rccl_unique_id.argtypes = [C.POINTER(C.c_uint8)]
        lib.p264fan_rccl_transport.restype = C.c_int
        lib.p264fan_rccl_transport.argtypes = [C.POINTER(Transport), C.c_int, C.c_int, C.POINTER(C.c_uint8), C.c_int]


def rccl_unique_id(lib=None):
    lib = lib or N.load()
    _proto(lib)
    buf = (C.c_uint8 * 128)()
    if lib.p264fan_rccl_unique_id(buf):
        raise RuntimeError("p264fan_rccl_unique_id failed (librccl not loadable?)")
    return bytes(buf)


class FanOut:
    """One rank of a fan-out job.  transport: ("tcp", host, port) or ("rccl", unique_id bytes); backend: None = the MI355X
    path of the library on `device`, or a Backend structure (tests)."""

    def __init__(self, rank, world, transport, device=0, backend=None, lib=None):
        self.lib = lib or N.load()
        _proto(self.lib)
        self.rank, self.world = rank, world
        self.t = Transport()
        if world > 1:
            if transport[0] == "tcp":
                rc = self.lib.p264fan_tcp_transport(C.byref(self.t), rank, world, transport[1].encode(), transport[2])
            elif transport[0] == "rccl":
                uid = (C.c_uint8 * 128).from_buffer_copy(transport[1])
                rc = self.lib.p264fan_rccl_transport(C.byref(self.t), rank, world, uid, device)
            else:
                raise ValueError(transport[0])
            if rc:
                raise RuntimeError("fan-out transport %s: %s" % (transport[0], self.lib.p264fan_last_error().decode()))
        self._backend = backend
        self.h = self.lib.p264fan_open(rank, world, C.byref(self.t) if world > 1 else None, C.byref(backend) if backend is not None else None, device)
        if not self.h:
            raise RuntimeError("p264fan_open: %s" % self.lib.p264fan_last_error().decode())

    def worker(self):
        if self.lib.p264fan_worker_run(self.h):
            raise RuntimeError("p264fan_worker_run: %s" % self.lib.p264fan_last_error().decode())

    def root(self, streams, max_pictures=0, on_frame=None):
        """streams: list of Annex-B byte strings.  on_frame(stream, picture, y, u, v) for every picture.  Returns the stats."""
        bufs = [(C.c_uint8 * len(s)).from_buffer_copy(s) for s in streams]
        ptrs = (C.c_void_p * len(streams))(*[C.addressof(b) for b in bufs])
        sizes = (C.c_int64 * len(streams))(*[len(s) for s in streams])

        def cb(user, stream, picture, w, h, p):
            if on_frame:
                a = np.ctypeslib.as_array(p, (w * h * 3 // 2,))
                y = a[:w * h].reshape(h, w)
                u = a[w * h:w * h * 5 // 4].reshape(h // 2, w // 2)
                v = a[w * h * 5 // 4:].reshape(h // 2, w // 2)
                on_frame(stream, picture, y, u, v)
        st = FanStats()
        rc = self.lib.p264fan_root_run(self.h, len(streams), ptrs, sizes, max_pictures, FRAME_CB(cb), None, C.byref(st))
        if rc:
            raise RuntimeError("p264fan_root_run: %s" % self.lib.p264fan_last_error().decode())
        return {f: getattr(st, f) for f, _ in st._fields_}

    def close(self):
        if getattr(self, "h", None):
            self.lib.p264fan_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
