"""Multi-stream decode pipeline (include/p264pipe.h): threaded host parse feeding batched MI355X reconstruction."""
import ctypes as C

import numpy as np

from . import _native as N


class Pipeline:
    """N Annex-B streams decoded side by side.  device=-1 runs the parsers only (no GPU)."""

    def __init__(self, streams, threads=8, device=0, lib=None):
        self.lib = lib or N.load()
        self._bufs = [(C.c_uint8 * len(s)).from_buffer_copy(s) for s in streams]      # borrowed by the C side
        self.h = self.lib.p264pipe_open(device, len(streams), threads)
        if not self.h:
            raise RuntimeError("p264pipe_open failed (no HIP device? the reconstruction has no CPU fallback; device=-1 parses only)")
        for i, b in enumerate(self._bufs):
            if self.lib.p264pipe_set_input(self.h, i, b, len(b)):
                raise RuntimeError("p264pipe_set_input(%d) failed" % i)

    def run(self, max_pictures=0):
        st = N.PipeStats()
        if self.lib.p264pipe_run(self.h, max_pictures, C.byref(st)):
            raise RuntimeError("p264pipe_run failed")
        return {f: getattr(st, f) for f, _ in st._fields_}

    def pictures(self, stream):
        return self.lib.p264pipe_stream_pictures(self.h, stream)

    def read_frame(self, stream):
        w, h = C.c_int(), C.c_int()
        if self.lib.p264pipe_frame_size(self.h, C.byref(w), C.byref(h)):
            raise RuntimeError("no picture decoded yet")
        y = np.empty((h.value, w.value), np.uint8); u = np.empty((h.value // 2, w.value // 2), np.uint8); v = np.empty_like(u)
        if self.lib.p264pipe_read_frame(self.h, stream, y.ctypes.data, w.value, u.ctypes.data, v.ctypes.data, w.value // 2):
            raise RuntimeError("p264pipe_read_frame(%d) failed" % stream)
        return y, u, v

    def close(self):
        if self.h:
            self.lib.p264pipe_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
