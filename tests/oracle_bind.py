"""ctypes binding of oracle/liboracle.so (TEST INFRASTRUCTURE).  Builds it with make on demand."""
import ctypes as C
import os
import subprocess

import numpy as np

from p264decoder_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")


def load():
    src = [os.path.join(ORACLE_DIR, f) for f in ("cpu_recon.c", "cpu_recon.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in src):
        subprocess.run(["make", "-C", ORACLE_DIR, "oracle"], check=True, stdout=subprocess.DEVNULL)
    lib = C.CDLL(LIB)
    for f in ("oracle_reconstruct", "oracle_reconstruct_nodeblock", "oracle_deblock_picture"):
        getattr(lib, f).argtypes = [C.POINTER(N.Picture), C.POINTER(C.c_void_p)]
        getattr(lib, f).restype = C.c_int
    return lib


class FrameStore:
    """Host-side frame store for the oracle: slots x (Y,U,V), unpadded, MB-aligned."""

    def __init__(self, mb_w, mb_h, slots):
        w, h = mb_w * 16, mb_h * 16
        self.frames = [[np.zeros((h, w), np.uint8), np.zeros((h // 2, w // 2), np.uint8), np.zeros((h // 2, w // 2), np.uint8)]
                       for _ in range(slots)]
        self.ptrs = (C.c_void_p * (slots * 3))(*[p.ctypes.data for f in self.frames for p in f])

    def __getitem__(self, slot):
        return self.frames[slot]


def reconstruct(lib, store, picture, deblock=True):
    fn = lib.oracle_reconstruct if deblock else lib.oracle_reconstruct_nodeblock
    fn(C.byref(picture.desc), store.ptrs)
    return store[picture.desc.dst_slot]
