"""The multi-stream pipeline end to end on the MI355X: threaded parse into pinned buffers, asynchronous uploads, one
batched reconstruction per round.  Streams of different content and length run side by side; the last picture of
every stream must be the real reference decoder's (committed SHA-256)."""
import os

import numpy as np
import pytest

from p264decoder_amd import Pipeline
from tests import synth_cases
from tests.conftest import frame_sha256

pytestmark = pytest.mark.gpu


def test_pipeline_mixed_streams(lib, f26, f26_hashes):
    cif = synth_cases.stream_bytes("cif_ip")                      # 352x288 like f26, 24 pictures
    _, cif_hashes = synth_cases.golden("cif_ip")
    streams = [f26, cif, cif, f26, cif, f26, cif]
    pipe = Pipeline(streams, threads=4, device=0, lib=lib)
    st = pipe.run()
    assert st["pictures"] == 3 * 300 + 4 * 24 and st["rounds"] == 300
    for i, s in enumerate(streams):
        want = f26_hashes[-1] if s is f26 else cif_hashes[-1]
        assert frame_sha256(*pipe.read_frame(i)) == want, "stream %d" % i
    pipe.close()


def test_pipeline_picture_limit(lib, f26, f26_hashes):
    pipe = Pipeline([f26] * 5, threads=2, device=0, lib=lib)
    st = pipe.run(max_pictures=9)
    assert st["pictures"] == 45
    for i in range(5):
        assert frame_sha256(*pipe.read_frame(i)) == f26_hashes[8]
    pipe.close()


def test_pipeline_1080p(lib):
    data = synth_cases.stream_bytes("cfg3_1080p_allp")
    _, hashes = synth_cases.golden("cfg3_1080p_allp")
    pipe = Pipeline([data] * 6, threads=6, device=0, lib=lib)
    st = pipe.run(max_pictures=10)
    assert st["pictures"] == 60
    for i in range(6):
        assert frame_sha256(*pipe.read_frame(i)) == hashes[9]
    pipe.close()


@pytest.mark.parametrize("mode", ["0", "1", "2", "3"])
def test_host_buffer_kinds(mode):
    """The four kinds of host memory the parsers' picture buffers and the frame downloads can live in (P264AMD_HOST_ALLOC:
    hipHostMalloc coherent / non-coherent, registered 4 KB pages, registered huge pages = the default since round 5): the same
    pictures through the pipeline and through the drop-in API.  The kind is latched at the first allocation of a process, so
    every kind gets a process of its own."""
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import hashlib, os, sys
sys.path.insert(0, %r)
from p264decoder_amd import Decoder, Pipeline
from tests.conftest import frame_sha256
data = open(%r, "rb").read()
hashes = [l.split()[1] for l in open(%r) if l.strip()]
pipe = Pipeline([data, data, data], threads=2, device=0)
st = pipe.run(max_pictures=12)
assert st["pictures"] == 36, st
for s in range(3):
    assert frame_sha256(*pipe.read_frame(s)) == hashes[11], "pipeline stream %%d" %% s
pipe.close()
dec = Decoder()
n = 0
for y, u, v in dec.decode_annexb(data):
    assert frame_sha256(y, u, v) == hashes[n], "drop-in picture %%d" %% n
    n += 1
    if n == 6: break
dec.close()
print("ok")
""" % (ROOT, os.path.join(ROOT, "tests", "golden", "f26.264"), os.path.join(ROOT, "tests", "golden", "f26_frames.sha256"))
    env = dict(os.environ, P264AMD_HOST_ALLOC=mode)
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:]


def test_parsed_pictures_go_up_in_one_copy(lib):
    """The parser lays a picture's arrays out like an input slot (tests/test_input_layout.py): p264hip_upload queues ONE host ->
    HBM copy per P / I picture; arrays that lie anywhere else (a caller's own: here numpy copies) still take five."""
    from p264decoder_amd import HipReconstructor, Parser, _native as N
    import ctypes as C
    h = lib.p264parse_open(1)
    data = synth_cases.stream_bytes("cif_ip")
    hip = None
    want = 0
    for typ, idc, rbsp in N.split_annexb(lib, data):
        pic = C.POINTER(N.Picture)()
        buf = (C.c_uint8 * max(len(rbsp), 1)).from_buffer_copy(rbsp if len(rbsp) else b"\0")
        if lib.p264parse_nal(h, typ, idc, buf, len(rbsp), C.byref(pic)) != 1:
            continue
        d = pic.contents
        if hip is None:
            hip = HipReconstructor(d.mb_w, d.mb_h, n_streams=2, slots=lib.p264parse_slots(h), max_pictures=2, lib=lib)
        assert lib.p264hip_upload(hip.h, 0, pic, 1) == 0
        want += 1
        assert lib.p264hip_upload_copies(hip.h) == want
        hip.reconstruct([0], [0])
    lib.p264parse_close(h)
    # the same stream through owned copies of the arrays (recon.ParsedPicture): five copies each, the same pictures
    pics = Parser(quiet=True, lib=lib).parse_stream(data)
    for p in pics:
        hip.upload(1, [p])
        want += 5
        assert lib.p264hip_upload_copies(hip.h) == want
        hip.reconstruct([1], [1])
    hip.sync()
    for a, b in zip(hip.read_frame(0, pics[-1].desc.dst_slot), hip.read_frame(1, pics[-1].desc.dst_slot)):
        assert np.array_equal(a, b)
    hip.close()
