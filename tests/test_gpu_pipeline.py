"""The multi-stream pipeline end to end on the MI355X: threaded parse into pinned buffers, asynchronous uploads, one
batched reconstruction per round.  Streams of different content and length run side by side; the last picture of
every stream must be the real reference decoder's (committed SHA-256)."""
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
