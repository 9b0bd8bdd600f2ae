import hashlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    """The product library.  Built in-tree on demand (hipcc cross-compiles without a GPU)."""
    import shutil
    from p264decoder_amd import _native, build
    # build.build() is incremental: a no-op when the library is newer than every source, a rebuild after any edit - a stale
    # binary (the .so is git-ignored but travels to the GPU box) cannot make the suite green
    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc") or not os.path.exists(_native.LIB_PATH):
        build.build()
    return _native.load()


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement of the hot path - the checker, never the product."""
    from tests import oracle_bind
    return oracle_bind.load()


@pytest.fixture(scope="session")
def f26():
    data = open(os.path.join(GOLDEN, "f26.264"), "rb").read()
    assert hashlib.sha256(data).hexdigest() == "4567e811d04ec20701c35cafcfb3b75523ae1433606df137b98a7041a02f3e36"
    return data


@pytest.fixture(scope="session")
def f26_hashes():
    return [l.split()[1] for l in open(os.path.join(GOLDEN, "f26_frames.sha256"))]


def frame_sha256(y, u, v):
    h = hashlib.sha256()
    for p in (y, u, v):
        h.update(p.tobytes())
    return h.hexdigest()
