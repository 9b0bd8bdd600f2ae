"""The command-line decoder: `p264decoder_amd -d f26.264 rec.yuv` must write exactly the
pictures of the reference's `p264decoder -d` (p264decoder.c:126-156 - MB-aligned planar I420, decode order),
checked frame by frame against the committed SHA-256 of the real reference decoder."""
import hashlib
import os
import subprocess

import pytest

from p264decoder_amd import build as _build
from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu

CLI = os.path.join(os.path.dirname(_build.__file__), "tools", "p264decoder_amd")


def test_cli_f26_matches_reference(tmp_path, f26_hashes):
    assert os.path.exists(CLI), "tools/p264decoder_amd is built by p264decoder_amd.build"
    out = tmp_path / "rec.yuv"
    r = subprocess.run([CLI, "-d", os.path.join(GOLDEN, "f26.264"), str(out)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "decoded total 300 frames" in r.stderr and "decoding speed:" in r.stderr
    data = out.read_bytes()
    frame = 352 * 288 * 3 // 2
    assert len(data) == 300 * frame
    for i in range(300):
        assert hashlib.sha256(data[i * frame:(i + 1) * frame]).hexdigest() == f26_hashes[i], "picture %d" % i


def test_cli_usage_and_errors(tmp_path):
    r = subprocess.run([CLI], stderr=subprocess.PIPE, text=True)
    assert r.returncode != 0 and "-d <test.264> [recon.yuv] [origin.yuv]" in r.stderr
    bad = tmp_path / "bad.264"
    bad.write_bytes(b"\x00\x00\x01\x67rest")             # 3-byte start code first: the reference refuses it (p264decoder.c:219)
    r = subprocess.run([CLI, "-d", str(bad)], stderr=subprocess.PIPE, text=True)
    assert r.returncode != 0 and "confirm the first start code failed" in r.stderr
    r = subprocess.run([CLI, "-d", str(tmp_path / "missing.264")], stderr=subprocess.PIPE, text=True)
    assert r.returncode != 0 and "open h264 stream file" in r.stderr
