"""Stream fan-out with the product backend: two processes (ranks) share the box's MI355X, rank 0 parses and scatters over
the TCP transport, both reconstruct their streams with the HIP kernels, rank 0 gathers - every picture against the real
reference decoder's hashes.  The RCCL transport needs one GPU per rank: test_rccl_two_ranks runs wherever at least two
devices are visible (the driver's multi-GPU node) and skips on a one-GPU box; bench.py's fan-out leg reports the same
exchange at N > 1."""
import os

import pytest

from tests import fan_helpers, synth_cases

pytestmark = pytest.mark.gpu


def test_fanout_two_ranks_one_gpu(lib, f26, f26_hashes):
    cif = synth_cases.stream_bytes("cif_ip")
    cif_h = synth_cases.golden("cif_ip")[1]
    got, st = fan_helpers.run_job(2, [f26, cif, cif, f26], 40, False, 29700 + (os.getpid() % 200))
    assert st["pictures"] == 40 + 24 + 24 + 40 and st["pictures_remote"] == 24 + 40
    for i in range(40):
        assert got[(0, i)] == f26_hashes[i] and got[(3, i)] == f26_hashes[i]
    for i in range(24):
        assert got[(1, i)] == cif_h[i] and got[(2, i)] == cif_h[i]


def test_fanout_1080p(lib):
    hashes = synth_cases.golden("cfg3_1080p_allp")[1]
    data = synth_cases.stream_bytes("cfg3_1080p_allp")
    got, st = fan_helpers.run_job(2, [data, data, data], 4, False, 29900 + (os.getpid() % 90))
    assert st["pictures"] == 12 and st["bytes_gathered"] == 4 * 3133440
    for s in range(3):
        for i in range(4):
            assert got[(s, i)] == hashes[i]


def test_rccl_transport_self_exchange(lib):
    """The RCCL transport on one GPU: a communicator of one rank, a grouped ncclSend / ncclRecv to itself through the
    transport's staging buffers (host -> device -> RCCL -> device -> host).  The multi-GPU exchange uses exactly these calls."""
    import ctypes as C
    import numpy as np
    from p264decoder_amd import fanout
    fanout._proto(lib)
    uid = fanout.rccl_unique_id(lib)
    t = fanout.Transport()
    assert lib.p264fan_rccl_transport(C.byref(t), 0, 1, (C.c_uint8 * 128).from_buffer_copy(uid), 0) == 0
    SEND = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t)
    GRP = C.CFUNCTYPE(C.c_int, C.c_void_p)
    CLOSE = C.CFUNCTYPE(None, C.c_void_p)
    send, recv, begin, end, close = SEND(t.send), SEND(t.recv), GRP(t.group_begin), GRP(t.group_end), CLOSE(t.close)
    rng = np.random.default_rng(5)
    for size in (256, 3133440, 1 << 20):
        a = rng.integers(0, 256, size=size, dtype=np.uint8)
        b = rng.integers(0, 256, size=4096, dtype=np.uint8)
        ra, rb = np.zeros_like(a), np.zeros_like(b)
        assert begin(t.ctx) == 0
        assert send(t.ctx, 0, a.ctypes.data, a.size) == 0 and send(t.ctx, 0, b.ctypes.data, b.size) == 0
        assert recv(t.ctx, 0, ra.ctypes.data, ra.size) == 0 and recv(t.ctx, 0, rb.ctypes.data, rb.size) == 0
        assert end(t.ctx) == 0
        assert np.array_equal(a, ra) and np.array_equal(b, rb)
    close(t.ctx)


def test_rccl_two_ranks(lib):
    """The RCCL transport between two DIFFERENT ranks, one GPU each: ncclCommInitRank(world 2), grouped ncclSend / ncclRecv of
    control blocks, packed pictures, status blocks and planes, with the product backend on both ranks.  Needs two visible
    devices: skipped on a one-GPU box (two ranks cannot share a device in one RCCL communicator)."""
    from p264decoder_amd import device_count, fanout
    if device_count(lib) < 2:
        pytest.skip("the RCCL fan-out between ranks needs at least two GPUs (have %d)" % device_count(lib))
    uid = fanout.rccl_unique_id(lib)
    hashes = synth_cases.golden("cfg3_1080p_allp")[1]
    data = synth_cases.stream_bytes("cfg3_1080p_allp")
    got, st = fan_helpers.run_job(2, [data] * 4, 6, False, 0, transport=("rccl", uid))
    assert st["pictures"] == 24 and st["pictures_remote"] == 12 and st["bytes_gathered"] == 12 * 3133440
    for s in range(4):
        for i in range(6):
            assert got[(s, i)] == hashes[i], "stream %d picture %d differs from the reference decoder" % (s, i)
