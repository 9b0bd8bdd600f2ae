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


def test_fanout_device_road_two_ranks_one_gpu(lib, f26, f26_hashes, monkeypatch):
    """The DEVICE ROAD of the protocol (p264fan_transport_t.send_dev / recv_dev + the backend's reserve / reconstruct_reserved /
    planes): the worker receives every picture straight into the input slot it is reconstructed from and sends the planes out
    of the device conversion buffer.  Two ranks share the box's GPU, so the transport is TCP with its device entry points
    switched on (P264AMD_FAN_TCP_DEVICE: a host bounce inside the transport stands in for xGMI); between two GPUs the RCCL
    transport offers the same entry points (test_rccl_transport_self_exchange covers its calls, test_rccl_two_ranks the job)."""
    monkeypatch.setenv("P264AMD_FAN_TCP_DEVICE", "1")
    cif = synth_cases.stream_bytes("cif_ip")
    cif_h = synth_cases.golden("cif_ip")[1]
    got, st = fan_helpers.run_job(2, [f26, cif, cif, f26], 30, False, 30100 + (os.getpid() % 200))
    assert st["pictures"] == 30 + 24 + 24 + 30 and st["pictures_remote"] == 24 + 30
    assert st["device_road_rounds"] == 30, st                     # every round of the one worker (its two streams end after 24 and 30 pictures)
    for i in range(30):
        assert got[(0, i)] == f26_hashes[i] and got[(3, i)] == f26_hashes[i]
    for i in range(24):
        assert got[(1, i)] == cif_h[i] and got[(2, i)] == cif_h[i]
    # B pictures (list-1 arrays and the weight table are part of the block) and 1080p
    from tests.test_input_layout import B_CIF
    from tests import oracle_bind
    from p264decoder_amd import Parser
    b = synth_cases.stream_bytes(B_CIF)
    parser = Parser(quiet=True, lib=lib)
    pics = parser.parse_stream(b)
    ora = oracle_bind.load()
    store = oracle_bind.FrameStore(pics[0].mb_w, pics[0].mb_h, parser.slots)
    import hashlib
    want = []
    for p in pics:
        h = hashlib.sha256()
        for plane in oracle_bind.reconstruct(ora, store, p):
            h.update(plane.tobytes())
        want.append(h.hexdigest())
    data = synth_cases.stream_bytes("cfg3_1080p_allp")
    hashes = synth_cases.golden("cfg3_1080p_allp")[1]
    for streams, n, ref in (([b, b, b], len(pics), want), ([data, data], 3, hashes)):
        got, st = fan_helpers.run_job(2, streams, n, False, 30400 + (os.getpid() % 200))
        assert st["device_road_rounds"] == n
        for s in range(len(streams)):
            for i in range(n):
                assert got[(s, i)] == ref[i], (s, i)


def test_default_road_is_the_host_one_without_device_entry_points(lib, f26, f26_hashes, monkeypatch):
    monkeypatch.delenv("P264AMD_FAN_TCP_DEVICE", raising=False)
    got, st = fan_helpers.run_job(2, [f26, f26], 5, False, 30700 + (os.getpid() % 200))
    assert st["device_road_rounds"] == 0 and all(got[(1, i)] == f26_hashes[i] for i in range(5))


def test_packed_upload_and_reserved_slots_match_the_plain_upload(lib):
    """p264hip_upload_packed (one copy of the packed block) and p264hip_input_reserve / commit (somebody else writes the
    block into the slot: here p264hip_copy_to_device) against p264hip_upload, picture by picture; the device-resident planar
    frame (p264hip_frame_planar_device) against p264hip_read_frame."""
    import ctypes as C
    import numpy as np
    from p264decoder_amd import HipReconstructor, Parser
    from tests.test_input_layout import B_CIF
    parser = Parser(quiet=True, lib=lib)
    pics = parser.parse_stream(synth_cases.stream_bytes(B_CIF))
    hip = HipReconstructor(pics[0].mb_w, pics[0].mb_h, n_streams=3, slots=parser.slots, max_pictures=3, lib=lib)
    for p in pics:
        blk = HipReconstructor.pack(p, lib)
        hip.upload(0, [p])
        hip.upload_packed(1, p, blk)
        dev, n = hip.input_reserve(2, p)
        assert n == blk.size
        with pytest.raises(Exception):
            hip.reconstruct([2], [2])                             # reserved, not committed: the slot is not usable yet
        assert lib.p264hip_copy_to_device(dev, blk.ctypes.data, n) == 0
        hip.input_commit(2)
        hip.reconstruct([0, 1, 2], [0, 1, 2])
        hip.sync()
        f0 = hip.read_frame(0, p.desc.dst_slot)
        for s in (1, 2):
            for a, b in zip(f0, hip.read_frame(s, p.desc.dst_slot)):
                assert np.array_equal(a, b)
        pdev, pn = hip.frame_planar_device(2, p.desc.dst_slot, index=1)
        hip.sync()
        flat = np.empty(pn, np.uint8)
        assert lib.p264hip_copy_from_device(flat.ctypes.data, pdev, pn) == 0
        assert np.array_equal(flat, np.concatenate([a.reshape(-1) for a in f0]))
    hip.close()


def test_a_committed_block_with_bad_records_is_an_error_not_a_fault(lib):
    """The device road does not trust its producer: a block written into a reserved slot whose records point outside its
    coefficient stream (a peer that packed it wrongly) must fail the batch with an error - p264hip_upload's host check, run by
    a kernel at commit - and the context must go on working."""
    import ctypes as C
    import numpy as np
    from p264decoder_amd import HipReconstructor, Parser
    from p264decoder_amd import _native as N
    parser = Parser(quiet=True, lib=lib)
    pics = parser.parse_stream(synth_cases.stream_bytes("cif_ip"))[:2]
    hip = HipReconstructor(pics[0].mb_w, pics[0].mb_h, n_streams=2, slots=parser.slots, max_pictures=2, lib=lib)
    p = pics[0]
    blk = HipReconstructor.pack(p, lib).copy()
    # record 5: coefficient blocks far beyond the stream (coef_index is the third dword of the 16-byte record)
    words = blk[:p.mb_w * p.mb_h * 16].view(np.uint32).reshape(-1, 4)
    coded = int(np.flatnonzero(words[:, 1] != 0)[0])
    words[coded, 2] = 0x7fffff00
    dev, n = hip.input_reserve(1, p)
    assert lib.p264hip_copy_to_device(dev, blk.ctypes.data, n) == 0
    hip.input_commit(1)
    with pytest.raises(Exception, match="outside coefs"):
        hip.reconstruct([1], [1])
    hip.sync()
    # the same slot, the intact block: decodes like the plain upload
    good = HipReconstructor.pack(p, lib)
    dev, n = hip.input_reserve(1, p)
    assert lib.p264hip_copy_to_device(dev, good.ctypes.data, n) == 0
    hip.input_commit(1)
    hip.upload(0, [p])
    hip.reconstruct([0, 1], [0, 1])
    hip.sync()
    for a, b in zip(hip.read_frame(0, p.desc.dst_slot), hip.read_frame(1, p.desc.dst_slot)):
        assert np.array_equal(a, b)
    hip.close()


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
    # device buffers: ncclSend / ncclRecv straight from / into them (send_dev / recv_dev), mixed with staged host messages.
    # (Two planar frame buffers of a 1080p context serve as the device memory: what a worker sends out of.)
    from p264decoder_amd import HipReconstructor
    assert t.send_dev and t.recv_dev
    send_dev, recv_dev = SEND(t.send_dev), SEND(t.recv_dev)
    hip = HipReconstructor(120, 68, n_streams=1, slots=2, max_pictures=1, lib=lib)
    (src, n), (dst, _) = hip.frame_planar_device(0, 0, index=0), hip.frame_planar_device(0, 1, index=1)
    hip.sync()
    assert n == 3133440
    data = rng.integers(0, 256, size=n, dtype=np.uint8)
    back = np.zeros(n, np.uint8)
    assert lib.p264hip_copy_to_device(src, data.ctypes.data, n) == 0 and lib.p264hip_copy_to_device(dst, back.ctypes.data, n) == 0
    ctrl, rctrl = rng.integers(0, 256, size=272, dtype=np.uint8), np.zeros(272, np.uint8)
    assert begin(t.ctx) == 0
    assert send(t.ctx, 0, ctrl.ctypes.data, ctrl.size) == 0 and send_dev(t.ctx, 0, src, n) == 0
    assert recv(t.ctx, 0, rctrl.ctypes.data, rctrl.size) == 0 and recv_dev(t.ctx, 0, dst, n) == 0
    assert end(t.ctx) == 0
    assert lib.p264hip_copy_from_device(back.ctypes.data, dst, n) == 0
    assert np.array_equal(ctrl, rctrl) and np.array_equal(data, back)
    hip.close()
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


@pytest.mark.parametrize("device_road", [False, True])
def test_fanout_config5_stream_two_ranks_one_gpu(lib, monkeypatch, device_road):
    """BASELINE config 5's own kind of stream - 1080p Main profile, CABAC, I + P + B - through two ranks with the product
    backend on both: parsed on rank 0 (CABAC), scattered with its list-1 arrays and weight tables, reconstructed by the HIP
    kernels of both ranks, gathered, every picture against the committed oracle hashes (the reference cannot decode it).  Host
    road and device road."""
    if device_road:
        monkeypatch.setenv("P264AMD_FAN_TCP_DEVICE", "1")
    else:
        monkeypatch.delenv("P264AMD_FAN_TCP_DEVICE", raising=False)
    name = "main_1080p_cabac_ipb"
    data = open(synth_cases.generate(synth_cases.ORACLE_CASES[name]), "rb").read()
    hashes = synth_cases.oracle_golden(name)[1]
    n = len(hashes)
    got, st = fan_helpers.run_job(2, [data, data, data], n, False, 31000 + (os.getpid() % 200))
    assert st["pictures"] == 3 * n and st["pictures_remote"] == n and st["bytes_gathered"] == n * 3133440
    assert st["device_road_rounds"] == (n if device_road else 0)
    for s in range(3):
        for i in range(n):
            assert got[(s, i)] == hashes[i], "stream %d picture %d differs from the oracle" % (s, i)


def test_a_clone_takes_the_verdict_of_the_record_check_along(lib):
    """k_check_records' verdict lives per input slot: p264hip_clone_picture of a block a device producer committed must carry it
    to the clone (a bad block's clone is an error too, not an out-of-bounds walk through coefs[]), and a clone of a good block
    must not inherit what an earlier, bad tenant of the destination slot left behind."""
    import numpy as np
    from p264decoder_amd import HipReconstructor, Parser
    parser = Parser(quiet=True, lib=lib)
    p = parser.parse_stream(synth_cases.stream_bytes("cif_ip"))[0]
    hip = HipReconstructor(p.mb_w, p.mb_h, n_streams=3, slots=parser.slots, max_pictures=4, lib=lib)
    good = HipReconstructor.pack(p, lib)
    bad = good.copy()
    words = bad[:p.mb_w * p.mb_h * 16].view(np.uint32).reshape(-1, 4)
    words[int(np.flatnonzero(words[:, 1] != 0)[0]), 2] = 0x7fffff00

    def commit(slot, blk):
        dev, n = hip.input_reserve(slot, p)
        assert lib.p264hip_copy_to_device(dev, blk.ctypes.data, n) == 0
        hip.input_commit(slot)
    commit(1, bad)
    hip.clone_picture(2, 1)                                       # unchecked and bad: the clone is as bad
    with pytest.raises(Exception, match="outside coefs"):
        hip.reconstruct([2], [2])
    hip.sync()
    with pytest.raises(Exception, match="outside coefs"):         # (and so is the original, still)
        hip.reconstruct([1], [1])
    hip.sync()
    # slot 2 has held a bad block; a good block committed into slot 3 and cloned into slot 2 must decode
    commit(3, good)
    hip.clone_picture(2, 3)
    hip.upload(0, [p])
    hip.reconstruct([0, 2], [0, 2])
    hip.sync()
    for a, b in zip(hip.read_frame(0, p.desc.dst_slot), hip.read_frame(2, p.desc.dst_slot)):
        assert np.array_equal(a, b)
    hip.close()


def test_compact_uploads_match_the_plain_upload(lib):
    """p264hip_upload_compact (the compact link format, include/p264hip.h) against p264hip_upload, picture by picture and as a
    batch: the blocks of several pictures wait for ONE expansion kernel (k_expand_compact) in front of the reconstruct call;
    parsed streams (all vector shapes, Intra4x4 modes, narrow levels) and seam-level random pictures with levels up to the
    int16 limits and sub-8x8 vectors; a slot that gets new content by another road before its block was expanded; a clone of
    a slot whose block has not been expanded yet."""
    import numpy as np
    from p264decoder_amd import HipReconstructor, Parser
    from tests import seam_fuzz
    parser = Parser(quiet=True, lib=lib)
    pics = parser.parse_stream(synth_cases.stream_bytes("cif_ip"))[:10]
    S = 3
    hip = HipReconstructor(pics[0].mb_w, pics[0].mb_h, n_streams=S + 1, slots=parser.slots, max_pictures=S + 2, lib=lib)
    for i, p in enumerate(pics):
        blk = HipReconstructor.pack_compact(p, lib)
        hip.upload(0, [p])
        for s in range(1, S):
            hip.upload_compact(s, p, blk)
        if i == 3:                                              # superseded before it was expanded: the plain upload must win
            hip.upload_compact(1, pics[0], HipReconstructor.pack_compact(pics[0], lib))
            hip.upload(1, [p])
        if i == 5:                                              # two compact blocks for one slot, no expansion in between: the later one counts
            hip.upload_compact(2, pics[0], HipReconstructor.pack_compact(pics[0], lib))
            hip.upload_compact(2, p, blk)
        hip.clone_picture(S, 2)                                 # (expands what is pending first)
        hip.reconstruct([0, 1, 2, S], [0, 1, 2, S])
        hip.sync()
        f0 = hip.read_frame(0, p.desc.dst_slot)
        for s in (1, 2, S):
            for a, b in zip(f0, hip.read_frame(s, p.desc.dst_slot)):
                assert np.array_equal(a, b), "picture %d stream %d" % (i, s)
    hip.close()
    # seam-level pictures: every level style, sub-8x8 partitions, several references
    rng = np.random.default_rng(99)
    mb_w, mb_h = 9, 6
    hip = HipReconstructor(mb_w, mb_h, n_streams=2, slots=3, max_pictures=2, lib=lib)
    for s in range(2):
        for slot in range(3):
            hip.write_frame(s, slot, *seam_fuzz.random_frame(np.random.default_rng(5 + slot), mb_w, mb_h))
    for k in range(12):
        p = seam_fuzz.make_picture(rng, mb_w, mb_h, p_picture=(k != 3), b_picture=(k >= 8), n_ref=2, n_ref_l1=2, slots=3, dst_slot=k % 3,
                                   level_style=["small", "large", "wrap", "mixed"][k % 4], sub8x8=True)
        hip.upload(0, [p])
        hip.upload_compact(1, p, HipReconstructor.pack_compact(p, lib))
        hip.reconstruct([0, 1], [0, 1])
        hip.sync()
        for a, b in zip(hip.read_frame(0, p.desc.dst_slot), hip.read_frame(1, p.desc.dst_slot)):
            assert np.array_equal(a, b), "seam picture %d" % k
    hip.close()
    # a parsed Main-profile stream (CABAC, I + P + B, implicit weights, direct prediction): compact against plain, picture by picture
    from tests.test_input_layout import B_CIF
    parser = Parser(quiet=True, lib=lib)
    pics = parser.parse_stream(synth_cases.stream_bytes(B_CIF + " --cabac"))
    assert sum(p.desc.slice_type == 1 for p in pics) >= 3
    hip = HipReconstructor(pics[0].mb_w, pics[0].mb_h, n_streams=2, slots=parser.slots, max_pictures=2, lib=lib)
    for i, p in enumerate(pics):
        hip.upload(0, [p])
        hip.upload_compact(1, p, HipReconstructor.pack_compact(p, lib))
        hip.reconstruct([0, 1], [0, 1])
        hip.sync()
        for a, b in zip(hip.read_frame(0, p.desc.dst_slot), hip.read_frame(1, p.desc.dst_slot)):
            assert np.array_equal(a, b), "Main-profile picture %d (slice type %d)" % (i, p.desc.slice_type)
    hip.close()
    # 1080p: the golden all-P stream through compact uploads only, against the reference decoder's hashes
    from tests.conftest import frame_sha256
    parser = Parser(quiet=True, lib=lib)
    pics = parser.parse_stream(synth_cases.stream_bytes("cfg3_1080p_allp"), limit=4)
    hashes = synth_cases.golden("cfg3_1080p_allp")[1]
    hip = HipReconstructor(120, 68, n_streams=1, slots=parser.slots, max_pictures=1, lib=lib)
    sizes = []
    for i, p in enumerate(pics):
        blk = HipReconstructor.pack_compact(p, lib)
        sizes.append((blk.size, HipReconstructor.pack(p, lib).size))
        hip.upload_compact(0, p, blk)
        hip.reconstruct([0], [0])
        assert frame_sha256(*hip.read_frame(0, p.desc.dst_slot)) == hashes[i], "1080p picture %d through the compact format" % i
    hip.close()
    assert all(c < 0.45 * f for c, f in sizes[1:]), sizes          # a P picture of the bench's kind: well under half its slot layout


def test_an_inconsistent_compact_block_is_a_wrong_picture_not_a_fault(lib):
    """p264hip_upload_compact checks a block's HEADER; what the block's bits imply is clamped to the block's sections on the device
    (kernel_expand.h).  Blocks whose header is fine but whose shape / flag bits claim far more vectors, mode entries and 16-bit
    levels than their sections hold must go through upload + expansion + reconstruction without an error from the device, and the
    context must decode the intact block right afterwards.  (p264hip_compact_check refuses every one of them: an untrusted
    producer's blocks go through it first.)"""
    import ctypes as C
    import numpy as np
    from p264decoder_amd import HipReconstructor, Parser, _native as N
    parser = Parser(quiet=True, lib=lib)
    pics = parser.parse_stream(synth_cases.stream_bytes("cif_ip"))[:3]
    hip = HipReconstructor(pics[0].mb_w, pics[0].mb_h, n_streams=2, slots=parser.slots, max_pictures=2, lib=lib)
    for p in pics:
        good = HipReconstructor.pack_compact(p, lib)
        hdr = N.CompactHdr.from_buffer_copy(good[:128].tobytes())
        n = p.desc.mb_w * p.desc.mb_h
        for what, lo, size, value in (("every macroblock sixteen vectors", hdr.list[0].off_shape, (n + 3) // 4, 0xff),
                                      ("every macroblock an Intra4x4 entry", hdr.off_i4flag, (n + 7) // 8, 0xff),
                                      ("every block sixteen-bit levels", hdr.off_lvflag, (p.desc.n_coef_blocks + 7) // 8, 0x00),
                                      ("every block eight-bit levels", hdr.off_lvflag, (p.desc.n_coef_blocks + 7) // 8, 0xff)):
            bad = good.copy()
            bad[lo:lo + size] = value
            if what == "every block eight-bit levels" and hdr.level_bytes == 16 * p.desc.n_coef_blocks:
                continue                                       # (every block of this picture is narrow already: only padding bits would change)
            assert lib.p264hip_compact_header_ok(C.byref(p.desc), bad.ctypes.data, bad.size) == 1, what
            assert lib.p264hip_compact_check(C.byref(p.desc), bad.ctypes.data, bad.size) != 0, what
            hip.upload_compact(1, p, bad)
            hip.reconstruct([1], [1])                          # a wrong picture in stream 1's store; no fault, no error
            hip.sync()
        hip.upload(0, [p])
        hip.upload_compact(1, p, good)
        # stream 1's store holds garbage references by now: give both streams the same ones
        for slot in range(parser.slots):
            hip.write_frame(1, slot, *hip.read_frame(0, slot))
        hip.reconstruct([0, 1], [0, 1])
        hip.sync()
        for a, b in zip(hip.read_frame(0, p.desc.dst_slot), hip.read_frame(1, p.desc.dst_slot)):
            assert np.array_equal(a, b)
    hip.close()
