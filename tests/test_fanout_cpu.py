"""Stream fan-out (include/p264fan.h, BASELINE config 5 / SURVEY 8e) on CPUs: rank 0 owns the Annex-B inputs and the
I420 outputs, parses, scatters the parsed pictures to the ranks that own the streams and gathers the planes - here over
the TCP transport with world sizes 1, 2 and 3, the reconstruction behind the backend interface being the CPU oracle
(no GPU in this suite; the product backend is exercised by tests/test_gpu_fanout.py).  Every gathered picture must hash
to what the REAL reference decoder produced for it (committed golden hashes)."""
import os

import pytest

from tests import fan_helpers, synth_cases


@pytest.mark.parametrize("world", [1, 2, 3])
def test_fanout_tcp_world(lib, f26, f26_hashes, world):
    cif = synth_cases.stream_bytes("cif_ip")
    cif_h = synth_cases.golden("cif_ip")[1]
    qpd = synth_cases.stream_bytes("qpdelta")                   # 11x9 MBs: a different size is a different job
    streams = [cif, cif, cif, cif, cif]                          # 5 streams over 1..3 ranks: uneven shares
    got, st = fan_helpers.run_job(world, streams, 10, True, 29500 + world + (os.getpid() % 500))
    assert st["pictures"] == 50 and st["world"] == world and st["rounds"] == 10
    assert st["pictures_remote"] == sum(10 for s in range(5) if s % world)
    assert st["bytes_gathered"] == st["pictures_remote"] * 352 * 288 * 3 // 2
    for s in range(5):
        for i in range(10):
            assert got[(s, i)] == cif_h[i], "stream %d picture %d differs from the reference decoder" % (s, i)


def test_fanout_streams_of_different_length(lib, f26, f26_hashes):
    cif = synth_cases.stream_bytes("cif_ip")                    # 24 pictures, 352x288 like f26
    cif_h = synth_cases.golden("cif_ip")[1]
    got, st = fan_helpers.run_job(2, [f26, cif, cif], 0, True, 29600 + (os.getpid() % 300))
    assert st["pictures"] == 300 + 24 + 24 and st["rounds"] == 300
    assert all(got[(0, i)] == f26_hashes[i] for i in range(300))
    assert all(got[(1, i)] == cif_h[i] and got[(2, i)] == cif_h[i] for i in range(24))


@pytest.mark.parametrize("failing_rank", [0, 1, 2])
def test_fanout_failure_leaves_nobody_waiting(lib, failing_rank):
    """A backend that fails on purpose in the middle of the job, on the root or on a worker: every rank must come back
    with an error naming the failure (no rank waits inside a round for a message that never comes), the failing worker's
    text reaches the root through the round's status block."""
    cif = synth_cases.stream_bytes("cif_ip")
    res = fan_helpers.run_job(3, [cif] * 6, 8, True, 30100 + failing_rank + (os.getpid() % 300), fail=(failing_rank, 5), expect_errors=True)
    assert len(res) == 3
    by_rank = {}
    for kind, a, b in res:
        if kind == "error":
            by_rank[int(a.split(":")[0])] = a
        elif kind == "worker":
            by_rank[a] = "ok"
        else:
            by_rank[0] = "ok"
    assert by_rank[0] != "ok", "the root must report the failure"
    if failing_rank:
        assert ("worker %d" % failing_rank) in by_rank[0] and by_rank[failing_rank] != "ok"
        assert all(by_rank[r] == "ok" for r in (1, 2) if r != failing_rank), by_rank     # the other worker left in step, on FINISHED
    else:
        assert "root" in by_rank[0] and by_rank[1] == "ok" and by_rank[2] == "ok", by_rank


def test_fanout_rounds_overlap_parse_and_exchange(lib):
    """Rounds are double-buffered: while round r is exchanged and reconstructed (the scalar oracle takes tens of
    milliseconds per 1080p picture), round r+1 is parsed.  The exchange side must have waited for the parser for little
    more than the first round: the rest of the parse time is hidden."""
    data = synth_cases.stream_bytes("cfg3_1080p_allp")
    hashes = synth_cases.golden("cfg3_1080p_allp")[1]
    got, st = fan_helpers.run_job(2, [data] * 4, 8, True, 30500 + (os.getpid() % 300))
    assert st["pictures"] == 32 and all(got[(s, i)] == hashes[i] for s in range(4) for i in range(8))
    assert st["parse_threads"] == 4 and st["rounds"] == 8
    # (the first round cannot be hidden: 1/8 at best; without the overlap the wait is the whole parse time.  0.7 until round 5: the
    #  parser has become faster since and a loaded machine brought the ratio over it now and then)
    assert st["parse_seconds"] > 0 and st["parse_wait_seconds"] < 0.9 * st["parse_seconds"], st


def test_fanout_main_profile_cabac_b_streams(lib, oracle):
    """BASELINE config 5's kind of stream through the fan-out (Main profile, CABAC, I+P+B: list-1 arrays and weights travel in
    the packed pictures): every gathered picture equals what the oracle gives for the stream decoded on its own."""
    import subprocess, hashlib
    from p264decoder_amd import Parser
    from tests import oracle_bind
    from tests.conftest import frame_sha256
    args = "--mbw 8 --mbh 6 --frames 13 --seed 104 --refs 2 --bframes 2 --sub8x8 --implicit --coded 12 --maxlevel 8 --cabac"
    data = open(synth_cases.generate(args), "rb").read()
    parser = Parser(quiet=True, lib=lib)
    pics = parser.parse_stream(data)
    store = oracle_bind.FrameStore(pics[0].mb_w, pics[0].mb_h, parser.slots)
    want = [frame_sha256(*oracle_bind.reconstruct(oracle, store, p)) for p in pics]
    got, st = fan_helpers.run_job(3, [data] * 4, 0, True, 30900 + (os.getpid() % 300))
    assert st["pictures"] == 4 * 13
    for s in range(4):
        for i in range(13):
            assert got[(s, i)] == want[i], "stream %d picture %d" % (s, i)


def test_fanout_symbols_and_errors(lib):
    from p264decoder_amd.fanout import FanOut
    for sym in ("p264fan_open", "p264fan_root_run", "p264fan_worker_run", "p264fan_close", "p264fan_tcp_transport",
                "p264fan_rccl_unique_id", "p264fan_rccl_transport", "p264fan_last_error"):
        assert hasattr(lib, sym)
    with pytest.raises(RuntimeError):
        FanOut(1, 2, ("tcp", "not-an-address", 1), lib=lib)


def test_worker_out_of_step_aborts_the_transport(lib):
    """A worker that cannot stay in step with the root (here: a control block that announces more pictures than a round can
    hold) gives up ON the transport, not just on its loop: the root's side of the connection is shut down at once, so a root
    in the middle of a round gets an error instead of waiting for that worker's status (over RCCL, where a peer cannot
    "close", the same call is ncclCommAbort)."""
    import socket
    import struct
    import threading
    from p264decoder_amd.fanout import FanOut
    port = 30700 + (os.getpid() % 300)
    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    srv.bind(("127.0.0.1", port))
    srv.listen(1)
    result = {}

    def worker():
        try:
            fan = FanOut(1, 2, ("tcp", "127.0.0.1", port), device=0, backend=None, lib=lib)
            try:
                fan.worker()
                result["rc"] = "ok"
            except Exception as e:                              # noqa: BLE001
                result["rc"] = repr(e)
            fan.close()
        except Exception as e:                                  # noqa: BLE001
            result["rc"] = "open: %r" % (e,)
    th = threading.Thread(target=worker)
    th.start()
    conn, _ = srv.accept()
    conn.settimeout(20)
    assert struct.unpack("<i", conn.recv(4))[0] == 1            # the worker introduces itself with its rank
    conn.sendall(struct.pack("<5i64I", 1000, 22, 18, 2, 1, *([0] * 64)))     # fan_ctrl_t with n = 1000 (> 64 per round)
    assert conn.recv(16) == b"", "the worker must shut the connection down, not leave the root waiting"
    th.join(20)
    assert not th.is_alive() and "bad control block" in result.get("rc", ""), result
    conn.close()
    srv.close()
