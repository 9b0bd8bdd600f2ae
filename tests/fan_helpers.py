"""Helpers of the fan-out tests: a reconstruction backend made of the CPU oracle (TEST INFRASTRUCTURE - the product's
backend is the MI355X path; the oracle only stands in for it where no GPU exists) and the per-rank entry point."""
import ctypes as C
import hashlib

import numpy as np

from p264decoder_amd import _native as N
from p264decoder_amd.fanout import BK_CLOSE, BK_OPEN, BK_RECON, Backend, FanOut


def oracle_backend(fail_at=None, delay=0.0):
    """p264fan_backend_t whose reconstruct() is oracle_reconstruct on host planes (one frame store per local stream).
    fail_at = n: the n-th reconstruct call of this rank fails on purpose (the failure-path tests); delay: seconds of sleep
    per call (stands in for reconstruction time in the overlap test)."""
    import time
    from tests import oracle_bind
    ora = oracle_bind.load()
    state = {}
    calls = [0]

    def bk_open(ctx, device, mb_w, mb_h, n_local, slots):
        state.update(mb_w=mb_w, mb_h=mb_h, stores=[oracle_bind.FrameStore(mb_w, mb_h, slots) for _ in range(n_local)])
        ctx[0] = 1
        return 0

    def bk_recon(ctx, s, pic, out):
        calls[0] += 1
        if fail_at is not None and calls[0] == fail_at:
            return -1
        if delay:
            time.sleep(delay)
        store = state["stores"][s]
        ora.oracle_reconstruct(pic, store.ptrs)
        w, h = state["mb_w"] * 16, state["mb_h"] * 16
        dst = np.ctypeslib.as_array(out, (w * h * 3 // 2,))
        y, u, v = store[pic.contents.dst_slot]
        dst[:w * h] = y.reshape(-1); dst[w * h:w * h * 5 // 4] = u.reshape(-1); dst[w * h * 5 // 4:] = v.reshape(-1)
        return 0

    def bk_close(ctx):
        state.clear()
    cbs = (BK_OPEN(bk_open), BK_RECON(bk_recon), BK_CLOSE(bk_close))
    b = Backend(None, *cbs)
    b._keep = cbs
    return b


def run_rank(rank, world, port, streams, max_pictures, use_oracle, q, fail=None, delay=0.0, transport=None):
    """Entry point of one rank (spawned process).  The root returns {(stream, picture): sha256} through the queue.
    fail = (rank, n): that rank's n-th reconstruct call fails on purpose.  transport: None = TCP on `port`, or
    ("rccl", unique id): one GPU per rank (device = rank)."""
    try:
        lib = N.load()
        bk = oracle_backend(fail_at=fail[1] if fail and fail[0] == rank else None, delay=delay) if use_oracle else None
        fan = FanOut(rank, world, transport or ("tcp", "127.0.0.1", port), device=rank if transport else 0, backend=bk, lib=lib)
        if rank == 0:
            got = {}

            def on_frame(s, i, y, u, v):
                h = hashlib.sha256()
                for p in (y, u, v):
                    h.update(p.tobytes())
                got[(s, i)] = h.hexdigest()
            st = fan.root(streams, max_pictures=max_pictures, on_frame=on_frame)
            q.put(("ok", got, st))
        else:
            fan.worker()
            q.put(("worker", rank, None))
        fan.close()
    except Exception as e:                                     # noqa: BLE001 - reported to the parent
        q.put(("error", "%d: %r" % (rank, e), None))


def free_port(hint):
    """`hint` if nobody holds it (not even a socket of an earlier run in TIME_WAIT), else a port the kernel picks: two test
    runs shortly after one another - or a port the rendezvous of another test left behind - otherwise fail now and then."""
    import socket
    for want in (hint, 0):
        try:
            with socket.socket() as s:
                s.bind(("127.0.0.1", want))
                return s.getsockname()[1]
        except OSError:
            continue
    return hint


def run_job(world, streams, max_pictures, use_oracle, port, fail=None, delay=0.0, transport=None, expect_errors=False):
    """Runs one job with `world` processes.  Every rank must come back (a hang fails the test by time-out) and exit.
    expect_errors: return the raw per-rank results instead of asserting that nobody failed."""
    import multiprocessing as mp
    if transport is None:
        port = free_port(port)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=run_rank, args=(r, world, port, streams, max_pictures, use_oracle, q, fail, delay, transport)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode is not None, "a rank did not exit"
    if expect_errors:
        return results
    errs = [r for r in results if r[0] == "error"]
    assert not errs, errs
    return [r for r in results if r[0] == "ok"][0][1:]
