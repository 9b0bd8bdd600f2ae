"""The RCCL transport's bounded wait (csrc/hip/fan_wait.h) against stubs, on the CPU: the deadline, the communicator's
asynchronous error and a stream error each end the wait; a stream that completes ends it with success.  (The wait itself
cannot be provoked on the one-GPU box - a dead peer needs a second device - so its logic is factored out of fan_rccl.hip,
which only plugs hipStreamQuery / ncclCommGetAsyncError into it.)"""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include <stdio.h>
#include "fan_wait.h"
int main(void) {
    int code = 0, fails = 0;
    double clock = 0.0; int naps = 0, polls = 0;
    auto now = [&]() { return clock; };
    auto nap = [&]() { naps++; clock += 0.001; };
    // 1. completes after 5000 polls: success, the first 2000 polls spin, the rest nap
    polls = 0; naps = 0; clock = 0;
    int w = fan_bounded_wait([&]() { return ++polls > 5000 ? 0 : 1; }, []() { return 0; }, now, nap, 30.0, 2000, &code);
    if (w != FAN_WAIT_DONE || naps != 5000 - 2000) { printf("case 1: %d naps %d\n", w, naps); fails++; }
    // 2. never completes: the deadline (simulated clock), no sooner
    polls = 0; naps = 0; clock = 0;
    w = fan_bounded_wait([&]() { polls++; return 1; }, []() { return 0; }, now, nap, 2.0, 10, &code);
    if (w != FAN_WAIT_TIMEOUT || clock <= 2.0 || clock > 2.01) { printf("case 2: %d clock %f\n", w, clock); fails++; }
    // 3. the communicator reports an asynchronous error on the third poll
    polls = 0; clock = 0; code = 0;
    w = fan_bounded_wait([&]() { polls++; return 1; }, [&]() { return polls == 3 ? 7 : 0; }, now, nap, 30.0, 2000, &code);
    if (w != FAN_WAIT_COMM_ERROR || code != 7 || polls != 3) { printf("case 3: %d code %d polls %d\n", w, code, polls); fails++; }
    // 4. the stream itself fails
    code = 0;
    w = fan_bounded_wait([&]() { return 0x10000 + 719; }, []() { return 0; }, now, nap, 30.0, 2000, &code);
    if (w != FAN_WAIT_STREAM_ERROR || code != 0x10000 + 719) { printf("case 4: %d code %d\n", w, code); fails++; }
    // 5. already complete: no poll of the communicator, no nap
    naps = 0; int asked = 0;
    w = fan_bounded_wait([]() { return 0; }, [&]() { asked++; return 0; }, now, nap, 30.0, 2000, &code);
    if (w != FAN_WAIT_DONE || asked || naps) { printf("case 5: %d\n", w); fails++; }
    printf(fails ? "FAILED\n" : "ok\n");
    return fails;
}
'''


def test_bounded_wait_against_stubs():
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "t.cpp")
        open(src, "w").write(SRC)
        exe = os.path.join(td, "t")
        subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "p264decoder_amd", "csrc", "hip"), src, "-o", exe], check=True)
        r = subprocess.run([exe], stdout=subprocess.PIPE, text=True)
        assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout


def test_the_transport_waits_before_it_copies():
    """finish() must poll the stream BEFORE it queues the device -> host copies of what was received (a copy into pageable
    memory queued behind a receive blocks inside hipMemcpyAsync: with a dead peer the deadline was never reached)."""
    text = open(os.path.join(ROOT, "p264decoder_amd", "csrc", "hip", "fan_rccl.hip")).read()
    body = text[text.index("int finish(Rccl *r)"):text.index("int rc_send(")]
    assert body.index("fan_bounded_wait(") < body.index("hipMemcpyAsync(")
    assert "hipMemcpyAsync" not in text[text.index("void rc_abort("):text.index("void *stage_buf(")]
