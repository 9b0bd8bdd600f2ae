// fan_wait.h - the bounded wait of the RCCL transport (fan_rccl.hip), free of HIP and RCCL so that it can be tested on a CPU
// with stubs (tests/test_fan_wait.py).
//
// RCCL has no "peer closed": a rank whose peer died would sit in the stream forever.  So the transport never blocks in a HIP
// call behind a transfer; it polls the stream against a deadline and the communicator's asynchronous error, and only when
// the transfers are known to be complete does it touch what they delivered.
#pragma once

enum { FAN_WAIT_DONE = 0, FAN_WAIT_STREAM_ERROR = 1, FAN_WAIT_COMM_ERROR = 2, FAN_WAIT_TIMEOUT = 3 };

// query():      0 = everything on the stream has completed, 1 = not yet, anything else = a stream error (returned in *code)
// async_err():  0 = the communicator is healthy, anything else = its asynchronous error (returned in *code)
// now():        seconds, monotonic;   nap(): called between polls once the first `spin_polls` polls have not been enough
template <class Query, class AsyncErr, class Clock, class Nap>
static inline int fan_bounded_wait(Query query, AsyncErr async_err, Clock now, Nap nap, double timeout_s, unsigned spin_polls, int *code)
{
    const double t0 = now();
    unsigned polls = 0;
    for (;;) {
        const int q = query();
        if (q == 0) return FAN_WAIT_DONE;
        if (q != 1) { *code = q; return FAN_WAIT_STREAM_ERROR; }
        const int ae = async_err();
        if (ae != 0) { *code = ae; return FAN_WAIT_COMM_ERROR; }
        if (now() - t0 > timeout_s) return FAN_WAIT_TIMEOUT;
        if (++polls > spin_polls) nap();
    }
}
