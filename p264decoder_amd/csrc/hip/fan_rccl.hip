// fan_rccl.hip - the RCCL transport of the stream fan-out (include/p264fan.h): grouped ncclSend / ncclRecv over xGMI,
// one process per GPU.  librccl is loaded on demand (dlopen), so the library itself does not depend on it.  The
// interface's send / recv hand over host buffers; they are staged through device memory (a send copies host -> device
// and posts ncclSend, a receive posts ncclRecv and copies device -> host once the group has completed).  send_dev / recv_dev take
// device buffers as they are: a worker's pictures and planes never touch its host memory.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <vector>
#include "p264fan.h"
#include "fan_wait.h"

extern "C" int p264fan_set_error(const char *fmt, ...);       // fanout.c: the message p264fan_last_error() returns

namespace {
struct Uid { char internal[128]; };                       // ncclUniqueId (rccl.h:43)
typedef void *Comm;
typedef int (*fn_uid)(Uid *);
typedef int (*fn_init)(Comm *, int, Uid, int);
typedef int (*fn_sr)(void *, size_t, int, int, Comm, hipStream_t);
typedef int (*fn_v)();
typedef int (*fn_destroy)(Comm);
typedef const char *(*fn_err)(int);
typedef int (*fn_async)(Comm, int *);
struct Api { void *lib = nullptr; fn_uid uid; fn_init init; fn_sr send, recv; fn_v gstart, gend; fn_destroy destroy, abort; fn_err err; fn_async async_err; };
Api g_api;
const char *nccl_str(int e) { return g_api.err ? g_api.err(e) : "RCCL error"; }
#define RFAIL(what, e) p264fan_set_error("rccl transport: %s: %s (%d)", what, nccl_str(e), e)
#define HFAIL(what, e) p264fan_set_error("rccl transport: %s: %s", what, hipGetErrorString(e))
bool load_api()
{
    if (g_api.lib) return true;
    // an RCCL that is already part of the process (PyTorch brings its own) first: two instances must not be mixed
    void *l = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!l) l = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!l) l = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!l) l = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!l) l = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!l) { p264fan_set_error("rccl transport: cannot load librccl.so: %s", dlerror()); return false; }
    g_api.uid = (fn_uid)dlsym(l, "ncclGetUniqueId"); g_api.init = (fn_init)dlsym(l, "ncclCommInitRank");
    g_api.send = (fn_sr)dlsym(l, "ncclSend"); g_api.recv = (fn_sr)dlsym(l, "ncclRecv");
    g_api.gstart = (fn_v)dlsym(l, "ncclGroupStart"); g_api.gend = (fn_v)dlsym(l, "ncclGroupEnd");
    g_api.destroy = (fn_destroy)dlsym(l, "ncclCommDestroy"); g_api.abort = (fn_destroy)dlsym(l, "ncclCommAbort");
    g_api.err = (fn_err)dlsym(l, "ncclGetErrorString");
    g_api.async_err = (fn_async)dlsym(l, "ncclCommGetAsyncError");
    if (!g_api.uid || !g_api.init || !g_api.send || !g_api.recv || !g_api.gstart || !g_api.gend || !g_api.destroy) { dlclose(l); p264fan_set_error("rccl transport: librccl.so lacks a symbol"); return false; }
    g_api.lib = l;
    return true;
}
struct Pending { void *host; void *dev; size_t bytes; };    // a receive whose data still sits in its staging buffer
struct Rccl {
    Comm comm = nullptr; int device = 0; hipStream_t stream = nullptr;
    std::vector<void *> stage; std::vector<size_t> cap; size_t used = 0;      // staging buffers of the current group
    std::vector<Pending> pending;
    bool in_group = false, broken = false;
    double timeout_s = 30.0;                                  // P264AMD_FAN_TIMEOUT_S: longest wait for a group before the communicator is aborted
};
double now_s() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
// ncclCommAbort tears down THIS rank's communicator: its own pending operations fail instead of waiting.  The peers are not
// told - each of them ends through its own deadline (P264AMD_FAN_TIMEOUT_S) or its communicator's asynchronous error.
void rc_abort(void *c)
{
    Rccl *r = (Rccl *)c;
    if (!r || !r->comm) return;
    (void)hipSetDevice(r->device);
    if (g_api.abort) g_api.abort(r->comm);
    r->comm = nullptr; r->broken = true;
    // what the aborted operations left on the stream drains now (bounded: no transfer of ours targets host memory before its
    // group has completed, so nothing can land in a buffer the caller has given up - this only keeps the stream reusable)
    const double t0 = now_s();
    while (hipStreamQuery(r->stream) == hipErrorNotReady && now_s() - t0 < 2.0) usleep(200);
}
void *stage_buf(Rccl *r, size_t bytes)
{
    if (r->used == r->stage.size()) { r->stage.push_back(nullptr); r->cap.push_back(0); }
    if (r->cap[r->used] < bytes) {
        if (r->stage[r->used]) (void)hipFree(r->stage[r->used]);
        r->stage[r->used] = nullptr; r->cap[r->used] = 0;
        hipError_t e = hipMalloc(&r->stage[r->used], bytes + bytes / 4);
        if (e != hipSuccess) { HFAIL("hipMalloc of a staging buffer", e); return nullptr; }
        r->cap[r->used] = bytes + bytes / 4;
    }
    return r->stage[r->used++];
}
// The group's transfers are on the stream.  WAIT FIRST, COPY AFTERWARDS: the stream is polled against the deadline and the
// communicator's asynchronous error (fan_wait.h) with nothing but the transfers on it, and the device -> host copies of
// what was received are issued only once the receives have completed.  (Queued behind the receives, a copy into pageable
// host memory - control blocks and statuses live on stacks and in malloc memory - blocks INSIDE hipMemcpyAsync until the
// receive in front of it completes: with a dead peer the deadline below would never be reached, and a copy queued before
// an abort could land in a buffer its owner has left.)
int finish(Rccl *r)
{
    int code = 0;
    const int w = fan_bounded_wait(
        [&]() { const hipError_t q = hipStreamQuery(r->stream); return q == hipSuccess ? 0 : q == hipErrorNotReady ? 1 : 0x10000 + (int)q; },
        [&]() { int ae = 0; return (g_api.async_err && r->comm && g_api.async_err(r->comm, &ae) == 0) ? ae : 0; },
        now_s, []() { usleep(200); }, r->timeout_s, 2000 /* the first polls spin: a round's transfers take well under a millisecond */, &code);
    if (w != FAN_WAIT_DONE) {
        r->pending.clear(); r->used = 0;
        if (w == FAN_WAIT_STREAM_ERROR) { r->broken = true; return HFAIL("completing a group", (hipError_t)(code - 0x10000)); }
        const int rc = w == FAN_WAIT_COMM_ERROR ? RFAIL("asynchronous error of the communicator", code)
                                                : p264fan_set_error("rccl transport: a group did not complete within %.0f s (a peer has gone?): communicator aborted", r->timeout_s);
        rc_abort(r);
        return rc;
    }
    hipError_t e = hipSuccess;
    for (auto &p : r->pending) if (e == hipSuccess) e = hipMemcpyAsync(p.host, p.dev, p.bytes, hipMemcpyDeviceToHost, r->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(r->stream);    // (copies only: nothing here waits for a peer)
    r->pending.clear(); r->used = 0;
    if (e != hipSuccess) { r->broken = true; return HFAIL("copying what a group received", e); }
    return 0;
}
int rc_send(void *c, int peer, const void *buf, size_t n)
{
    Rccl *r = (Rccl *)c;
    if (!r->comm) return p264fan_set_error("rccl transport: the communicator has been aborted");
    (void)hipSetDevice(r->device);
    void *d = stage_buf(r, n);
    if (!d) return -1;
    hipError_t e = hipMemcpyAsync(d, buf, n, hipMemcpyHostToDevice, r->stream);
    if (e != hipSuccess) return HFAIL("host -> device copy of a send", e);
    if (int ne = g_api.send(d, n, 1 /* ncclUint8 */, peer, r->comm, r->stream)) { r->broken = true; return RFAIL("ncclSend", ne); }
    return r->in_group ? 0 : finish(r);
}
int rc_recv(void *c, int peer, void *buf, size_t n)
{
    Rccl *r = (Rccl *)c;
    if (!r->comm) return p264fan_set_error("rccl transport: the communicator has been aborted");
    (void)hipSetDevice(r->device);
    void *d = stage_buf(r, n);
    if (!d) return -1;
    if (int ne = g_api.recv(d, n, 1, peer, r->comm, r->stream)) { r->broken = true; return RFAIL("ncclRecv", ne); }
    r->pending.push_back({ buf, d, n });
    return r->in_group ? 0 : finish(r);
}
// device buffers: no staging on this side (the caller's buffer is complete / free on the device when it calls: ordering
// against the streams that produce or consume it is the caller's - the fan-out waits for its reconstruction stream first)
int rc_send_dev(void *c, int peer, const void *dev, size_t n)
{
    Rccl *r = (Rccl *)c;
    if (!r->comm) return p264fan_set_error("rccl transport: the communicator has been aborted");
    (void)hipSetDevice(r->device);
    if (int ne = g_api.send((void *)dev, n, 1 /* ncclUint8 */, peer, r->comm, r->stream)) { r->broken = true; return RFAIL("ncclSend", ne); }
    return r->in_group ? 0 : finish(r);
}
int rc_recv_dev(void *c, int peer, void *dev, size_t n)
{
    Rccl *r = (Rccl *)c;
    if (!r->comm) return p264fan_set_error("rccl transport: the communicator has been aborted");
    (void)hipSetDevice(r->device);
    if (int ne = g_api.recv(dev, n, 1, peer, r->comm, r->stream)) { r->broken = true; return RFAIL("ncclRecv", ne); }
    return r->in_group ? 0 : finish(r);
}
int rc_begin(void *c) { Rccl *r = (Rccl *)c; r->in_group = true; if (int ne = g_api.gstart()) { r->broken = true; return RFAIL("ncclGroupStart", ne); } return 0; }
int rc_end(void *c) { Rccl *r = (Rccl *)c; r->in_group = false; if (int ne = g_api.gend()) { r->broken = true; r->pending.clear(); r->used = 0; return RFAIL("ncclGroupEnd", ne); } return finish(r); }
void rc_close(void *c)
{
    Rccl *r = (Rccl *)c;
    if (!r) return;
    (void)hipSetDevice(r->device);
    // a communicator that has seen an error is aborted, not destroyed: ncclCommDestroy waits for outstanding operations
    if (r->comm) { if (r->broken && g_api.abort) g_api.abort(r->comm); else g_api.destroy(r->comm); }
    for (void *p : r->stage) if (p) (void)hipFree(p);
    if (r->stream) (void)hipStreamDestroy(r->stream);
    delete r;
}
}  // namespace

extern "C" int p264fan_rccl_unique_id(uint8_t id[128])
{
    if (!id) return p264fan_set_error("p264fan_rccl_unique_id: null argument");
    if (!load_api()) return -1;
    Uid u;
    if (int ne = g_api.uid(&u)) return RFAIL("ncclGetUniqueId", ne);
    memcpy(id, u.internal, 128);
    return 0;
}

extern "C" int p264fan_rccl_transport(p264fan_transport_t *t, int rank, int world, const uint8_t id[128], int device)
{
    if (!t || !id || world < 1 || rank < 0 || rank >= world) return p264fan_set_error("p264fan_rccl_transport: bad argument");
    if (!load_api()) return -1;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return p264fan_set_error("rccl transport: no HIP device %d (have %d)", device, ndev);
    Rccl *r = new Rccl();
    r->device = device;
    Uid u; memcpy(u.internal, id, 128);
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { HFAIL("stream on the rank's device", e); rc_close(r); return -1; }
    if (int ne = g_api.init(&r->comm, world, u, rank)) {
        p264fan_set_error("rccl transport: ncclCommInitRank(rank %d of %d, device %d): %s (%d)", rank, world, device, nccl_str(ne), ne);
        r->comm = nullptr;
        rc_close(r);
        return -1;
    }
    t->ctx = r; t->send = rc_send; t->recv = rc_recv; t->group_begin = rc_begin; t->group_end = rc_end; t->close = rc_close; t->name = "rccl"; t->abort = rc_abort; t->send_dev = rc_send_dev; t->recv_dev = rc_recv_dev;
    if (const char *env = getenv("P264AMD_FAN_TIMEOUT_S")) { const double v = atof(env); if (v > 0) r->timeout_s = v; }
    return 0;
}
