/* fanout.c - stream fan-out over a point-to-point transport (include/p264fan.h): protocol, root and worker loops, the
 * TCP transport and the default (MI355X) backend.  The RCCL transport lives next to the HIP code (csrc/hip/fan_rccl.hip). */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <errno.h>
#include <time.h>
#include <unistd.h>
#include <pthread.h>
#include <sys/socket.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <arpa/inet.h>
#include "p264fan.h"
#include "p264parse.h"
#include "p264_dropin.h"
#include "host_cpu.h"

static __thread char g_err[512] = "";
static int fail(const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
    return -1;
}
const char *p264fan_last_error(void) { return g_err; }
/* (for the RCCL transport in csrc/hip/fan_rccl.hip: same message slot) */
int p264fan_set_error(const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
    return -1;
}
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

/* ---------------------------------------------------------------- messages -------------- */
/* Per round and worker, always in this order and always of these sizes (a fixed-size exchange keeps both sides in step
 * whatever fails):
 *   root -> worker   fan_ctrl_t                      how many pictures follow and their sizes, or FAN_FINISHED
 *   root -> worker   n fan_head_t in one message     the pictures' descriptors (host memory: the worker lays its input slots out
 *                                                    from them)
 *   root -> worker   n packed pictures               each ONE block in the layout of an input slot (p264hip_input_layout_t); a worker
 *                                                    whose transport and backend have the device road receives it straight into the
 *                                                    slot (recv_dev), otherwise into host memory
 *   worker -> root   fan_status_t                    0, or what failed on the worker (it keeps serving: the root ends the job
 *                                                    with FAN_FINISHED at the next round boundary)
 *   worker -> root   n frames of MB-aligned I420     only when the status is 0
 * The root sends FAN_FINISHED only where a worker expects a control block.  Everything on the root that can fail
 * without the transport failing (parse, allocation, packing) happens BEFORE the round's control block goes out, and a
 * failure of the root's own reconstruction is reported only after the round's gather: the root never leaves a worker
 * in the middle of a round.  If the transport itself fails there is nothing left to say - both transports' calls
 * return an error then (TCP: peer closed). */
#define FAN_MAX_PER_ROUND 64            /* pictures one worker takes per round */
#define FAN_FINISHED (-1)
typedef struct {                        /* root -> worker, once per round, fixed size */
    int32_t n;                          /* pictures that follow, or FAN_FINISHED */
    int32_t mb_w, mb_h, slots, n_local_streams;
    uint32_t bytes[FAN_MAX_PER_ROUND];  /* size of each packed picture */
} fan_ctrl_t;
typedef struct {                        /* worker -> root, once per round, fixed size */
    int32_t rc, n;
    int32_t device_road;                /* this round's pictures and planes never touched the worker's host memory */
    char msg[244];
} fan_status_t;
typedef struct {                        /* descriptor of a packed picture (travels apart from the arrays) */
    uint32_t magic;
    int32_t  local_stream;
    p264hip_picture_t desc;             /* pointers are meaningless on the wire */
    uint32_t n_mb;
} fan_head_t;
#define FAN_MAGIC 0x70464e33u
/* the arrays of a picture as one block in the layout of an input slot; the head beside it */
static size_t packed_size(const p264hip_picture_t *p)
{
    p264hip_input_layout_t L;
    return p264hip_input_layout(p, &L) ? 0 : L.bytes;
}
static int pack_picture(fan_head_t *h, uint8_t *dst, size_t cap, int local_stream, const p264hip_picture_t *p)
{
    memset(h, 0, sizeof *h);
    h->magic = FAN_MAGIC; h->local_stream = local_stream; h->desc = *p; h->n_mb = (uint32_t)((size_t)p->mb_w * p->mb_h);
    h->desc.mb = NULL; h->desc.mv = NULL; h->desc.ref_idx = NULL; h->desc.i4modes = NULL; h->desc.coefs = NULL; h->desc.mv_l1 = NULL; h->desc.ref_idx_l1 = NULL;
    return p264hip_pack_input(p, dst, cap) < 0 ? fail("a parsed picture does not pack (inconsistent macroblock records)") : 0;
}
static int check_head(const fan_head_t *h, size_t bytes)
{
    if (h->magic != FAN_MAGIC || h->desc.mb_w < 1 || h->desc.mb_h < 1 || h->n_mb != (uint32_t)(h->desc.mb_w * h->desc.mb_h)) return fail("packed picture: bad header");
    if (packed_size(&h->desc) != bytes) return fail("packed picture: %zu bytes, its header says %zu", bytes, packed_size(&h->desc));
    return 0;
}
/* the picture described by a head and its block, the arrays pointing into the block */
static int unpack_picture(const fan_head_t *h, const uint8_t *body, size_t bytes, p264hip_picture_t *out, int *local_stream)
{
    if (check_head(h, bytes)) return -1;
    *local_stream = h->local_stream;
    return p264hip_unpack_input(&h->desc, body, bytes, out) ? fail("packed picture: bad layout") : 0;
}

/* ---------------------------------------------------------------- default backend ------- */
/* The MI355X path of this library.  reconstruct() only uploads (asynchronously) and NOTES the picture; sync() reconstructs
 * everything noted since the last sync() as ONE batch - the pictures of a round belong to different local streams, and the
 * kernels are built for batches: a round of 64 pictures as 64 single-picture batches would run the GPU nearly empty, one
 * launch latency after the other - then converts and downloads the frames and waits once.  The picture's arrays and the
 * output buffers must stay untouched until sync() (the fan-out keeps a round's messages and frames alive until then; frames
 * live in pinned memory so that the downloads are real DMA transfers). */
typedef struct { p264hip_ctx *hip; int mb_w, mb_h, n_local, n_pend, n_last; int *stream, *slot; uint8_t **out; void **planes; size_t plane_bytes; } hipbk_t;
static void hipbk_close(void *ctx);
static int hipbk_open(void **ctx, int device, int mb_w, int mb_h, int n_local, int slots)
{
    hipbk_t *b = (hipbk_t *)calloc(1, sizeof *b);
    if (!b) return fail("out of memory");
    b->stream = (int *)malloc(sizeof(int) * (size_t)n_local); b->slot = (int *)malloc(sizeof(int) * (size_t)n_local);
    b->out = (uint8_t **)malloc(sizeof(uint8_t *) * (size_t)n_local); b->planes = (void **)calloc((size_t)n_local, sizeof(void *));
    if (!b->stream || !b->slot || !b->out || !b->planes) { hipbk_close(b); return fail("out of memory"); }
    if (p264hip_create(&b->hip, device, mb_w, mb_h, n_local, slots, n_local) != P264HIP_OK) { fail("%s", p264hip_last_error()); b->hip = NULL; hipbk_close(b); return -1; }
    b->mb_w = mb_w; b->mb_h = mb_h; b->n_local = n_local;
    *ctx = b;
    return 0;
}
static int hipbk_sync(void *ctx)
{
    hipbk_t *b = (hipbk_t *)ctx;
    const int w = b->mb_w * 16, h = b->mb_h * 16, n = b->n_pend;
    b->n_pend = 0; b->n_last = 0;
    if (n) {
        /* input slot = local stream (hipbk_reconstruct); one batch, then the frames: downloaded, or - pictures that came the
         * device road - converted to planes that stay on the device (hipbk_planes) */
        if (p264hip_reconstruct(b->hip, b->stream, b->stream, n) != P264HIP_OK) return fail("%s", p264hip_last_error());
        for (int i = 0; i < n; i++) {
            uint8_t *o = b->out[i];
            b->planes[i] = NULL;
            if (o) { if (p264hip_read_frame_async(b->hip, b->stream[i], b->slot[i], o, w, o + (size_t)w * h, o + (size_t)w * h * 5 / 4, w / 2) != P264HIP_OK) return fail("%s", p264hip_last_error()); }
            else if (p264hip_frame_planar_device(b->hip, b->stream[i], b->slot[i], i, &b->planes[i], &b->plane_bytes) != P264HIP_OK) return fail("%s", p264hip_last_error());
        }
        b->n_last = n;
    }
    return p264hip_sync(b->hip) == P264HIP_OK ? 0 : fail("%s", p264hip_last_error());
}
static int hipbk_reconstruct(void *ctx, int s, const p264hip_picture_t *pic, uint8_t *i420)
{
    hipbk_t *b = (hipbk_t *)ctx;
    if (s < 0 || s >= b->n_local) return fail("local stream %d out of range", s);
    for (int i = 0; i < b->n_pend; i++)
        if (b->stream[i] == s) { if (hipbk_sync(b)) return -1; break; }     /* (a second picture of a stream: the first one has to be through) */
    if (p264hip_upload_async(b->hip, s, pic) != P264HIP_OK) return fail("%s", p264hip_last_error());
    b->stream[b->n_pend] = s; b->slot[b->n_pend] = pic->dst_slot; b->out[b->n_pend] = i420;
    b->n_pend++;
    return 0;
}
/* the device road: the picture's arrays are written into the stream's input slot by the transport */
static int hipbk_reserve(void *ctx, int s, const p264hip_picture_t *desc, void **dev, size_t *bytes)
{
    hipbk_t *b = (hipbk_t *)ctx;
    if (s < 0 || s >= b->n_local) return fail("local stream %d out of range", s);
    for (int i = 0; i < b->n_pend; i++)
        if (b->stream[i] == s) return fail("two pictures of local stream %d in one round", s);     /* (its slot is still to be read) */
    if (p264hip_input_reserve(b->hip, s, desc, dev, bytes) != P264HIP_OK) return fail("%s", p264hip_last_error());
    return 0;
}
static int hipbk_reconstruct_reserved(void *ctx, int s, const p264hip_picture_t *desc)
{
    hipbk_t *b = (hipbk_t *)ctx;
    if (s < 0 || s >= b->n_local || b->n_pend >= b->n_local) return fail("local stream %d out of range", s);
    if (p264hip_input_commit(b->hip, s) != P264HIP_OK) return fail("%s", p264hip_last_error());
    b->stream[b->n_pend] = s; b->slot[b->n_pend] = desc->dst_slot; b->out[b->n_pend] = NULL;
    b->n_pend++;
    return 0;
}
static int hipbk_planes(void *ctx, int k, void **dev, size_t *bytes)
{
    hipbk_t *b = (hipbk_t *)ctx;
    if (k < 0 || k >= b->n_last || !b->planes[k]) return fail("no device planes for picture %d of the round", k);
    *dev = b->planes[k]; *bytes = b->plane_bytes;
    return 0;
}
static void hipbk_close(void *ctx)
{
    hipbk_t *b = (hipbk_t *)ctx;
    if (!b) return;
    if (b->hip) p264hip_destroy(b->hip);
    free(b->stream); free(b->slot); free(b->out); free(b->planes); free(b);
}
static const p264fan_backend_t g_hip_backend = { NULL, hipbk_open, hipbk_reconstruct, hipbk_close, hipbk_sync, hipbk_reserve, hipbk_reconstruct_reserved, hipbk_planes };
/* frames: pinned when a HIP device is there (downloads and RCCL staging copies become DMA), plain memory otherwise */
static uint8_t *frames_alloc(size_t bytes, int *pinned)
{
    uint8_t *p = (uint8_t *)p264hip_host_alloc(bytes);
    *pinned = p != NULL;
    return p ? p : (uint8_t *)malloc(bytes);
}
static void frames_free(uint8_t *p, int pinned) { if (pinned) p264hip_host_free(p); else free(p); }

/* ---------------------------------------------------------------- TCP transport --------- */
typedef struct { int rank, world; int *fd; uint8_t *bounce; size_t bounce_cap; } tcp_t;        /* fd[peer]; root: one per worker, worker: fd[0] */
static int io_all(int fd, void *buf, size_t n, int wr)
{
    uint8_t *p = (uint8_t *)buf;
    while (n) {
        ssize_t k = wr ? send(fd, p, n, MSG_NOSIGNAL) : recv(fd, p, n, 0);
        if (k < 0 && errno == EINTR) continue;
        if (k <= 0) return fail("tcp %s: %s", wr ? "send" : "recv", k == 0 ? "peer closed" : strerror(errno));
        p += k; n -= (size_t)k;
    }
    return 0;
}
static int tcp_send(void *c, int peer, const void *buf, size_t n) { tcp_t *t = (tcp_t *)c; return io_all(t->fd[peer], (void *)buf, n, 1); }
static int tcp_recv(void *c, int peer, void *buf, size_t n) { tcp_t *t = (tcp_t *)c; return io_all(t->fd[peer], buf, n, 0); }
static int tcp_nop(void *c) { (void)c; return 0; }
/* P264AMD_FAN_TCP_DEVICE=1: device buffers through a host bounce buffer - stands in for a device-to-device transport where
 * ranks share one GPU (tests of the device road; RCCL needs one GPU per rank) */
static uint8_t *tcp_bounce(tcp_t *t, size_t n)
{
    if (n > t->bounce_cap) { free(t->bounce); t->bounce = (uint8_t *)malloc(n + n / 4); t->bounce_cap = t->bounce ? n + n / 4 : 0; }
    if (!t->bounce) fail("out of memory");
    return t->bounce;
}
static int tcp_send_dev(void *c, int peer, const void *dev, size_t n)
{
    tcp_t *t = (tcp_t *)c;
    uint8_t *b = tcp_bounce(t, n);
    if (!b) return -1;
    if (p264hip_copy_from_device(b, dev, n) != P264HIP_OK) return fail("%s", p264hip_last_error());
    return io_all(t->fd[peer], b, n, 1);
}
static int tcp_recv_dev(void *c, int peer, void *dev, size_t n)
{
    tcp_t *t = (tcp_t *)c;
    uint8_t *b = tcp_bounce(t, n);
    if (!b) return -1;
    if (io_all(t->fd[peer], b, n, 0)) return -1;
    return p264hip_copy_to_device(dev, b, n) != P264HIP_OK ? fail("%s", p264hip_last_error()) : 0;
}
static void tcp_abort(void *c)
{
    tcp_t *t = (tcp_t *)c;
    if (!t) return;
    for (int i = 0; i < t->world; i++) if (t->fd[i] >= 0) shutdown(t->fd[i], SHUT_RDWR);
}
static void tcp_close(void *c)
{
    tcp_t *t = (tcp_t *)c;
    if (!t) return;
    for (int i = 0; i < t->world; i++) if (t->fd[i] >= 0) close(t->fd[i]);
    free(t->fd); free(t->bounce); free(t);
}
int p264fan_tcp_transport(p264fan_transport_t *out, int rank, int world, const char *host, int port)
{
    if (p264amd_cpu_refuse("p264fan_tcp_transport")) return p264fan_set_error("p264fan_tcp_transport: CPU older than the build's target (x86-64-v3)");
    if (!out || world < 1 || rank < 0 || rank >= world || port < 1 || port > 65535) return fail("p264fan_tcp_transport: bad argument");
    tcp_t *t = (tcp_t *)calloc(1, sizeof *t);
    if (!t) return fail("out of memory");
    t->rank = rank; t->world = world; t->fd = (int *)malloc(sizeof(int) * (size_t)world);
    if (!t->fd) { free(t); return fail("out of memory"); }
    for (int i = 0; i < world; i++) t->fd[i] = -1;
    struct sockaddr_in a; memset(&a, 0, sizeof a);
    a.sin_family = AF_INET; a.sin_port = htons((uint16_t)port);
    const int one = 1;
    if (rank == 0) {
        a.sin_addr.s_addr = htonl(INADDR_ANY);
        int ls = socket(AF_INET, SOCK_STREAM, 0);
        if (ls < 0 || setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one) || bind(ls, (struct sockaddr *)&a, sizeof a) || listen(ls, world)) {
            if (ls >= 0) close(ls);
            tcp_close(t); return fail("tcp root: cannot listen on port %d: %s", port, strerror(errno));
        }
        for (int k = 1; k < world; k++) {                     /* every worker introduces itself with its rank */
            int fd = accept(ls, NULL, NULL);
            int32_t r = -1;
            if (fd < 0 || io_all(fd, &r, sizeof r, 0) || r < 1 || r >= world || t->fd[r] >= 0) { if (fd >= 0) close(fd); close(ls); tcp_close(t); return fail("tcp root: bad worker connection"); }
            setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
            t->fd[r] = fd;
        }
        close(ls);
    } else {
        if (inet_pton(AF_INET, host ? host : "127.0.0.1", &a.sin_addr) != 1) { tcp_close(t); return fail("tcp worker: bad root address %s", host); }
        int fd = -1;
        for (int tries = 0; tries < 600; tries++) {            /* the root may not be listening yet */
            fd = socket(AF_INET, SOCK_STREAM, 0);
            if (fd >= 0 && connect(fd, (struct sockaddr *)&a, sizeof a) == 0) break;
            if (fd >= 0) close(fd);
            fd = -1;
            usleep(100000);
        }
        int32_t r = rank;
        if (fd < 0 || io_all(fd, &r, sizeof r, 1)) { if (fd >= 0) close(fd); tcp_close(t); return fail("tcp worker %d: cannot reach the root at %s:%d", rank, host ? host : "127.0.0.1", port); }
        setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
        t->fd[0] = fd;
    }
    out->ctx = t; out->send = tcp_send; out->recv = tcp_recv; out->group_begin = tcp_nop; out->group_end = tcp_nop; out->close = tcp_close; out->name = "tcp"; out->abort = tcp_abort;
    out->send_dev = NULL; out->recv_dev = NULL;
    { const char *e = getenv("P264AMD_FAN_TCP_DEVICE"); if (e && atoi(e) > 0) { out->send_dev = tcp_send_dev; out->recv_dev = tcp_recv_dev; } }
    return 0;
}

/* ---------------------------------------------------------------- fan-out --------------- */
struct p264fan {
    int rank, world, device;
    p264fan_transport_t t;
    p264fan_backend_t bk; void *bk_ctx;
};

p264fan *p264fan_open(int rank, int world, const p264fan_transport_t *t, const p264fan_backend_t *backend, int device)
{
    if (p264amd_cpu_refuse("p264fan_open")) { p264fan_set_error("p264fan_open: CPU older than the build's target (x86-64-v3)"); return NULL; }
    if (world < 1 || rank < 0 || rank >= world || (world > 1 && (!t || !t->send || !t->recv))) { fail("p264fan_open: bad argument"); return NULL; }
    p264fan *f = (p264fan *)calloc(1, sizeof *f);
    if (!f) { fail("out of memory"); return NULL; }
    f->rank = rank; f->world = world; f->device = device;
    if (t) f->t = *t;
    f->bk = backend ? *backend : g_hip_backend;
    return f;
}
void p264fan_close(p264fan *f)
{
    if (!f) return;
    if (f->bk_ctx && f->bk.close) f->bk.close(f->bk_ctx);
    if (f->t.close) f->t.close(f->t.ctx);
    free(f);
}
static int gb(p264fan *f) { return f->t.group_begin ? f->t.group_begin(f->t.ctx) : 0; }
static int ge(p264fan *f) { return f->t.group_end ? f->t.group_end(f->t.ctx) : 0; }
static int bk_sync(p264fan *f) { return (f->bk.sync && f->bk_ctx) ? f->bk.sync(f->bk_ctx) : 0; }

int p264fan_worker_run(p264fan *f)
{
    if (!f || f->rank == 0) return fail("p264fan_worker_run: not a worker");
    uint8_t *msg[FAN_MAX_PER_ROUND] = { 0 }; size_t cap[FAN_MAX_PER_ROUND] = { 0 };
    fan_head_t *heads = (fan_head_t *)malloc(sizeof(fan_head_t) * FAN_MAX_PER_ROUND);
    uint8_t *out = NULL; size_t frame = 0; int out_pinned = 0;
    int rc = heads ? 0 : fail("out of memory"), fatal = heads ? 0 : 1;  /* rc: transport failures only, they end the loop; fatal: this rank cannot stay in step */
    char first_err[248] = "";
    /* the device road: pictures straight into their input slots, planes straight out of the conversion buffers */
    const int dev_road = f->t.recv_dev && f->t.send_dev && f->bk.reserve && f->bk.reconstruct_reserved && f->bk.planes;
    while (!rc) {
        fan_ctrl_t c;
        if (gb(f) || f->t.recv(f->t.ctx, 0, &c, sizeof c) || ge(f)) { rc = -1; break; }
        if (c.n == FAN_FINISHED) break;
        if (c.n < 0 || c.n > FAN_MAX_PER_ROUND) { rc = fail("worker %d: bad control block", f->rank); fatal = 1; break; }   /* (out of step: nothing sane left to do) */
        fan_status_t st; memset(&st, 0, sizeof st); st.n = c.n;
        /* ---- the round's descriptors */
        if (c.n && (gb(f) || f->t.recv(f->t.ctx, 0, heads, sizeof(fan_head_t) * (size_t)c.n) || ge(f))) { rc = -1; break; }
        for (int k = 0; k < c.n && !st.rc; k++) if (check_head(&heads[k], c.bytes[k])) st.rc = -1;
        /* ---- the backend, once */
        if (!st.rc && !f->bk_ctx && !first_err[0]) {
            if (f->bk.open(&f->bk_ctx, f->device, c.mb_w, c.mb_h, c.n_local_streams, c.slots)) { st.rc = -1; f->bk_ctx = NULL; }
            else {
                frame = (size_t)c.mb_w * c.mb_h * 384;
                if (!dev_road) { out = frames_alloc(frame * FAN_MAX_PER_ROUND, &out_pinned); if (!out) { st.rc = -1; fail("worker %d: out of memory", f->rank); } }
            }
        }
        if (first_err[0]) { st.rc = -1; fail("%s", first_err); }      /* an earlier round failed: this worker's frame stores are stale */
        /* ---- the round's pictures: always received, whatever state this worker is in.  Device road: every picture's slot is
         *      reserved first; if that fails for one of them the whole round goes to host buffers and is answered with the error */
        void *slot_dev[FAN_MAX_PER_ROUND]; int on_device = dev_road && !st.rc;
        for (int k = 0; k < c.n && on_device; k++) {
            size_t bytes = 0;
            if (f->bk.reserve(f->bk_ctx, heads[k].local_stream, &heads[k].desc, &slot_dev[k], &bytes) || bytes != c.bytes[k]) { if (bytes && bytes != c.bytes[k]) fail("worker %d: slot of %zu bytes for a picture of %u", f->rank, bytes, c.bytes[k]); st.rc = -1; on_device = 0; }
        }
        if (!on_device)
            for (int k = 0; k < c.n; k++)
                if (c.bytes[k] > cap[k]) {
                    free(msg[k]); cap[k] = 0;
                    msg[k] = (uint8_t *)malloc((size_t)c.bytes[k] + c.bytes[k] / 4);
                    if (!msg[k]) { rc = fail("worker %d: out of memory", f->rank); fatal = 1; break; }       /* (cannot even receive: the transport is aborted below) */
                    cap[k] = (size_t)c.bytes[k] + c.bytes[k] / 4;
                }
        if (rc) break;
        char keep_err[sizeof g_err]; memcpy(keep_err, g_err, sizeof keep_err);
        if (gb(f)) { rc = -1; break; }
        for (int k = 0; k < c.n && !rc; k++)
            if (on_device ? f->t.recv_dev(f->t.ctx, 0, slot_dev[k], c.bytes[k]) : f->t.recv(f->t.ctx, 0, msg[k], c.bytes[k])) rc = -1;
        if (ge(f) || rc) { rc = -1; break; }
        if (st.rc) memcpy(g_err, keep_err, sizeof keep_err);
        /* ---- reconstruct; a failure becomes the round's status and the worker keeps serving */
        for (int k = 0; k < c.n && !st.rc; k++) {
            if (on_device) { if (f->bk.reconstruct_reserved(f->bk_ctx, heads[k].local_stream, &heads[k].desc)) st.rc = -1; continue; }
            p264hip_picture_t pic; int ls = 0;
            if (unpack_picture(&heads[k], msg[k], c.bytes[k], &pic, &ls) || f->bk.reconstruct(f->bk_ctx, ls, &pic, out + frame * (size_t)k)) st.rc = -1;
        }
        if (!st.rc && bk_sync(f)) st.rc = -1;
        void *plane_dev[FAN_MAX_PER_ROUND];
        for (int k = 0; k < c.n && !st.rc && on_device; k++) {
            size_t bytes = 0;
            if (f->bk.planes(f->bk_ctx, k, &plane_dev[k], &bytes) || bytes != frame) { if (bytes && bytes != frame) fail("worker %d: planes of %zu bytes, a frame has %zu", f->rank, bytes, frame); st.rc = -1; }
        }
        if (st.rc) { snprintf(st.msg, sizeof st.msg, "%.236s", g_err[0] ? g_err : "reconstruction failed"); if (!first_err[0]) snprintf(first_err, sizeof first_err, "%.236s", st.msg); }
        /* ---- status, then the frames */
        st.device_road = on_device && !st.rc && c.n > 0;
        if (gb(f) || f->t.send(f->t.ctx, 0, &st, sizeof st) || ge(f)) { rc = -1; break; }
        if (st.rc) continue;
        if (gb(f)) { rc = -1; break; }
        for (int k = 0; k < c.n && !rc; k++)
            if (on_device ? f->t.send_dev(f->t.ctx, 0, plane_dev[k], frame) : f->t.send(f->t.ctx, 0, out + frame * (size_t)k, frame)) rc = -1;
        if (ge(f) || rc) { rc = -1; break; }
    }
    /* leaving out of step: the root must not wait for this rank's status or frames (RCCL has no "peer closed") */
    if (fatal && f->t.abort) { char keep[sizeof g_err]; memcpy(keep, g_err, sizeof keep); f->t.abort(f->t.ctx); memcpy(g_err, keep, sizeof keep); }
    for (int k = 0; k < FAN_MAX_PER_ROUND; k++) free(msg[k]);
    free(heads);
    if (out) frames_free(out, out_pinned);
    if (!rc && first_err[0]) rc = fail("%s", first_err);    /* the job failed on this worker, even though it left in step */
    return rc;
}

typedef struct { p264parse *parser; const uint8_t *in; int64_t size, pos; uint8_t *rbsp; int64_t rbsp_cap; int done; int64_t pictures; } fstream_t;
/* next picture of a stream, or NULL at its end */
static const p264hip_picture_t *next_picture(fstream_t *s, int max_pictures, int *failed)
{
    if (s->done || (max_pictures > 0 && s->pictures >= max_pictures)) { s->done = 1; return NULL; }
    int64_t off, len;
    while (p264_annexb_next(s->in, s->size, &s->pos, &off, &len)) {
        if (len < 1) continue;
        if (len + 8 > s->rbsp_cap) { free(s->rbsp); s->rbsp_cap = len * 2 + 64; s->rbsp = (uint8_t *)malloc((size_t)s->rbsp_cap); if (!s->rbsp) { *failed = 1; return NULL; } }
        p264_nal_t nal; nal.p_payload = s->rbsp;
        p264_nal_decode(&nal, (void *)(s->in + off), (int)len);
        const p264hip_picture_t *pic = NULL;
        int rc = p264parse_nal(s->parser, nal.i_type, nal.i_ref_idc, nal.p_payload, nal.i_payload, &pic);
        if (rc < 0) { *failed = 1; return NULL; }
        if (rc == 1) { s->pictures++; return pic; }
    }
    s->done = 1;
    return NULL;
}

/* One round's worth of work, produced by the parse side and consumed by the exchange side: every stream's next picture,
 * PACKED (the parser's arrays only live until the stream's next call; a packed copy lets the next round be parsed while
 * this one travels and is reconstructed - for the root's own streams too). */
typedef struct {
    uint8_t **msg; size_t *cap, *len;   /* [n_streams]; len 0 = the stream has no picture in this round */
    fan_head_t *head;                   /* [n_streams] the descriptors of the packed pictures */
    int64_t *index;                     /* picture number inside its stream */
    int n, failed, slots, mb_w, mb_h;
    char err[200];
} fan_round_t;
typedef struct {
    fstream_t *st; int n_streams, world, max_pictures, threads;
    fan_round_t rounds[2];
    int ready[2];                       /* 0 free, 1 filled */
    int stop;
    pthread_mutex_t mu; pthread_cond_t cv;
    double parse_seconds;
} fan_producer_t;

typedef struct { fan_producer_t *P; fan_round_t *R; int first, step, failed; } parse_job_t;
static void *parse_worker(void *arg)
{
    parse_job_t *j = (parse_job_t *)arg;
    fan_producer_t *P = j->P; fan_round_t *R = j->R;
    for (int s = j->first; s < P->n_streams; s += j->step) {
        R->len[s] = 0;
        const p264hip_picture_t *pic = next_picture(&P->st[s], P->max_pictures, &j->failed);
        if (!pic) continue;
        const size_t need = packed_size(pic);
        if (need > R->cap[s]) {
            free(R->msg[s]); R->cap[s] = 0;
            R->msg[s] = (uint8_t *)malloc(need + need / 4);
            if (!R->msg[s]) { j->failed = 1; continue; }
            R->cap[s] = need + need / 4;
        }
        if (!need || pack_picture(&R->head[s], R->msg[s], R->cap[s], s / P->world, pic)) { j->failed = 1; continue; }
        R->len[s] = need;
        R->index[s] = P->st[s].pictures - 1;
    }
    return NULL;
}
/* the round's pictures are parsed side by side on `threads` host threads: the streams are independent, and the serial
 * parse of one stream is what bounds the fan-out */
static void fill_round(fan_producer_t *P, fan_round_t *R)
{
    int threads = P->threads;
    if (threads > P->n_streams) threads = P->n_streams;
    if (threads > 64) threads = 64;
    parse_job_t jobs[64];
    pthread_t tid[64];
    int started = 0;
    for (int t = 0; t < threads; t++) jobs[t] = (parse_job_t){ P, R, t, threads, 0 };
    for (int t = 1; t < threads; t++) { if (pthread_create(&tid[t], NULL, parse_worker, &jobs[t])) break; started = t; }
    for (int t = started + 1; t < threads; t++) parse_worker(&jobs[t]);     /* (threads that could not be started: done here) */
    parse_worker(&jobs[0]);
    for (int t = 1; t <= started; t++) pthread_join(tid[t], NULL);
    R->n = 0; R->failed = 0; R->slots = 0; R->mb_w = R->mb_h = 0; R->err[0] = 0;
    for (int t = 0; t < threads; t++) if (jobs[t].failed) { R->failed = 1; snprintf(R->err, sizeof R->err, "a stream failed to parse (or the host ran out of memory)"); }
    for (int s = 0; s < P->n_streams && !R->failed; s++) {
        if (!R->len[s]) continue;
        const fan_head_t h = R->head[s];
        if (!R->n) { R->mb_w = h.desc.mb_w; R->mb_h = h.desc.mb_h; }
        else if (h.desc.mb_w != R->mb_w || h.desc.mb_h != R->mb_h) { R->failed = 1; snprintf(R->err, sizeof R->err, "stream %d has a different picture size (%dx%d macroblocks, the job runs at %dx%d)", s, h.desc.mb_w, h.desc.mb_h, R->mb_w, R->mb_h); }
        const int sl = p264parse_slots(P->st[s].parser);       /* every rank's frame stores are sized for the stream that needs most */
        if (sl > R->slots) R->slots = sl;
        R->n++;
    }
}
static void *producer_main(void *arg)
{
    fan_producer_t *P = (fan_producer_t *)arg;
    for (int k = 0;; k ^= 1) {
        pthread_mutex_lock(&P->mu);
        while (P->ready[k] && !P->stop) pthread_cond_wait(&P->cv, &P->mu);
        const int stop = P->stop;
        pthread_mutex_unlock(&P->mu);
        if (stop) break;
        const double t0 = now_s();
        fill_round(P, &P->rounds[k]);
        const int last = P->rounds[k].n == 0 || P->rounds[k].failed;
        pthread_mutex_lock(&P->mu);
        P->parse_seconds += now_s() - t0;
        P->ready[k] = 1;
        pthread_cond_broadcast(&P->cv);
        pthread_mutex_unlock(&P->mu);
        if (last) break;
    }
    return NULL;
}

int p264fan_root_run(p264fan *f, int n_streams, const uint8_t *const *annexb, const int64_t *sizes, int max_pictures,
                     p264fan_frame_cb on_frame, void *user, p264fan_stats_t *stats)
{
    if (!f || f->rank != 0 || n_streams < 1 || !annexb || !sizes) return fail("p264fan_root_run: bad argument");
    const int W = f->world;
    if ((n_streams + W - 1) / W > FAN_MAX_PER_ROUND) return fail("p264fan_root_run: more than %d streams per rank", FAN_MAX_PER_ROUND);
    fan_producer_t P; memset(&P, 0, sizeof P);
    P.st = (fstream_t *)calloc((size_t)n_streams, sizeof *P.st);
    fan_ctrl_t *ctrl = (fan_ctrl_t *)calloc((size_t)W, sizeof *ctrl);
    fan_status_t *status = (fan_status_t *)calloc((size_t)W, sizeof *status);
    fan_head_t *heads = (fan_head_t *)malloc(sizeof(fan_head_t) * (size_t)n_streams);     /* a round's descriptors, grouped by worker */
    int *head_at = (int *)calloc((size_t)W + 1, sizeof(int));
    uint8_t *frames = NULL; size_t frame = 0; int frames_pinned = 0;
    int rc = (P.st && ctrl && status && heads && head_at) ? 0 : fail("out of memory");
    for (int k = 0; k < 2 && !rc; k++) {
        fan_round_t *R = &P.rounds[k];
        R->msg = (uint8_t **)calloc((size_t)n_streams, sizeof *R->msg); R->cap = (size_t *)calloc((size_t)n_streams, sizeof *R->cap);
        R->len = (size_t *)calloc((size_t)n_streams, sizeof *R->len); R->index = (int64_t *)calloc((size_t)n_streams, sizeof *R->index);
        R->head = (fan_head_t *)calloc((size_t)n_streams, sizeof *R->head);
        if (!R->msg || !R->cap || !R->len || !R->index || !R->head) rc = fail("out of memory");
    }
    for (int s = 0; s < n_streams && !rc; s++) {
        P.st[s].parser = p264parse_open(P264PARSE_OPT_QUIET);
        P.st[s].in = annexb[s]; P.st[s].size = sizes[s];
        if (!P.st[s].parser) rc = fail("p264parse_open failed");
    }
    p264fan_stats_t S; memset(&S, 0, sizeof S); S.world = W;
    P.n_streams = n_streams; P.world = W; P.max_pictures = max_pictures;
    P.threads = n_streams;                                   /* one parser thread per stream unless P264AMD_FAN_THREADS says otherwise */
    { const char *e = getenv("P264AMD_FAN_THREADS"); if (e && atoi(e) >= 1) P.threads = atoi(e); }
    S.parse_threads = P.threads < n_streams ? P.threads : n_streams;
    if (S.parse_threads > 64) S.parse_threads = 64;
    pthread_t producer; int have_producer = 0;
    if (!rc) {
        pthread_mutex_init(&P.mu, NULL); pthread_cond_init(&P.cv, NULL);
        if (pthread_create(&producer, NULL, producer_main, &P)) rc = fail("cannot start the parse thread");
        else have_producer = 1;
    }
    const double t0 = now_s();
    int mb_w = 0, mb_h = 0, slots = 0;
    for (int k = 0; !rc; k ^= 1) {
        /* ---- the next round, parsed and packed while the previous one travelled */
        const double w0 = now_s();
        pthread_mutex_lock(&P.mu);
        while (!P.ready[k]) pthread_cond_wait(&P.cv, &P.mu);
        pthread_mutex_unlock(&P.mu);
        S.parse_wait_seconds += now_s() - w0;
        fan_round_t *R = &P.rounds[k];
        if (R->failed) { rc = fail("%s", R->err); break; }
        if (!R->n) break;
        if (!mb_w) {
            mb_w = R->mb_w; mb_h = R->mb_h; slots = R->slots;
            frame = (size_t)mb_w * mb_h * 384;
            frames = frames_alloc(frame * (size_t)n_streams, &frames_pinned);
            if (!frames) { rc = fail("out of memory"); break; }
            if (f->bk.open(&f->bk_ctx, f->device, mb_w, mb_h, (n_streams + W - 1) / W, slots)) { f->bk_ctx = NULL; rc = -1; break; }
        }
        if (R->mb_w != mb_w || R->mb_h != mb_h) { rc = fail("the picture size changed inside the job (%dx%d -> %dx%d macroblocks)", mb_w, mb_h, R->mb_w, R->mb_h); break; }
        if (R->slots > slots) { rc = fail("a stream now needs %d frame slots, the ranks' frame stores were opened with %d (num_ref_frames grew inside the job)", R->slots, slots); break; }
        /* ---- scatter: control blocks, then the packed pictures of the remote streams.  From here to the end of the gather
         *      nothing but the transport may end the round. */
        const double e0 = now_s();
        int rc_local = 0; char err_local[256] = "";
        for (int r = 1; r < W; r++) { memset(&ctrl[r], 0, sizeof ctrl[r]); ctrl[r].mb_w = mb_w; ctrl[r].mb_h = mb_h; ctrl[r].slots = slots; ctrl[r].n_local_streams = (n_streams + W - 1) / W; }
        for (int s = 0; s < n_streams; s++) {
            const int r = s % W;
            if (!R->len[s] || r == 0) continue;
            ctrl[r].bytes[ctrl[r].n++] = (uint32_t)R->len[s];
            S.bytes_scattered += (int64_t)R->len[s]; S.pictures_remote++;
        }
        /* (the descriptors of worker r's pictures, in the order of its control block: heads[head_at[r] .. head_at[r + 1])) */
        head_at[0] = head_at[1] = 0;
        for (int r = 1; r < W; r++) {
            int at = head_at[r];
            for (int s = r; s < n_streams; s += W) if (R->len[s]) heads[at++] = R->head[s];
            head_at[r + 1] = at;
        }
        if (gb(f)) { rc = -1; break; }
        for (int r = 1; r < W && !rc; r++) if (f->t.send(f->t.ctx, r, &ctrl[r], sizeof ctrl[r])) rc = -1;
        if (ge(f) || rc || gb(f)) { rc = -1; break; }
        for (int r = 1; r < W && !rc; r++) if (ctrl[r].n && f->t.send(f->t.ctx, r, heads + head_at[r], sizeof(fan_head_t) * (size_t)ctrl[r].n)) rc = -1;
        if (ge(f) || rc || gb(f)) { rc = -1; break; }
        for (int s = 0; s < n_streams && !rc; s++) if (R->len[s] && s % W) if (f->t.send(f->t.ctx, s % W, R->msg[s], R->len[s])) rc = -1;
        if (ge(f) || rc) { rc = -1; break; }
        S.exchange_seconds += now_s() - e0;
        /* ---- the root's own streams while the workers are busy */
        const double r0 = now_s();
        for (int s = 0; s < n_streams && !rc_local; s += W) {
            if (!R->len[s]) continue;
            p264hip_picture_t pic; int ls = 0;
            if (unpack_picture(&R->head[s], R->msg[s], R->len[s], &pic, &ls) || f->bk.reconstruct(f->bk_ctx, ls, &pic, frames + frame * (size_t)s)) rc_local = -1;
        }
        if (!rc_local && bk_sync(f)) rc_local = -1;
        if (rc_local) snprintf(err_local, sizeof err_local, "root: %.240s", g_err);
        S.reconstruct_seconds += now_s() - r0;
        /* ---- gather: every worker's status, then the frames of those that have them */
        const double g0 = now_s();
        if (gb(f)) { rc = -1; break; }
        for (int r = 1; r < W && !rc; r++) if (f->t.recv(f->t.ctx, r, &status[r], sizeof status[r])) rc = -1;
        if (ge(f) || rc || gb(f)) { rc = -1; break; }
        for (int s = 0; s < n_streams && !rc; s++)
            if (R->len[s] && s % W && status[s % W].rc == 0) { if (f->t.recv(f->t.ctx, s % W, frames + frame * (size_t)s, frame)) rc = -1; S.bytes_gathered += (int64_t)frame; }
        if (ge(f) || rc) { rc = -1; break; }
        S.exchange_seconds += now_s() - g0;
        for (int r = 1; r < W; r++) if (!status[r].rc && status[r].device_road) S.device_road_rounds++;
        for (int r = 1; r < W && !rc; r++)
            if (status[r].rc) { status[r].msg[sizeof status[r].msg - 1] = 0; rc = fail("worker %d: %s", r, status[r].msg); }
        if (!rc && rc_local) rc = fail("%s", err_local);
        if (rc) break;
        for (int s = 0; s < n_streams; s++) if (R->len[s]) { S.pictures++; if (on_frame) on_frame(user, s, R->index[s], mb_w * 16, mb_h * 16, frames + frame * (size_t)s); }
        S.rounds++;
        pthread_mutex_lock(&P.mu); P.ready[k] = 0; pthread_cond_broadcast(&P.cv); pthread_mutex_unlock(&P.mu);
    }
    /* ---- round boundary (or a dead transport): tell the workers to leave */
    char keep[sizeof g_err]; memcpy(keep, g_err, sizeof keep);
    if (ctrl && f->t.send) {
        gb(f);
        for (int r = 1; r < W; r++) { memset(&ctrl[r], 0, sizeof ctrl[r]); ctrl[r].n = FAN_FINISHED; f->t.send(f->t.ctx, r, &ctrl[r], sizeof ctrl[r]); }
        ge(f);
    }
    if (rc) memcpy(g_err, keep, sizeof keep);
    if (have_producer) {
        pthread_mutex_lock(&P.mu); P.stop = 1; P.ready[0] = P.ready[1] = 0; pthread_cond_broadcast(&P.cv); pthread_mutex_unlock(&P.mu);
        pthread_join(producer, NULL);
        pthread_mutex_destroy(&P.mu); pthread_cond_destroy(&P.cv);
    }
    S.seconds = now_s() - t0;
    S.parse_seconds = P.parse_seconds;
    if (stats) *stats = S;
    if (P.st) for (int s = 0; s < n_streams; s++) { if (P.st[s].parser) p264parse_close(P.st[s].parser); free(P.st[s].rbsp); }
    for (int k = 0; k < 2; k++) {
        fan_round_t *R = &P.rounds[k];
        if (R->msg) for (int s = 0; s < n_streams; s++) free(R->msg[s]);
        free(R->msg); free(R->cap); free(R->len); free(R->index); free(R->head);
    }
    if (frames) frames_free(frames, frames_pinned);
    free(P.st); free(ctrl); free(status); free(heads); free(head_at);
    return rc;
}
