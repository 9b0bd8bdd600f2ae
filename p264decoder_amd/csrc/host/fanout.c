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

static __thread char g_err[512] = "";
static int fail(const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
    return -1;
}
const char *p264fan_last_error(void) { return g_err; }
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

/* ---------------------------------------------------------------- messages -------------- */
#define FAN_MAX_PER_ROUND 64            /* pictures one worker takes per round */
#define FAN_FINISHED (-1)
typedef struct {                        /* root -> worker, once per round, fixed size */
    int32_t n;                          /* pictures that follow, or FAN_FINISHED */
    int32_t mb_w, mb_h, slots, n_local_streams;
    uint32_t bytes[FAN_MAX_PER_ROUND];  /* size of each packed picture */
} fan_ctrl_t;
typedef struct {                        /* head of a packed picture; the arrays follow, each padded to 16 bytes */
    uint32_t magic;
    int32_t  local_stream;
    p264hip_picture_t desc;             /* pointers are meaningless on the wire */
    uint32_t n_mb;
} fan_head_t;
#define FAN_MAGIC 0x70464e31u
static size_t pad16(size_t v) { return (v + 15) & ~(size_t)15; }
static size_t packed_size(const p264hip_picture_t *p)
{
    const size_t n = (size_t)p->mb_w * p->mb_h;
    return pad16(sizeof(fan_head_t)) + pad16(n * sizeof(p264hip_mb_t)) + pad16(n * 64) + pad16(n * 4) + pad16(n * 16) + pad16((size_t)p->n_coef_blocks * 32);
}
static void pack_picture(uint8_t *dst, int local_stream, const p264hip_picture_t *p)
{
    const size_t n = (size_t)p->mb_w * p->mb_h;
    fan_head_t h; memset(&h, 0, sizeof h);
    h.magic = FAN_MAGIC; h.local_stream = local_stream; h.desc = *p; h.n_mb = (uint32_t)n;
    h.desc.mb = NULL; h.desc.mv = NULL; h.desc.ref_idx = NULL; h.desc.i4modes = NULL; h.desc.coefs = NULL; h.desc.quads = NULL; h.desc.n_quads = 0;
    memcpy(dst, &h, sizeof h); dst += pad16(sizeof h);
    memcpy(dst, p->mb, n * sizeof(p264hip_mb_t)); dst += pad16(n * sizeof(p264hip_mb_t));
    memcpy(dst, p->mv, n * 64); dst += pad16(n * 64);
    memcpy(dst, p->ref_idx, n * 4); dst += pad16(n * 4);
    memcpy(dst, p->i4modes, n * 16); dst += pad16(n * 16);
    if (p->n_coef_blocks) memcpy(dst, p->coefs, (size_t)p->n_coef_blocks * 32);
}
/* the picture described by a packed message, its arrays pointing into the message */
static int unpack_picture(const uint8_t *src, size_t bytes, p264hip_picture_t *out, int *local_stream)
{
    fan_head_t h;
    if (bytes < sizeof h) return fail("packed picture too short");
    memcpy(&h, src, sizeof h);
    if (h.magic != FAN_MAGIC || h.n_mb != (uint32_t)(h.desc.mb_w * h.desc.mb_h)) return fail("packed picture: bad header");
    *out = h.desc; *local_stream = h.local_stream;
    if (packed_size(out) != bytes) return fail("packed picture: %zu bytes, header says %zu", bytes, packed_size(out));
    const size_t n = h.n_mb;
    src += pad16(sizeof h);
    out->mb = (const p264hip_mb_t *)src; src += pad16(n * sizeof(p264hip_mb_t));
    out->mv = (const int16_t *)src; src += pad16(n * 64);
    out->ref_idx = (const int8_t *)src; src += pad16(n * 4);
    out->i4modes = src; src += pad16(n * 16);
    out->coefs = (const int16_t *)src;
    return 0;
}

/* ---------------------------------------------------------------- default backend ------- */
typedef struct { p264hip_ctx *hip; int mb_w, mb_h; } hipbk_t;
static int hipbk_open(void **ctx, int device, int mb_w, int mb_h, int n_local, int slots)
{
    hipbk_t *b = (hipbk_t *)calloc(1, sizeof *b);
    if (!b) return fail("out of memory");
    if (p264hip_create(&b->hip, device, mb_w, mb_h, n_local, slots, n_local) != P264HIP_OK) { fail("%s", p264hip_last_error()); free(b); return -1; }
    b->mb_w = mb_w; b->mb_h = mb_h;
    *ctx = b;
    return 0;
}
static int hipbk_reconstruct(void *ctx, int s, const p264hip_picture_t *pic, uint8_t *i420)
{
    hipbk_t *b = (hipbk_t *)ctx;
    const int w = b->mb_w * 16, h = b->mb_h * 16;
    if (p264hip_submit(b->hip, s, pic) != P264HIP_OK) return fail("%s", p264hip_last_error());
    if (p264hip_read_frame(b->hip, s, pic->dst_slot, i420, w, i420 + (size_t)w * h, i420 + (size_t)w * h * 5 / 4, w / 2) != P264HIP_OK) return fail("%s", p264hip_last_error());
    return 0;
}
static void hipbk_close(void *ctx) { hipbk_t *b = (hipbk_t *)ctx; if (b) { if (b->hip) p264hip_destroy(b->hip); free(b); } }
static const p264fan_backend_t g_hip_backend = { NULL, hipbk_open, hipbk_reconstruct, hipbk_close };

/* ---------------------------------------------------------------- TCP transport --------- */
typedef struct { int rank, world; int *fd; } tcp_t;        /* fd[peer]; root: one per worker, worker: fd[0] */
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
static void tcp_close(void *c)
{
    tcp_t *t = (tcp_t *)c;
    if (!t) return;
    for (int i = 0; i < t->world; i++) if (t->fd[i] >= 0) close(t->fd[i]);
    free(t->fd); free(t);
}
int p264fan_tcp_transport(p264fan_transport_t *out, int rank, int world, const char *host, int port)
{
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
    out->ctx = t; out->send = tcp_send; out->recv = tcp_recv; out->group_begin = tcp_nop; out->group_end = tcp_nop; out->close = tcp_close; out->name = "tcp";
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

int p264fan_worker_run(p264fan *f)
{
    if (!f || f->rank == 0) return fail("p264fan_worker_run: not a worker");
    uint8_t *msg[FAN_MAX_PER_ROUND] = { 0 }; size_t cap[FAN_MAX_PER_ROUND] = { 0 };
    uint8_t *out = NULL; size_t frame = 0;
    int rc = 0;
    for (;;) {
        fan_ctrl_t c;
        if (gb(f) || f->t.recv(f->t.ctx, 0, &c, sizeof c) || ge(f)) { rc = -1; break; }
        if (c.n == FAN_FINISHED) break;
        if (c.n < 0 || c.n > FAN_MAX_PER_ROUND) { rc = fail("worker %d: bad control block", f->rank); break; }
        if (!f->bk_ctx) {
            if (f->bk.open(&f->bk_ctx, f->device, c.mb_w, c.mb_h, c.n_local_streams, c.slots)) { rc = -1; break; }
            frame = (size_t)c.mb_w * c.mb_h * 384;
            out = (uint8_t *)malloc(frame * FAN_MAX_PER_ROUND);
            if (!out) { rc = fail("out of memory"); break; }
        }
        if (gb(f)) { rc = -1; break; }
        for (int k = 0; k < c.n && !rc; k++) {
            if (c.bytes[k] > cap[k]) { free(msg[k]); cap[k] = c.bytes[k] + c.bytes[k] / 4; msg[k] = (uint8_t *)malloc(cap[k]); if (!msg[k]) { rc = fail("out of memory"); break; } }
            if (f->t.recv(f->t.ctx, 0, msg[k], c.bytes[k])) rc = -1;
        }
        if (ge(f) || rc) { rc = -1; break; }
        for (int k = 0; k < c.n && !rc; k++) {
            p264hip_picture_t pic; int ls = 0;
            if (unpack_picture(msg[k], c.bytes[k], &pic, &ls) || f->bk.reconstruct(f->bk_ctx, ls, &pic, out + frame * (size_t)k)) rc = -1;
        }
        if (rc) break;
        if (gb(f)) { rc = -1; break; }
        for (int k = 0; k < c.n && !rc; k++) if (f->t.send(f->t.ctx, 0, out + frame * (size_t)k, frame)) rc = -1;
        if (ge(f) || rc) { rc = -1; break; }
    }
    for (int k = 0; k < FAN_MAX_PER_ROUND; k++) free(msg[k]);
    free(out);
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

/* the round's pictures are parsed side by side: the streams are independent, and the serial parse is what bounds the
 * fan-out (a 1080p picture takes the parser longer than eight GPUs need to reconstruct one each) */
typedef struct { fstream_t *st; const p264hip_picture_t **pics; int first, step, n_streams, max_pictures, failed; } parse_job_t;
static void *parse_worker(void *arg)
{
    parse_job_t *j = (parse_job_t *)arg;
    for (int s = j->first; s < j->n_streams; s += j->step) j->pics[s] = next_picture(&j->st[s], j->max_pictures, &j->failed);
    return NULL;
}
static int parse_round(fstream_t *st, const p264hip_picture_t **pics, int n_streams, int max_pictures, int threads)
{
    if (threads > n_streams) threads = n_streams;
    if (threads > 64) threads = 64;
    parse_job_t jobs[64];
    pthread_t tid[64];
    int started = 0, failed = 0;
    for (int t = 0; t < threads; t++) jobs[t] = (parse_job_t){ st, pics, t, threads, n_streams, max_pictures, 0 };
    for (int t = 1; t < threads; t++) { if (pthread_create(&tid[t], NULL, parse_worker, &jobs[t])) break; started = t; }
    for (int t = started + 1; t < threads; t++) parse_worker(&jobs[t]);     /* (threads that could not be started: done here) */
    parse_worker(&jobs[0]);
    for (int t = 1; t <= started; t++) pthread_join(tid[t], NULL);
    for (int t = 0; t < threads; t++) failed |= jobs[t].failed;
    return failed;
}

int p264fan_root_run(p264fan *f, int n_streams, const uint8_t *const *annexb, const int64_t *sizes, int max_pictures,
                     p264fan_frame_cb on_frame, void *user, p264fan_stats_t *stats)
{
    if (!f || f->rank != 0 || n_streams < 1 || !annexb || !sizes) return fail("p264fan_root_run: bad argument");
    const int W = f->world;
    if ((n_streams + W - 1) / W > FAN_MAX_PER_ROUND) return fail("p264fan_root_run: more than %d streams per rank", FAN_MAX_PER_ROUND);
    fstream_t *st = (fstream_t *)calloc((size_t)n_streams, sizeof *st);
    fan_ctrl_t *ctrl = (fan_ctrl_t *)calloc((size_t)W, sizeof *ctrl);
    uint8_t **msg = (uint8_t **)calloc((size_t)n_streams, sizeof *msg);      /* packed picture of stream s in this round */
    size_t *cap = (size_t *)calloc((size_t)n_streams, sizeof *cap);
    int *has = (int *)calloc((size_t)n_streams, sizeof *has);
    uint8_t *frames = NULL; size_t frame = 0;
    int rc = (st && ctrl && msg && cap && has) ? 0 : fail("out of memory");
    for (int s = 0; s < n_streams && !rc; s++) {
        st[s].parser = p264parse_open(P264PARSE_OPT_QUIET);
        st[s].in = annexb[s]; st[s].size = sizes[s];
        if (!st[s].parser) rc = fail("p264parse_open failed");
    }
    p264fan_stats_t S; memset(&S, 0, sizeof S); S.world = W;
    int parse_threads = 8;                                    /* P264AMD_FAN_THREADS: host threads parsing a round's pictures */
    { const char *e = getenv("P264AMD_FAN_THREADS"); if (e && atoi(e) >= 1) parse_threads = atoi(e); }
    const double t0 = now_s();
    int mb_w = 0, mb_h = 0, slots = 0;
    while (!rc) {
        /* ---- parse: the next picture of every stream (the serial CPU part; its arrays live until the stream's next call) */
        const double p0 = now_s();
        int n = 0;
        const p264hip_picture_t **pics = (const p264hip_picture_t **)alloca(sizeof(void *) * (size_t)n_streams);
        const int failed = parse_round(st, pics, n_streams, max_pictures, parse_threads);
        for (int s = 0; s < n_streams; s++) { has[s] = pics[s] != NULL; n += has[s]; }
        S.parse_seconds += now_s() - p0;
        if (failed) { rc = fail("a stream failed to parse"); break; }
        if (!n) break;
        if (!mb_w) {
            for (int s = 0; s < n_streams; s++) if (has[s]) { mb_w = pics[s]->mb_w; mb_h = pics[s]->mb_h; slots = p264parse_slots(st[s].parser); break; }
            frame = (size_t)mb_w * mb_h * 384;
            frames = (uint8_t *)malloc(frame * (size_t)n_streams);
            if (!frames) { rc = fail("out of memory"); break; }
            if (f->bk.open(&f->bk_ctx, f->device, mb_w, mb_h, (n_streams + W - 1) / W, slots)) { rc = -1; break; }
        }
        for (int s = 0; s < n_streams; s++) if (has[s] && (pics[s]->mb_w != mb_w || pics[s]->mb_h != mb_h)) { rc = fail("stream %d has a different picture size", s); break; }
        if (rc) break;
        /* ---- scatter: control blocks, then the packed pictures of the remote streams */
        const double e0 = now_s();
        for (int r = 1; r < W; r++) { memset(&ctrl[r], 0, sizeof ctrl[r]); ctrl[r].mb_w = mb_w; ctrl[r].mb_h = mb_h; ctrl[r].slots = slots; ctrl[r].n_local_streams = (n_streams + W - 1) / W; }
        for (int s = 0; s < n_streams; s++) {
            const int r = s % W;
            if (!has[s] || r == 0) continue;
            const size_t need = packed_size(pics[s]);
            if (need > cap[s]) { free(msg[s]); cap[s] = need + need / 4; msg[s] = (uint8_t *)malloc(cap[s]); if (!msg[s]) { rc = fail("out of memory"); break; } }
            pack_picture(msg[s], s / W, pics[s]);
            ctrl[r].bytes[ctrl[r].n++] = (uint32_t)need;
            S.bytes_scattered += (int64_t)need; S.pictures_remote++;
        }
        if (rc || gb(f)) { rc = -1; break; }
        for (int r = 1; r < W && !rc; r++) if (f->t.send(f->t.ctx, r, &ctrl[r], sizeof ctrl[r])) rc = -1;
        if (ge(f) || rc || gb(f)) { rc = -1; break; }
        for (int s = 0; s < n_streams && !rc; s++) if (has[s] && s % W) if (f->t.send(f->t.ctx, s % W, msg[s], packed_size(pics[s]))) rc = -1;
        if (ge(f) || rc) { rc = -1; break; }
        S.exchange_seconds += now_s() - e0;
        /* ---- the root's own streams while the workers are busy */
        for (int s = 0; s < n_streams && !rc; s += W) if (has[s] && f->bk.reconstruct(f->bk_ctx, s / W, pics[s], frames + frame * (size_t)s)) rc = -1;
        if (rc) break;
        /* ---- gather */
        const double g0 = now_s();
        if (gb(f)) { rc = -1; break; }
        for (int s = 0; s < n_streams && !rc; s++) if (has[s] && s % W) { if (f->t.recv(f->t.ctx, s % W, frames + frame * (size_t)s, frame)) rc = -1; S.bytes_gathered += (int64_t)frame; }
        if (ge(f) || rc) { rc = -1; break; }
        S.exchange_seconds += now_s() - g0;
        for (int s = 0; s < n_streams; s++) if (has[s]) { S.pictures++; if (on_frame) on_frame(user, s, st[s].pictures - 1, mb_w * 16, mb_h * 16, frames + frame * (size_t)s); }
        S.rounds++;
    }
    /* ---- tell the workers to leave, whatever happened */
    if (ctrl && f->t.send) {
        gb(f);
        for (int r = 1; r < W; r++) { memset(&ctrl[r], 0, sizeof ctrl[r]); ctrl[r].n = FAN_FINISHED; f->t.send(f->t.ctx, r, &ctrl[r], sizeof ctrl[r]); }
        ge(f);
    }
    S.seconds = now_s() - t0;
    if (stats) *stats = S;
    if (st) for (int s = 0; s < n_streams; s++) { if (st[s].parser) p264parse_close(st[s].parser); free(st[s].rbsp); }
    if (msg) for (int s = 0; s < n_streams; s++) free(msg[s]);
    free(st); free(ctrl); free(msg); free(cap); free(has); free(frames);
    return rc;
}
