/* pipeline.c - multi-stream, multi-threaded decode pipeline on top of the two C ABIs
 * (p264parse.h: host bitstream layer, p264hip.h: MI355X reconstruction).  See include/p264pipe.h.
 *
 * Round r = the next picture of every stream that still has data.  The parser threads pull (round, stream) tasks from one
 * running counter and do NOT stop at the end of a round (round 5; until then they met at a barrier per round and the
 * stragglers of every round cost 4 - 8 % of 16 threads): a task of round R may start once the stream's picture of round R - 1 is
 * parsed and the device has finished with round R - 2, whose buffers the parser is about to reuse (it has two per stream).
 * The main thread waits for the last task of round r, hands the pictures to the GPU - asynchronous uploads out of the parsers'
 * pinned double buffers, one batched reconstruct - and waits for the marker behind them: by then the threads are well into
 * round r + 1, and that wait is what lets them into round r + 2.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "p264pipe.h"
#include "p264parse.h"
#include "p264hip.h"
#include "p264_dropin.h"
#include "host_cpu.h"

typedef struct {
    p264parse *parser;
    const uint8_t *in; int64_t size, pos;
    uint8_t *rbsp; int64_t rbsp_cap;
    const p264hip_picture_t *pic;        /* picture completed in the current round, or NULL */
    int done, failed, last_slot;
    int64_t pictures;
} pstream_t;

struct p264pipe {
    int device, n_streams, n_threads;
    pstream_t *st;
    p264hip_ctx *ctx; int mb_w, mb_h, slots;
    /* thread pool */
    pthread_t *threads; int started;
    pthread_mutex_t mu; pthread_cond_t go, idle, turn;
    int turn_waiters;                    /* (under mu) threads blocked on `turn`: a stream's previous picture is still being parsed */
    int generation, busy, quit, max_pictures;
    /* one run: tasks t = round * n_streams + stream, taken in order */
    long long next_task;                 /* (atomic) */
    int done_rounds;                     /* rounds the device has finished with (under mu; waited for through `go`) */
    int stop;                            /* (atomic) no more tasks */
    int *parsed;                         /* [stream] rounds parsed so far (atomic) */
    int parsed_in_round[2];              /* tasks finished per round parity (under mu; its last one signals `idle`) */
    int failed_in_round[2];              /* a task of the round failed (under mu) */
    const p264hip_picture_t **round_pic[2];  /* [round parity][stream]: the picture a round's task produced, or NULL */
    double parse_seconds;
};

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

/* feed NAL units of one stream until a picture completes or the stream ends */
static void parse_one(p264pipe *p, pstream_t *s)
{
    s->pic = NULL;
    if (s->done || (p->max_pictures > 0 && s->pictures >= p->max_pictures)) { s->done = 1; return; }
    int64_t off, len;
    while (p264_annexb_next(s->in, s->size, &s->pos, &off, &len)) {
        if (len < 1) continue;
        if (len + 8 > s->rbsp_cap) {
            free(s->rbsp); s->rbsp_cap = len * 2 + 64; s->rbsp = (uint8_t *)malloc((size_t)s->rbsp_cap);
            if (!s->rbsp) { __atomic_store_n(&s->failed, 1, __ATOMIC_RELAXED); s->done = 1; return; }
        }
        p264_nal_t nal; nal.p_payload = s->rbsp;
        p264_nal_decode(&nal, (void *)(s->in + off), (int)len);
        const p264hip_picture_t *pic = NULL;
        int rc = p264parse_nal(s->parser, nal.i_type, nal.i_ref_idc, nal.p_payload, nal.i_payload, &pic);
        if (rc < 0) { __atomic_store_n(&s->failed, 1, __ATOMIC_RELAXED); s->done = 1; return; }
        if (rc == 1) { s->pic = pic; s->pictures++; return; }
    }
    s->done = 1;
}

static void *worker(void *arg)
{
    p264pipe *p = (p264pipe *)arg;
    int seen = 0;
    for (;;) {
        pthread_mutex_lock(&p->mu);
        while (p->generation == seen && !p->quit) pthread_cond_wait(&p->go, &p->mu);
        if (p->quit) { pthread_mutex_unlock(&p->mu); return NULL; }
        seen = p->generation;
        pthread_mutex_unlock(&p->mu);
        double spent = 0;
        const int S = p->n_streams;
        while (!__atomic_load_n(&p->stop, __ATOMIC_ACQUIRE)) {
            const long long t = __atomic_fetch_add(&p->next_task, 1, __ATOMIC_RELAXED);
            const int R = (int)(t / S), s = (int)(t % S);
            /* the device is done with round R - 2 (whose buffers this task writes) */
            pthread_mutex_lock(&p->mu);
            while (R > p->done_rounds + 1 && !p->stop) pthread_cond_wait(&p->go, &p->mu);
            pthread_mutex_unlock(&p->mu);
            if (__atomic_load_n(&p->stop, __ATOMIC_ACQUIRE)) break;
            /* the stream's previous picture is parsed (tasks are taken in order, so it nearly always is: a parser is not reentrant) */
            /* (a few yields, then BLOCK: with about as many threads as streams a finished thread routinely draws a stream whose
             * previous picture another thread is still parsing - spinning there burns the CPU quota the parsers need) */
            for (int spins = 0; __atomic_load_n(&p->parsed[s], __ATOMIC_ACQUIRE) < R && !__atomic_load_n(&p->stop, __ATOMIC_ACQUIRE); ) {
                if (++spins <= 32) { sched_yield(); continue; }
                pthread_mutex_lock(&p->mu);
                p->turn_waiters++;
                while (__atomic_load_n(&p->parsed[s], __ATOMIC_ACQUIRE) < R && !__atomic_load_n(&p->stop, __ATOMIC_ACQUIRE)) pthread_cond_wait(&p->turn, &p->mu);
                p->turn_waiters--;
                pthread_mutex_unlock(&p->mu);
            }
            if (__atomic_load_n(&p->stop, __ATOMIC_ACQUIRE)) break;
            double t0 = now_s();
            parse_one(p, &p->st[s]);
            spent += now_s() - t0;
            p->round_pic[R & 1][s] = p->st[s].pic;
            __atomic_store_n(&p->parsed[s], R + 1, __ATOMIC_RELEASE);
            const int failed = __atomic_load_n(&p->st[s].failed, __ATOMIC_RELAXED);
            pthread_mutex_lock(&p->mu);
            if (failed) p->failed_in_round[R & 1] = 1;
            if (p->turn_waiters) pthread_cond_broadcast(&p->turn);     /* (parsed[s] was stored before mu was taken: a waiter has seen it or is waiting) */
            if (++p->parsed_in_round[R & 1] == S) pthread_cond_signal(&p->idle);
            pthread_mutex_unlock(&p->mu);
        }
        pthread_mutex_lock(&p->mu);
        p->parse_seconds += spent;
        if (--p->busy == 0) pthread_cond_broadcast(&p->idle);
        pthread_mutex_unlock(&p->mu);
    }
}

/* release the threads into a run / end it and wait until every thread has left its task loop */
static void start_run(p264pipe *p)
{
    pthread_mutex_lock(&p->mu);
    p->next_task = 0; p->done_rounds = 0; p->stop = 0; p->parsed_in_round[0] = p->parsed_in_round[1] = 0; p->failed_in_round[0] = p->failed_in_round[1] = 0;
    memset(p->parsed, 0, sizeof(int) * (size_t)p->n_streams);
    p->busy = p->n_threads; p->generation++;
    pthread_cond_broadcast(&p->go);
    pthread_mutex_unlock(&p->mu);
}
static void end_run(p264pipe *p)
{
    pthread_mutex_lock(&p->mu);
    __atomic_store_n(&p->stop, 1, __ATOMIC_RELEASE);
    pthread_cond_broadcast(&p->go);
    pthread_cond_broadcast(&p->turn);
    while (p->busy) pthread_cond_wait(&p->idle, &p->mu);
    pthread_mutex_unlock(&p->mu);
}
/* wait for the last task of round r; returns whether one of them failed */
static int finish_round(p264pipe *p, int r)
{
    pthread_mutex_lock(&p->mu);
    while (p->parsed_in_round[r & 1] < p->n_streams) pthread_cond_wait(&p->idle, &p->mu);
    const int failed = p->failed_in_round[r & 1];
    p->parsed_in_round[r & 1] = 0; p->failed_in_round[r & 1] = 0;   /* (nobody starts round r + 2 before round r is through the device) */
    pthread_mutex_unlock(&p->mu);
    return failed;
}
/* the device has finished with rounds 0 .. r: the threads may start round r + 2 */
static void rounds_done(p264pipe *p, int r)
{
    pthread_mutex_lock(&p->mu);
    p->done_rounds = r + 1;
    pthread_cond_broadcast(&p->go);
    pthread_mutex_unlock(&p->mu);
}

p264pipe *p264pipe_open(int device, int n_streams, int n_threads)
{
    if (p264amd_cpu_refuse("p264pipe_open")) return NULL;
    if (n_streams < 1 || n_threads < 1) return NULL;
    if (device >= 0 && p264hip_device_count() <= device) {
        fprintf(stderr, "p264pipe_open: no HIP device %d (there is no CPU fallback for the reconstruction; device -1 runs the parsers only)\n", device);
        return NULL;
    }
    p264pipe *p = (p264pipe *)calloc(1, sizeof *p);
    if (!p) return NULL;
    p->device = device; p->n_streams = n_streams; p->n_threads = n_threads < n_streams ? n_threads : n_streams;
    p->st = (pstream_t *)calloc((size_t)n_streams, sizeof *p->st);
    p->threads = (pthread_t *)calloc((size_t)p->n_threads, sizeof *p->threads);
    p->parsed = (int *)calloc((size_t)n_streams, sizeof(int));
    p->round_pic[0] = (const p264hip_picture_t **)calloc((size_t)n_streams, sizeof(void *));
    p->round_pic[1] = (const p264hip_picture_t **)calloc((size_t)n_streams, sizeof(void *));
    pthread_mutex_init(&p->mu, NULL); pthread_cond_init(&p->go, NULL); pthread_cond_init(&p->idle, NULL); pthread_cond_init(&p->turn, NULL);
    if (!p->st || !p->threads || !p->parsed || !p->round_pic[0] || !p->round_pic[1]) { p264pipe_close(p); return NULL; }
    for (int i = 0; i < n_streams; i++) {
        p->st[i].parser = p264parse_open(P264PARSE_OPT_QUIET);
        if (!p->st[i].parser) { p264pipe_close(p); return NULL; }
        /* picture buffers in pinned host memory when they are uploaded (P264AMD_PIPE_PINNED=0 / 1 forces either kind: experiments) */
        const char *pin = getenv("P264AMD_PIPE_PINNED");
        if (pin ? atoi(pin) != 0 : device >= 0) p264parse_set_allocator(p->st[i].parser, p264hip_host_alloc, p264hip_host_free);
        p->st[i].last_slot = -1;
    }
    for (int i = 0; i < p->n_threads; i++) {
        if (pthread_create(&p->threads[i], NULL, worker, p)) { p264pipe_close(p); return NULL; }
        p->started++;
    }
    return p;
}

int p264pipe_set_input(p264pipe *p, int stream, const uint8_t *annexb, int64_t size)
{
    if (!p || stream < 0 || stream >= p->n_streams || !annexb || size < 0) return -1;
    pstream_t *s = &p->st[stream];
    s->in = annexb; s->size = size; s->pos = 0; s->done = 0; s->failed = 0; s->pictures = 0; s->pic = NULL;
    return 0;
}

int p264pipe_run(p264pipe *p, int max_pictures, p264pipe_stats_t *stats)
{
    if (!p) return -1;
    for (int i = 0; i < p->n_streams; i++) if (!p->st[i].in) { fprintf(stderr, "p264pipe_run: stream %d has no input\n", i); return -1; }
    p->max_pictures = max_pictures; p->parse_seconds = 0;
    int *ids = (int *)malloc(sizeof(int) * (size_t)p->n_streams), *sts = (int *)malloc(sizeof(int) * (size_t)p->n_streams);
    const p264hip_picture_t **pics = (const p264hip_picture_t **)malloc(sizeof(void *) * (size_t)p->n_streams);
    if (!ids || !sts || !pics) { free(ids); free(sts); free(pics); return -1; }
    int rounds = 0, rc = 0;
    int64_t pictures = 0, uploaded = 0;
    double submit = 0, wait_parse = 0, wait_gpu = 0;          /* (the main thread's waits: P264AMD_PIPE_DEBUG=1 prints them) */
    const double t0 = now_s();
    start_run(p);
    for (int r = 0;; r++) {
        { const double w0 = now_s(); if (finish_round(p, r)) rc = -1; wait_parse += now_s() - w0; }
        int n = 0;
        for (int i = 0; i < p->n_streams; i++) {
            pstream_t *s = &p->st[i];
            const p264hip_picture_t *pic = p->round_pic[r & 1][i];
            if (pic) { pics[n] = pic; sts[n] = i; ids[n] = i * 2 + (r & 1); s->last_slot = pic->dst_slot; n++; }
        }
        if (n == 0 || rc) break;
        rounds++; pictures += n;
        if (p->device >= 0 && !p->ctx) {                     /* geometry is known after the first picture */
            p->mb_w = pics[0]->mb_w; p->mb_h = pics[0]->mb_h; p->slots = p264parse_slots(p->st[sts[0]].parser);
            if (p264hip_create(&p->ctx, p->device, p->mb_w, p->mb_h, p->n_streams, p->slots, p->n_streams * 2)) {
                fprintf(stderr, "p264pipe_run: %s\n", p264hip_last_error()); rc = -1; break;
            }
        }
        if (p->ctx) {
            const double s0 = now_s();
            for (int k = 0; k < n && !rc; k++) {
                if (pics[k]->mb_w != p->mb_w || pics[k]->mb_h != p->mb_h) { fprintf(stderr, "p264pipe_run: stream %d has a different picture size\n", sts[k]); rc = -1; }
                else if (p264hip_upload_async(p->ctx, ids[k], pics[k])) { fprintf(stderr, "p264pipe_run: %s\n", p264hip_last_error()); rc = -1; }
                else uploaded += (int64_t)p->mb_w * p->mb_h * (16 + 64 + 4 + 16) + (int64_t)pics[k]->n_coef_blocks * 32;
            }
            if (!rc && p264hip_reconstruct(p->ctx, ids, sts, n)) { fprintf(stderr, "p264pipe_run: %s\n", p264hip_last_error()); rc = -1; }
            int marker = -1;
            if (!rc) { marker = p264hip_marker(p->ctx); if (marker < 0) rc = -1; }
            submit += now_s() - s0;
            if (rc) break;
            /* the parsers reuse round r's buffers in round r + 2: its uploads must have been consumed (the threads are in round r + 1) */
            const double w0 = now_s();
            if (p264hip_marker_wait(p->ctx, marker)) { rc = -1; break; }
            wait_gpu += now_s() - w0;
        }
        rounds_done(p, r);
    }
    end_run(p);
    if (p->ctx && p264hip_sync(p->ctx)) { fprintf(stderr, "p264pipe_run: %s\n", p264hip_last_error()); rc = -1; }
    const double t1 = now_s();
    if (getenv("P264AMD_PIPE_DEBUG"))
        fprintf(stderr, "p264pipe_run: %d rounds, %.3f s: main thread waited %.3f s for the parsers, %.3f s for the device, submitted for %.3f s; parser threads %.3f s in all (%d threads)\n",
                rounds, t1 - t0, wait_parse, wait_gpu, submit, p->parse_seconds, p->n_threads);
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->pictures = pictures; stats->seconds = t1 - t0; stats->parse_seconds = p->parse_seconds; stats->submit_seconds = submit;
        stats->wait_parse_seconds = wait_parse; stats->wait_device_seconds = wait_gpu;
        stats->rounds = rounds; stats->streams = p->n_streams; stats->threads = p->n_threads; stats->bytes_uploaded = uploaded;
        for (int i = 0; i < p->n_streams; i++) stats->bytes += p->st[i].pos;
    }
    free(ids); free(sts); free(pics);
    return rc;
}

int p264pipe_frame_size(p264pipe *p, int *width, int *height)
{
    if (!p || !p->mb_w) return -1;
    if (width) *width = p->mb_w * 16;
    if (height) *height = p->mb_h * 16;
    return 0;
}

int p264pipe_read_frame(p264pipe *p, int stream, uint8_t *y, int y_stride, uint8_t *u, uint8_t *v, int c_stride)
{
    if (!p || !p->ctx || stream < 0 || stream >= p->n_streams || p->st[stream].last_slot < 0) return -1;
    return p264hip_read_frame(p->ctx, stream, p->st[stream].last_slot, y, y_stride, u, v, c_stride) ? -1 : 0;
}

int64_t p264pipe_stream_pictures(p264pipe *p, int stream)
{
    return (p && stream >= 0 && stream < p->n_streams) ? p->st[stream].pictures : -1;
}

void p264pipe_close(p264pipe *p)
{
    if (!p) return;
    pthread_mutex_lock(&p->mu); p->quit = 1; pthread_cond_broadcast(&p->go); pthread_mutex_unlock(&p->mu);
    for (int i = 0; i < p->started; i++) pthread_join(p->threads[i], NULL);
    if (p->ctx) { (void)p264hip_sync(p->ctx); }
    if (p->st) for (int i = 0; i < p->n_streams; i++) { if (p->st[i].parser) p264parse_close(p->st[i].parser); free(p->st[i].rbsp); }
    if (p->ctx) p264hip_destroy(p->ctx);
    pthread_mutex_destroy(&p->mu); pthread_cond_destroy(&p->go); pthread_cond_destroy(&p->idle); pthread_cond_destroy(&p->turn);
    free(p->parsed); free((void *)p->round_pic[0]); free((void *)p->round_pic[1]);
    free(p->st); free(p->threads); free(p);
}
