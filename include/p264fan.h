/* p264fan.h - C ABI of the stream fan-out (BASELINE.json configs[4], SURVEY 8e "Collective").
 *
 * One rank - the ROOT, rank 0 - owns the Annex-B inputs and the I420 outputs, as the reference's CLI does for its single
 * stream (p264decoder.c:164-381: read file, decode, write_frame).  It runs the CAVLC parsers and SCATTERS every parsed
 * picture (the arrays of p264hip_picture_t, typically ~1.5 MB for 1080p) to the rank that owns the picture's stream -
 * stream s lives on rank s % world with its own frame store, because P pictures need the stream's previous pictures - and
 * GATHERS the reconstructed picture (3 133 440 bytes of MB-aligned I420 for 1080p) back.  There is no counterpart in the
 * reference: it is single-threaded and single-stream (core/core.c:48; core/core.h:239 is never populated).
 *
 * The exchange is point to point - RCCL has no scatter / gather primitive - through a small transport interface:
 *   rccl  grouped ncclSend / ncclRecv over xGMI, one process per GPU (librccl is loaded on demand); the host buffers of
 *         this interface are staged through device memory on both ends
 *   tcp   blocking sockets (127.0.0.1 or any host): the same protocol without RCCL - world-size-2 tests on CPUs,
 *         or ranks on different machines
 * and the reconstruction of a picture goes through a backend interface whose default is this library's MI355X path
 * (p264hip_create / p264hip_submit / p264hip_read_frame on the rank's device).  A rank without a HIP device fails
 * loudly: there is no CPU reconstruction in the product (tests plug the CPU oracle in through the backend interface).
 *
 * Per round the root sends every worker ONE control block (how many pictures, their sizes, or "finished"), then ONE message
 * with the pictures' descriptors (host memory on both sides: the worker needs them to lay its input slots out) and then
 * the pictures' arrays, each packed as ONE block in the layout of an input slot (p264hip_input_layout_t, include/p264hip.h);
 * the worker reconstructs them and answers with ONE fixed-size status block (0 or its error text) followed, when
 * the status is 0, by the planes.  A worker that fails keeps answering (with its error) until the root says "finished",
 * which it only does where a worker expects a control block: no failure on either side leaves the other one waiting
 * inside a round.  A rank that cannot even stay in step (no memory to receive a message into, a control block out of range)
 * aborts the transport before it returns.  An abort is LOCAL (ncclCommAbort tears down the caller's communicator; TCP: the
 * peers do see "peer closed"): over RCCL the peers of a rank that left end through their own deadline - the RCCL transport
 * polls every group's completion against P264AMD_FAN_TIMEOUT_S (default 30 s) and the communicator's asynchronous error,
 * then aborts - so a dead peer ends the job with an error, not with a hang.  Not bounded: ncclGroupEnd itself can block in
 * the first-use connection handshake with a peer that never arrives.  All sends and receives of a step are posted between
 * group_begin / group_end (ncclGroupStart / ncclGroupEnd).  Rounds are double-buffered on the root: while round r is exchanged and reconstructed, round r+1 is
 * parsed (one host thread per stream) and packed.
 */
#ifndef P264FAN_H
#define P264FAN_H
#include <stdint.h>
#include <stddef.h>
#include "p264hip.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct p264fan p264fan;

/* Point-to-point transport between ranks.  Buffers are host memory; a call may return before the transfer has happened
 * when it is issued inside group_begin / group_end, which completes everything posted in between.  0 = ok. */
typedef struct p264fan_transport {
    void *ctx;
    int (*send)(void *ctx, int peer, const void *buf, size_t bytes);
    int (*recv)(void *ctx, int peer, void *buf, size_t bytes);
    int (*group_begin)(void *ctx);
    int (*group_end)(void *ctx);
    void (*close)(void *ctx);
    const char *name;
    /* optional: give up on the peers NOW - this rank's pending and later calls fail instead of waiting (RCCL: ncclCommAbort,
     * local: the peers run into their own deadline; TCP: the sockets are shut down, the peers see "peer closed").  A rank
     * that cannot go on inside a round (out of memory for a message, a control block that makes no sense) calls it before
     * it leaves. */
    void (*abort)(void *ctx);
    /* optional: the same for buffers in DEVICE memory of the rank's GPU, no staging (RCCL: ncclSend / ncclRecv straight from /
     * into the buffer).  Where both the transport and the backend offer their device entry points a worker receives a
     * picture's arrays straight into the input slot it is reconstructed from and sends the reconstructed planes straight
     * out of the conversion buffer: nothing of a round's data touches the worker's host memory.  The buffers must be
     * complete (send) / unused (recv) on the device when the call is made; they are done with when the group ends. */
    int (*send_dev)(void *ctx, int peer, const void *dev, size_t bytes);
    int (*recv_dev)(void *ctx, int peer, void *dev, size_t bytes);
} p264fan_transport_t;

/* Reconstruction of one picture of one local stream; i420 receives the MB-aligned planes Y, U, V back to back. */
typedef struct p264fan_backend {
    void *ctx;
    int (*open)(void **ctx, int device, int mb_w, int mb_h, int n_local_streams, int slots);
    int (*reconstruct)(void *ctx, int local_stream, const p264hip_picture_t *pic, uint8_t *i420);
    void (*close)(void *ctx);
    /* optional (NULL: reconstruct() is synchronous): reconstruct() may only enqueue its work; sync() returns when every
     * picture handed over since the last sync() is in its i420 buffer.  The picture's arrays and the buffers stay valid
     * until then. */
    int (*sync)(void *ctx);
    /* optional, all three or none - the device road of a worker (see p264fan_transport_t.send_dev):
     * reserve: where the packed arrays (p264hip_input_layout_t) of local stream's next picture are to be written, in device
     *          memory; reconstruct_reserved: they are there - note the picture like reconstruct() does (no i420 buffer: the
     *          planes stay on the device); planes: after sync(), the device address of the planar I420 frame of the k-th
     *          picture noted since the sync() before. */
    int (*reserve)(void *ctx, int local_stream, const p264hip_picture_t *desc, void **dev, size_t *bytes);
    int (*reconstruct_reserved)(void *ctx, int local_stream, const p264hip_picture_t *desc);
    int (*planes)(void *ctx, int k, void **dev, size_t *bytes);
} p264fan_backend_t;

typedef struct {
    int64_t pictures, pictures_remote;   /* decoded in total / on other ranks */
    int64_t bytes_scattered, bytes_gathered;
    double  seconds, parse_seconds, exchange_seconds;
    int     rounds, world;
    /* rounds are double-buffered: round r+1 is parsed and packed (parse_seconds, on parse_threads host threads) while round
     * r is exchanged and reconstructed; parse_wait_seconds is the part of the parse the exchange side had to wait for (the
     * rest was hidden), reconstruct_seconds the root's own reconstruction */
    double  parse_wait_seconds, reconstruct_seconds;
    int     parse_threads;
    int     device_road_rounds;          /* (worker, round) pairs served the device road: pictures into their input slots, planes out of the conversion buffers */
} p264fan_stats_t;

/* called on the root for every reconstructed picture, in decode order per stream */
typedef void (*p264fan_frame_cb)(void *user, int stream, int64_t picture, int width, int height, const uint8_t *i420);

/* transports.  (P264AMD_FAN_TCP_DEVICE=1 makes the TCP transport offer send_dev / recv_dev too, through a host bounce buffer
 * and p264hip_copy_*: the device road of the protocol can then be exercised by ranks that share one GPU, where RCCL cannot
 * form a communicator.) */
int  p264fan_tcp_transport(p264fan_transport_t *t, int rank, int world, const char *root_host, int port);
int  p264fan_rccl_unique_id(uint8_t id[128]);                                      /* on one rank; hand the bytes to all */
int  p264fan_rccl_transport(p264fan_transport_t *t, int rank, int world, const uint8_t id[128], int device);

/* backend = NULL: the MI355X path of this library on `device` */
p264fan *p264fan_open(int rank, int world, const p264fan_transport_t *t, const p264fan_backend_t *backend, int device);
/* rank 0: decode n_streams Annex-B streams (all of one picture size) to their end or max_pictures each (0 = all) */
int  p264fan_root_run(p264fan *f, int n_streams, const uint8_t *const *annexb, const int64_t *sizes, int max_pictures,
                      p264fan_frame_cb on_frame, void *user, p264fan_stats_t *stats);
/* every other rank: serve the root until it says "finished" */
int  p264fan_worker_run(p264fan *f);
void p264fan_close(p264fan *f);           /* closes the transport too */
const char *p264fan_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
