/* p264pipe.h - C ABI of the multi-stream decode pipeline.
 *
 * N Annex-B streams are decoded side by side: a pool of host threads runs the CAVLC parsers (one stream at a
 * time per thread, the part of the decoder that stays on the CPU), and the next picture of every stream goes to the
 * MI355X as ONE batch (p264hip_upload_async + p264hip_reconstruct).  The parsers write their picture arrays straight
 * into pinned memory (registered huge pages, p264hip_host_alloc), so the uploads are plain DMA; while the GPU works on round r
 * the threads already parse round r+1 - they do not stop at the end of a round: a stream's next picture may be parsed as
 * soon as its previous one is and the device has finished with the round before that (whose buffers the parser reuses).  Output pictures stay in HBM (the frame stores of p264hip); p264pipe_read_frame fetches the last one of a
 * stream.  There is no counterpart in the reference (its decoder is single-stream, single-threaded,
 * p264decoder.c:164-381); the per-stream behaviour is that of p264_decoder_decode.
 *
 * device < 0 runs the parsers only (no GPU is touched): the host-side ceiling of the pipeline.
 */
#ifndef P264PIPE_H
#define P264PIPE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct p264pipe p264pipe;

typedef struct {
    int64_t pictures;            /* pictures decoded, all streams */
    int64_t bytes;               /* Annex-B bytes consumed */
    double  seconds;             /* wall clock of p264pipe_run */
    double  parse_seconds;       /* summed over the threads: time inside the parsers */
    double  submit_seconds;      /* host time spent enqueuing uploads and launches */
    int     rounds, streams, threads;
    int     reserved;
    int64_t bytes_uploaded;      /* parsed-picture bytes copied host -> HBM (0 when the parsers run alone) */
    double  wait_parse_seconds;  /* the main thread's wait for the last parse task of a round, summed (parser-bound run: most of `seconds`) */
    double  wait_device_seconds; /* ... and for the device to be through a round's uploads and kernels (a GPU-bound run shows here) */
} p264pipe_stats_t;

p264pipe *p264pipe_open(int device, int n_streams, int n_threads);
/* The stream's bytes are borrowed until p264pipe_close.  All streams must have the same picture size. */
int  p264pipe_set_input(p264pipe *p, int stream, const uint8_t *annexb, int64_t size);
/* Decode every stream to its end (or max_pictures per stream, 0 = no limit).  0 on success, -1 on error. */
int  p264pipe_run(p264pipe *p, int max_pictures, p264pipe_stats_t *stats);
/* Last decoded picture of a stream (MB-aligned planes, as p264_decoder_decode returns them). */
int  p264pipe_frame_size(p264pipe *p, int *width, int *height);
int  p264pipe_read_frame(p264pipe *p, int stream, uint8_t *y, int y_stride, uint8_t *u, uint8_t *v, int c_stride);
int64_t p264pipe_stream_pictures(p264pipe *p, int stream);
void p264pipe_close(p264pipe *p);

#ifdef __cplusplus
}
#endif
#endif
