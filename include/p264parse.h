/* p264parse.h - C ABI of the host-side bitstream layer.
 *
 * The serial, branchy part of the decoder that stays on the CPU (north star): NAL
 * dispatch, SPS/PPS, slice header, reference-list / frame-store bookkeeping and the
 * CAVLC macroblock-layer parse with MV / intra-mode / nC prediction.  Its product is one
 * p264hip_picture_t per coded picture - exactly the buffers the GPU layer consumes.
 *
 * Stands in for, in the reference: decoder/decoder.c:70-301,368-593,745-806 (NAL switch,
 * slice header, MB loop), decoder/set.c:37-272, decoder/lists.c:72-228,
 * decoder/macroblock.c:72-592, decoder/dec_cavlc.c:1371-1524 and the neighbour cache of
 * core/macroblock.c:40-252,870-1398.
 */
#ifndef P264PARSE_H
#define P264PARSE_H
#include <stdint.h>
#include <stddef.h>
#include "p264hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct p264parse p264parse;

enum {
    P264PARSE_OPT_QUIET  = 1,   /* do not print SPS/PPS/size lines to stderr (the reference prints them) */
    P264PARSE_OPT_STRICT = 2    /* H.264-conformant QP accumulation also for Baseline CAVLC streams, where the default is the
                                   reference's rule (decoder/macroblock.c:568, core/macroblock.c:1247-1252; SURVEY A-Q2).
                                   CABAC, B slices and every profile but Baseline - none of which the reference decodes -
                                   always get the conformant chain */
};

p264parse *p264parse_open(int options);
/* Where the arrays of the picture descriptors live (default: malloc/free).  A pipeline passes pinned-memory
 * functions so that the parser writes straight into DMA-able staging.  Call before the first slice NAL. */
void       p264parse_set_allocator(p264parse *p, void *(*alloc)(size_t bytes), void (*release)(void *ptr));
void       p264parse_close(p264parse *p);

/* Feed one NAL unit (header byte already split off, emulation-prevention bytes already
 * removed - see p264_nal_decode).  Returns <0 on error, 0 if no picture was completed,
 * 1 if *pic now describes a complete picture.  The descriptor and its arrays stay valid
 * until the next call that completes a picture. */
int p264parse_nal(p264parse *p, int nal_type, int nal_ref_idc,
                  const uint8_t *payload, int size, const p264hip_picture_t **pic);

/* geometry / frame-store size of the active SPS (0 before the first slice) */
int p264parse_mb_width(const p264parse *p);
int p264parse_mb_height(const p264parse *p);
int p264parse_slots(const p264parse *p);          /* num_ref_frames + 1 */
/* bumped every time a new SPS/PPS pair is activated (context re-init, decoder.c:304-343) */
int p264parse_generation(const p264parse *p);

/* Annex-B helper: find the next NAL unit in buf[*pos..size).  On success returns 1 and sets
 * *nal_off / *nal_len (start code and its leading zeros excluded, like p264decoder.c:259-325);
 * returns 0 at end of buffer. */
int p264_annexb_next(const uint8_t *buf, int64_t size, int64_t *pos, int64_t *nal_off, int64_t *nal_len);

/* The CABAC arithmetic decoding engine of the entropy layer, driven bin by bin (csrc/host/cabac.h; replaces
 * p264_cabac_context_init / _decode_init / _decode_decision / _decode_bypass / _decode_terminal, core/cabac.c:819-902).
 * Contexts are initialised for the slice (9.3.1.1), decoding starts at data[0] (9.3.1.2), then one bin per op:
 * op >= 0: a decision with context op (< 460); -1: a bypass bin; -2: a terminate bin.  bins[i] receives bin i.
 * Returns 0, 1 if the ops read past the end of the data (bins beyond it are then undefined), -1 on a bad argument.
 * This is what the known-answer test of the engine drives; the macroblock-layer binarisations are built on the same
 * inline functions. */
int p264cabac_decode_ops(const uint8_t *data, size_t bytes, int is_i_slice, int cabac_init_idc, int slice_qp,
                         const int16_t *ops, int n_ops, uint8_t *bins);

#ifdef __cplusplus
}
#endif
#endif
