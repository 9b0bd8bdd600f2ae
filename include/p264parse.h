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

/* Known-answer surface of the B-picture derivations (the macroblock layer's own static functions, run on hand-made state;
 * tests/test_direct_kat.py drives them with vectors recorded from the reference's encoder-side
 * p264_macroblock_bipred_init / p264_mb_predict_mv_direct16x16, core/macroblock.c:1400-1430, 254-413).
 *
 * p264parse_kat_bipred: implicit weights (H.264 8.4.2.3.1) of every (list-0, list-1) index pair, weights[r0 * 16 + r1] = weight
 * of the list-0 prediction, from the picture order counts of the list entries (n0, n1 <= 8) and of the current picture.
 *
 * p264parse_kat_direct: direct prediction (8.4.1.2) of one macroblock.  nb_ref[l * 4 + n], nb_mv[(l * 4 + n) * 2 + c]: reference
 * index and vector of the macroblock's neighbours n = A (left), B (top), C (top right), D (top left) in list l (-2 = not
 * available, -1 = intra / list unused); col_*: the co-located macroblock of RefPicList1[0] - intra or not, col_ref[l * 4 + q]
 * per 8x8 quadrant, col_mv[(l * 16 + b) * 2 + c] per 4x4 block in raster order; poc0[n0]: order counts of the current list 0,
 * poc1_0 of RefPicList1[0], cur_poc of the current picture; col_list_poc[n_col_list]: order counts of the list 0 the co-located
 * picture was decoded with.  direct_8x8_inference off, no long-term pictures.  out_ref[l * 4 + q], out_mv[(l * 16 + b) * 2 + c].
 * Both return 0, or -1 on a bad argument. */
int p264parse_kat_bipred(int n0, const int *poc0, int n1, const int *poc1, int cur_poc, int16_t *weights);
int p264parse_kat_direct(int spatial, const int8_t *nb_ref, const int16_t *nb_mv, int col_intra, const int8_t *col_ref, const int16_t *col_mv,
                         int n0, const int *poc0, int poc1_0, int cur_poc, int n_col_list, const int *col_list_poc,
                         int8_t *out_ref, int16_t *out_mv);

#ifdef __cplusplus
}
#endif
#endif
