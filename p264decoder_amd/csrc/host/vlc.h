/* vlc.h - two-level prefix-code lookup trees built at start-up from (length, code)
 * tables, plus the CAVLC residual-block reader.
 *
 * Stands in for the hand-unrolled decode trees of the reference
 * (decoder/dec_cavlc.c:786-1369) and for p264dec_read_residual_block_cavlc
 * (decoder/dec_cavlc.c:1371-1524).
 */
#ifndef P264_VLC_H
#define P264_VLC_H
#include <stdint.h>
#include "bits.h"

#define VLC_ROOT_BITS 8

typedef struct { int16_t sym; uint8_t len; uint8_t sub; } vlc_ent_t;   /* sub>0: sym = offset of a sub-table of 2^sub entries */
typedef struct { vlc_ent_t *ent; int n_ent; } vlc_t;

/* Build from n (len, code, sym) triples; len==0 entries are skipped.  Returns 0 / -1. */
int  vlc_build(vlc_t *v, int n, const uint8_t *len, const uint16_t *code, const int16_t *sym);
void vlc_free(vlc_t *v);

static inline int vlc_get(bitrd_t *b, const vlc_t *v)
{
    vlc_ent_t e = v->ent[br_peek(b, VLC_ROOT_BITS)];
    if (e.sub) {
        br_skip(b, VLC_ROOT_BITS);
        e = v->ent[e.sym + (int)br_peek(b, e.sub)];
    }
    if (e.len == 0) return -1;           /* invalid code */
    br_skip(b, e.len);
    return e.sym;
}

/* all CAVLC trees (process-wide, built once) */
int cavlc_global_init(void);

/* Read one residual block.  nC: predicted number of coefficients, or -1 for chroma DC.
 * max_coeff: 16 (full 4x4 / luma DC), 15 (AC), 4 (chroma DC).  When the block has coefficients, out[0 .. 16) (chroma DC: out[0 .. 4)) is
 * cleared and the levels are written at their scan positions; an empty block leaves out[] untouched.
 * Returns total_coeff (0..16) or -1 on a broken stream. */
int cavlc_read_block(bitrd_t *b, int nC, int max_coeff, int16_t *out);

#endif
