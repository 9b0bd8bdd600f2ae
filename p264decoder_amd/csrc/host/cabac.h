/* cabac.h - the CABAC arithmetic decoding engine of the host entropy layer (H.264 9.3.1.1, 9.3.1.2, 9.3.3.2).
 *
 * Replaces p264_cabac_context_init / p264_cabac_decode_init / _decision / _bypass / _terminal of the reference
 * (core/cabac.c:819-902).  Own formulation: the standard's (pStateIdx, valMPS) context state packed into one byte
 * (pStateIdx << 1 | valMPS), codIOffset kept scaled by the bits read ahead (32 at a time), renormalisation by a
 * leading-zero count instead of a bit loop, the LPS / MPS decision as a mask instead of a branch.
 * This is the first piece of the Main-profile entropy layer: binarisation and context selection of the
 * macroblock layer sit on top of it (the reference's own CABAC macroblock parse is a stub, decoder/macroblock.c:594-597).
 * Pinned by a known-answer test: bins ENCODED by the reference's p264_cabac_encode_* (core/cabac.c:907-1018) must decode
 * to the same bins here (tests/test_cabac_kat.py, tests/golden/kat_cabac.npz).
 */
#ifndef P264_CABAC_H
#define P264_CABAC_H
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include "cabac_tables.h"

#define P264_CABAC_CONTEXTS 460

typedef struct p264cabac {
    /* The standard's codIOffset is kept SCALED: low = codIOffset << n | the next n bits of the data, so that a
     * renormalisation by k bits is "n -= k" and the data is touched once per 32 bits.  Every operation leaves n >= 8 (no
     * renormalisation shifts by more than 7: the smallest rangeTabLPS is 6). */
    uint32_t range;                    /* codIRange (9 bits) */
    int      n;
    uint64_t low;
    const uint8_t *data;
    size_t   size, pos;                /* bytes of the data / bytes taken so far (counts on behind the end: zeros) */
    uint8_t  next[128][2];             /* state byte behind a bin decoded as the MPS / as the LPS */
    /* (16-bit entries although a byte would do: a byte store may alias every field of this struct, and the compiler then reloads
     * range, n and low from memory after each bin; with a 16-bit store they stay in registers across a run of bins) */
    uint16_t state[P264_CABAC_CONTEXTS];   /* pStateIdx << 1 | valMPS */
} p264cabac_t;

/* 9.3.1.1: every context from its (m, n) pair.  is_i_slice: table of I slices, else of cabac_init_idc (0..2). */
void p264cabac_init_contexts(p264cabac_t *c, int is_i_slice, int cabac_init_idc, int slice_qp);
/* 9.3.1.2: start at a byte-aligned position of the slice data */
void p264cabac_start(p264cabac_t *c, const uint8_t *data, size_t bytes);

/* bits of the data not yet consumed; negative = the decoder used bits from behind the end */
static inline int64_t p264cabac_bits_left(const p264cabac_t *c) { return (int64_t)c->size * 8 - ((int64_t)c->pos * 8 - c->n); }

static inline void p264cabac_refill(p264cabac_t *c)       /* 32 more bits behind low (n < 8 before: low stays below 2^50) */
{
    uint32_t v = 0;
    if (c->pos + 4 <= c->size) { memcpy(&v, c->data + c->pos, 4); v = __builtin_bswap32(v); }
    else for (int i = 0; i < 4; i++) v = v << 8 | (c->pos + (size_t)i < c->size ? c->data[c->pos + (size_t)i] : 0u);
    c->pos += 4;
    c->low = c->low << 32 | v;
    c->n += 32;
}

/* 9.3.3.2.1: one context-coded bin.  Branch-free up to the refill: the LPS / MPS decision is a mask. */
static inline int p264cabac_decision(p264cabac_t *c, int ctx)
{
    const uint32_t s = c->state[ctx];
    const uint32_t lps = cabac_range_lps[s >> 1][(c->range >> 6) & 3];
    uint32_t range = c->range - lps;
    const uint64_t scaled = (uint64_t)range << c->n;
    const uint64_t is_lps = (uint64_t)0 - (uint64_t)(c->low >= scaled);           /* all ones: least probable symbol */
    c->low -= scaled & is_lps;
    range = (range & ~(uint32_t)is_lps) | (lps & (uint32_t)is_lps);
    c->state[ctx] = c->next[s][is_lps & 1];
    const int k = __builtin_clz(range) - 23;                                        /* 9.3.3.2.2 renormalisation, all its shifts at once */
    c->range = range << k;
    c->n -= k;
    if (c->n < 8) p264cabac_refill(c);
    return (int)((s ^ (uint32_t)is_lps) & 1u);
}
/* 9.3.3.2.3: one equiprobable bin */
static inline int p264cabac_bypass(p264cabac_t *c)
{
    c->n -= 1;
    const uint64_t scaled = (uint64_t)c->range << c->n;
    const uint64_t one = (uint64_t)0 - (uint64_t)(c->low >= scaled);               /* (a mask, not a branch: bypass bins are coin flips) */
    c->low -= scaled & one;
    if (c->n < 8) p264cabac_refill(c);
    return (int)(one & 1u);
}
/* 9.3.3.2.2.x: end_of_slice_flag / the bin in front of I_PCM samples */
static inline int p264cabac_terminate(p264cabac_t *c)
{
    c->range -= 2;
    if (c->low >= (uint64_t)c->range << c->n) return 1;
    if (c->range < 256) { c->range <<= 1; c->n -= 1; if (c->n < 8) p264cabac_refill(c); }
    return 0;
}
#endif
