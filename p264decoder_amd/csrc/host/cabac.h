/* cabac.h - the CABAC arithmetic decoding engine of the host entropy layer (H.264 9.3.1.1, 9.3.1.2, 9.3.3.2).
 *
 * Replaces p264_cabac_context_init / p264_cabac_decode_init / _decision / _bypass / _terminal of the reference
 * (core/cabac.c:819-902).  Own formulation: the standard's (pStateIdx, valMPS) context state packed into one byte
 * (pStateIdx << 1 | valMPS), a 64-bit bit reservoir, renormalisation by a leading-zero count instead of a bit loop.
 * This is the first piece of the Main-profile entropy layer: binarisation and context selection of the
 * macroblock layer sit on top of it (the reference's own CABAC macroblock parse is a stub, decoder/macroblock.c:594-597).
 * Pinned by a known-answer test: bins ENCODED by the reference's p264_cabac_encode_* (core/cabac.c:907-1018) must decode
 * to the same bins here (tests/test_cabac_kat.py, tests/golden/kat_cabac.npz).
 */
#ifndef P264_CABAC_H
#define P264_CABAC_H
#include <stdint.h>
#include <stddef.h>
#include "cabac_tables.h"

#define P264_CABAC_CONTEXTS 460

typedef struct p264cabac {
    uint32_t range, offset;            /* codIRange (9 bits), codIOffset */
    uint64_t cache;                    /* upcoming bits, MSB first */
    int      cache_bits;
    const uint8_t *p, *end;
    int64_t  bits_left;                /* bits of the data not yet consumed; negative = the decoder read past the end */
    uint8_t  state[P264_CABAC_CONTEXTS];   /* pStateIdx << 1 | valMPS */
} p264cabac_t;

/* 9.3.1.1: every context from its (m, n) pair.  is_i_slice: table of I slices, else of cabac_init_idc (0..2). */
void p264cabac_init_contexts(p264cabac_t *c, int is_i_slice, int cabac_init_idc, int slice_qp);
/* 9.3.1.2: start at a byte-aligned position of the slice data */
void p264cabac_start(p264cabac_t *c, const uint8_t *data, size_t bytes);

static inline void p264cabac_refill(p264cabac_t *c)
{
    while (c->cache_bits <= 56) {
        uint64_t b = 0;
        if (c->p < c->end) b = *c->p++;                   /* (zeros behind the end; bits_left tells whether any were USED) */
        c->cache |= b << (56 - c->cache_bits);
        c->cache_bits += 8;
    }
}
static inline uint32_t p264cabac_bits(p264cabac_t *c, int n)       /* 1 <= n <= 16 */
{
    if (c->cache_bits < n) p264cabac_refill(c);
    const uint32_t v = (uint32_t)(c->cache >> (64 - n));
    c->cache <<= n; c->cache_bits -= n; c->bits_left -= n;
    return v;
}

/* 9.3.3.2.1: one context-coded bin */
static inline int p264cabac_decision(p264cabac_t *c, int ctx)
{
    uint32_t s = c->state[ctx];
    const uint32_t lps = cabac_range_lps[s >> 1][(c->range >> 6) & 3];
    int bin = (int)(s & 1);
    c->range -= lps;
    if (c->offset >= c->range) {                           /* least probable symbol */
        c->offset -= c->range;
        c->range = lps;
        bin ^= 1;
        if ((s >> 1) == 0) s ^= 1;                         /* pStateIdx 0: the MPS flips */
        s = (uint32_t)cabac_trans_lps[s >> 1] << 1 | (s & 1);
    } else if (s < 124) s += 2;                            /* transIdxMPS = min(pStateIdx + 1, 62) */
    c->state[ctx] = (uint8_t)s;
    if (c->range < 256) {                                  /* 9.3.3.2.2 renormalisation, all its shifts at once */
        const int n = __builtin_clz(c->range) - 23;
        c->range <<= n;
        c->offset = c->offset << n | p264cabac_bits(c, n);
    }
    return bin;
}
/* 9.3.3.2.3: one equiprobable bin */
static inline int p264cabac_bypass(p264cabac_t *c)
{
    c->offset = c->offset << 1 | p264cabac_bits(c, 1);
    if (c->offset >= c->range) { c->offset -= c->range; return 1; }
    return 0;
}
/* 9.3.3.2.2.x: end_of_slice_flag / the bin in front of I_PCM samples */
static inline int p264cabac_terminate(p264cabac_t *c)
{
    c->range -= 2;
    if (c->offset >= c->range) return 1;
    if (c->range < 256) { c->range <<= 1; c->offset = c->offset << 1 | p264cabac_bits(c, 1); }
    return 0;
}
#endif
