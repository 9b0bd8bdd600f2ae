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
#include <string.h>
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

/* the trees (process-wide, built once by cavlc_global_init; internal to the library) */
#pragma GCC visibility push(hidden)
extern vlc_t p264_vlc_ct[3], p264_vlc_ctdc, p264_vlc_tz[15], p264_vlc_tzdc[3], p264_vlc_rb[7];
#pragma GCC visibility pop

/* Read one residual block.  nC: predicted number of coefficients, or -1 for chroma DC.
 * max_coeff: 16 (full 4x4 / luma DC), 15 (AC), 4 (chroma DC).  When the block has coefficients, out[0 .. 16) (chroma DC: out[0 .. 4)) is
 * cleared and the levels are written at their scan positions; an empty block leaves out[] untouched.
 * Returns total_coeff (0..16) or -1 on a broken stream.
 * (Inline since round 5: called twenty-six times per macroblock, and as a function of its own the bit reader's window, bit
 * count and position went through memory at every call.) */
static inline int cavlc_read_block(bitrd_t *b, int nC, int max_coeff, int16_t *out)
{
    int tc, t1;
    if (nC < 0) {
        int s = vlc_get(b, &p264_vlc_ctdc);
        if (s < 0) return -1;
        tc = s & 31; t1 = s >> 5;
    } else if (nC >= 8) {                       /* 6-bit FLC */
        int v = (int)br_u(b, 6);
        if (v == 3) { tc = 0; t1 = 0; }
        else { tc = (v >> 2) + 1; t1 = v & 3; if (t1 > tc) return -1; }
    } else {
        /* the empty block first: its code is all ones, 1 / 2 / 4 bits long in the three tables (H.264 table 9-5), and it
         * is what most calls find */
        const int t = nC < 2 ? 0 : nC < 4 ? 1 : 2, n1 = t == 0 ? 1 : t == 1 ? 2 : 4;
        if (br_peek(b, n1) == (1u << n1) - 1u) { br_skip(b, n1); return 0; }
        int s = vlc_get(b, &p264_vlc_ct[t]);
        if (s < 0) return -1;
        tc = s & 31; t1 = s >> 5;
    }
    if (tc == 0) return 0;
    if (tc > max_coeff) return -1;

    int level[16];
    int suffix_len = (tc > 10 && t1 < 3) ? 1 : 0;
    if (t1) {                                               /* the trailing ones' signs, all at once */
        const uint32_t sg = br_u(b, t1);
        for (int i = 0; i < t1; i++) level[i] = ((sg >> (t1 - 1 - i)) & 1) ? -1 : 1;
    }
    for (int i = t1; i < tc; i++) {
        const uint32_t w = br_peek(b, 32);                  /* level_prefix: zeros up to the first 1, counted in one step */
        const int prefix = w ? __builtin_clz(w) : 32;
        if (prefix >= 32) return -1;                        /* (no level prefix is that long: a truncated or broken stream) */
        br_skip(b, prefix + 1);
        if (br_overrun(b)) return -1;
        int sufbits = suffix_len;
        if (prefix == 14 && suffix_len == 0) sufbits = 4;
        else if (prefix >= 15) sufbits = prefix - 3;
        int code = ((prefix < 15 ? prefix : 15) << suffix_len) + (sufbits ? (int)br_u(b, sufbits) : 0);
        if (prefix >= 15 && suffix_len == 0) code += 15;
        if (prefix >= 16) code += (1 << (prefix - 3)) - 4096;
        if (i == t1 && t1 < 3) code += 2;
        level[i] = (code & 1) ? (-code - 1) >> 1 : (code + 2) >> 1;
        if (suffix_len == 0) suffix_len = 1;
        int a = level[i] < 0 ? -level[i] : level[i];
        if (a > (3 << (suffix_len - 1)) && suffix_len < 6) suffix_len++;
    }

    int zeros_left = 0;
    if (tc < max_coeff) {
        int z = (max_coeff == 4) ? vlc_get(b, &p264_vlc_tzdc[tc - 1]) : vlc_get(b, &p264_vlc_tz[tc - 1]);
        if (z < 0) return -1;
        zeros_left = z;
    }
    /* levels were read from the highest frequency down: place them (the block is cleared here, not by the caller: most
     * calls find an empty block and never get this far) */
    int pos = zeros_left + tc - 1;
    if (pos >= max_coeff) return -1;
    memset(out, 0, (size_t)(max_coeff == 4 ? 4 : 16) * sizeof *out);     /* (AC blocks: the unused sixteenth entry too - the block is stored whole) */
    for (int i = 0; i < tc; i++) {
        out[pos] = (int16_t)level[i];
        if (i == tc - 1) break;
        int run = 0;
        if (zeros_left > 0) {
            run = vlc_get(b, &p264_vlc_rb[(zeros_left > 7 ? 7 : zeros_left) - 1]);
            if (run < 0 || run > zeros_left) return -1;
        }
        zeros_left -= run;
        pos -= run + 1;
    }
    return tc;
}

#endif
