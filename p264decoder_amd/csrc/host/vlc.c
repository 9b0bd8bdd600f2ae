/* vlc.c - prefix-code trees and the CAVLC residual block reader (see vlc.h). */
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "vlc.h"
#include "cavlc_tables.h"

int vlc_build(vlc_t *v, int n, const uint8_t *len, const uint16_t *code, const int16_t *sym)
{
    /* pass 1: how many sub-table bits does each 8-bit root prefix need? */
    uint8_t need[1 << VLC_ROOT_BITS];
    memset(need, 0, sizeof need);
    for (int i = 0; i < n; i++) {
        if (len[i] <= VLC_ROOT_BITS) continue;
        int extra = len[i] - VLC_ROOT_BITS;
        int root = code[i] >> extra;
        if (extra > need[root]) need[root] = (uint8_t)extra;
    }
    int total = 1 << VLC_ROOT_BITS;
    int off[1 << VLC_ROOT_BITS];
    for (int r = 0; r < (1 << VLC_ROOT_BITS); r++) {
        off[r] = total;
        if (need[r]) total += 1 << need[r];
    }
    v->ent = (vlc_ent_t *)calloc((size_t)total, sizeof(vlc_ent_t));
    if (!v->ent) return -1;
    v->n_ent = total;
    for (int r = 0; r < (1 << VLC_ROOT_BITS); r++)
        if (need[r]) { v->ent[r].sym = (int16_t)off[r]; v->ent[r].sub = need[r]; }
    /* pass 2: replicate every code over the entries it prefixes */
    for (int i = 0; i < n; i++) {
        if (!len[i]) continue;
        if (len[i] <= VLC_ROOT_BITS) {
            int span = 1 << (VLC_ROOT_BITS - len[i]);
            int base = code[i] << (VLC_ROOT_BITS - len[i]);
            for (int k = 0; k < span; k++) {
                if (v->ent[base + k].len || v->ent[base + k].sub) return -1;   /* not prefix-free */
                v->ent[base + k].sym = sym[i]; v->ent[base + k].len = len[i];
            }
        } else {
            int extra = len[i] - VLC_ROOT_BITS;
            int root = code[i] >> extra;
            int sb = need[root];
            int span = 1 << (sb - extra);
            int base = off[root] + ((code[i] & ((1 << extra) - 1)) << (sb - extra));
            for (int k = 0; k < span; k++) {
                if (v->ent[base + k].len) return -1;
                v->ent[base + k].sym = sym[i]; v->ent[base + k].len = (uint8_t)extra;
            }
        }
    }
    return 0;
}

void vlc_free(vlc_t *v) { free(v->ent); v->ent = NULL; v->n_ent = 0; }

/* ------------------------------------------------------------------------------------- */
static vlc_t g_ct[3], g_ctdc, g_tz[15], g_tzdc[3], g_rb[7];
static int   g_init_rc = -1;
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static int build1(vlc_t *v, int n, const uint8_t *len, const uint8_t *code8)
{
    uint16_t code[68]; int16_t sym[68];
    for (int i = 0; i < n; i++) { code[i] = code8[i]; sym[i] = (int16_t)i; }
    return vlc_build(v, n, len, code, sym);
}

static void do_init(void)
{
    int rc = 0;
    for (int c = 0; c < 3; c++) {            /* symbol = (t1 << 5) | tc */
        uint8_t len[68]; uint16_t code[68]; int16_t sym[68]; int n = 0;
        for (int t1 = 0; t1 < 4; t1++)
            for (int tc = 0; tc < 17; tc++) {
                len[n] = ct_len[c][t1][tc]; code[n] = ct_code[c][t1][tc]; sym[n] = (int16_t)((t1 << 5) | tc); n++;
            }
        rc |= vlc_build(&g_ct[c], n, len, code, sym);
    }
    {
        uint8_t len[20]; uint16_t code[20]; int16_t sym[20]; int n = 0;
        for (int t1 = 0; t1 < 4; t1++)
            for (int tc = 0; tc < 5; tc++) {
                len[n] = ctdc_len[t1][tc]; code[n] = ctdc_code[t1][tc]; sym[n] = (int16_t)((t1 << 5) | tc); n++;
            }
        rc |= vlc_build(&g_ctdc, n, len, code, sym);
    }
    for (int i = 0; i < 15; i++) rc |= build1(&g_tz[i], 16 - i, tz_len[i], tz_code[i]);
    for (int i = 0; i < 3; i++)  rc |= build1(&g_tzdc[i], 4 - i, tzdc_len[i], tzdc_code[i]);
    for (int i = 0; i < 7; i++)  rc |= build1(&g_rb[i], i < 6 ? i + 2 : 15, rb_len[i], rb_code[i]);
    g_init_rc = rc;
}

int cavlc_global_init(void)
{
    pthread_once(&g_once, do_init);
    return g_init_rc;
}

int cavlc_read_block(bitrd_t *b, int nC, int max_coeff, int16_t *out)
{
    int tc, t1;
    if (nC < 0) {
        int s = vlc_get(b, &g_ctdc);
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
        int s = vlc_get(b, &g_ct[t]);
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
        int z = (max_coeff == 4) ? vlc_get(b, &g_tzdc[tc - 1]) : vlc_get(b, &g_tz[tc - 1]);
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
            run = vlc_get(b, &g_rb[(zeros_left > 7 ? 7 : zeros_left) - 1]);
            if (run < 0 || run > zeros_left) return -1;
        }
        zeros_left -= run;
        pos -= run + 1;
    }
    return tc;
}
