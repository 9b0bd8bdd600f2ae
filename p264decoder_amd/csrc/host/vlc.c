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
vlc_t p264_vlc_ct[3], p264_vlc_ctdc, p264_vlc_tz[15], p264_vlc_tzdc[3], p264_vlc_rb[7];
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
        rc |= vlc_build(&p264_vlc_ct[c], n, len, code, sym);
    }
    {
        uint8_t len[20]; uint16_t code[20]; int16_t sym[20]; int n = 0;
        for (int t1 = 0; t1 < 4; t1++)
            for (int tc = 0; tc < 5; tc++) {
                len[n] = ctdc_len[t1][tc]; code[n] = ctdc_code[t1][tc]; sym[n] = (int16_t)((t1 << 5) | tc); n++;
            }
        rc |= vlc_build(&p264_vlc_ctdc, n, len, code, sym);
    }
    for (int i = 0; i < 15; i++) rc |= build1(&p264_vlc_tz[i], 16 - i, tz_len[i], tz_code[i]);
    for (int i = 0; i < 3; i++)  rc |= build1(&p264_vlc_tzdc[i], 4 - i, tzdc_len[i], tzdc_code[i]);
    for (int i = 0; i < 7; i++)  rc |= build1(&p264_vlc_rb[i], i < 6 ? i + 2 : 15, rb_len[i], rb_code[i]);
    g_init_rc = rc;
}

int cavlc_global_init(void)
{
    pthread_once(&g_once, do_init);
    return g_init_rc;
}
