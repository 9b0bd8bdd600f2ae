/* oracle/cpu_recon.c - TEST INFRASTRUCTURE ONLY (see cpu_recon.h).
 *
 * CPU restatement of the reference's reconstruction path, one function per reference
 * function, each citing the file:line it follows.  Plain scalar C on purpose: this is the
 * checker, not the product.  Integer semantics that matter for bit-exactness (int16
 * storage wrap, SURVEY A-Q8; unshifted deblock offsets, A-Q3; DC fall-back keyed on the
 * top-left flag, A-Q7) are kept.
 */
#include <stdlib.h>
#include <string.h>
#include "cpu_recon.h"

/* Coverage counters for the test-suite (not part of the arithmetic):
 * [0] HV (centre) half-pel samples evaluated, [1] of those outside [-80,335] - the domain of the
 * reference's clip LUT (core/clip1.h:36-70), beyond which the reference reads past its table
 * (undefined; we clamp like the standard), [2] luma edge lines with bS>0, [3] of those modified,
 * [4] chroma edge lines with bS>0, [5] of those modified, [6] macroblock edges with bS>0 whose two macroblocks have
 * different QPs (the filter parameters come from the mean), [7] dequantised luma/chroma AC coefficients whose int16 store
 * wrapped (A-Q8). */
static long long g_stats[8], g_bipred_blocks, g_bs_by_picture;
void oracle_stats_reset(void) { memset(g_stats, 0, sizeof g_stats); g_bipred_blocks = 0; g_bs_by_picture = 0; }
/* B pictures: edge segments whose strength by reference PICTURES (H.264 8.7.2.1) differs from the list-by-list comparison of
 * reference INDICES the reference's encoder-side loop makes (core/frame.c:565-577) */
long long oracle_bs_by_picture(void) { return g_bs_by_picture; }
long long oracle_bipred_blocks(void) { return g_bipred_blocks; }
void oracle_stats_get(long long out[8]) { memcpy(out, g_stats, sizeof g_stats); }

static inline int clip3(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
static inline int clip255(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }
static inline int iabs(int v) { return v < 0 ? -v : v; }

/* ---- tables ------------------------------------------------------------------------- */
/* core/set.c:27-35 x flat-16 scaling list (decoder/set.c:261-263, core/set.c:98) */
static const int dq_scale[6][3] = { {10,13,16}, {11,14,18}, {13,16,20}, {14,18,23}, {16,20,25}, {18,23,29} };
static inline int dq_mf(int qp, int pos) { return 16 * dq_scale[qp % 6][(pos & 1) + ((pos >> 2) & 1)]; }

/* core/macroblock.h:210-218 */
static const uint8_t chroma_qp[52] = {
     0, 1, 2, 3, 4, 5, 6, 7, 8, 9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,
    29,30,31,32,32,33,34,34,35,35,36,36,37,37,37,38,38,38,39,39,39,39 };

/* core/frame.c:262-291 */
static const uint8_t alpha_tab[52] = {
    0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,4,4,5,6,7,8,9,10,12,13,15,17,20,22,
    25,28,32,36,40,45,50,56,63,71,80,90,101,113,127,144,162,182,203,226,255,255 };
static const uint8_t beta_tab[52] = {
    0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,2,2,2,3,3,3,3,4,4,4,6,6,7,7,
    8,8,9,9,10,10,11,11,12,12,13,13,14,14,15,15,16,16,17,17,18,18 };
static const uint8_t tc0_tab[52][3] = {
    {0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},
    {0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,1},{0,0,1},{0,0,1},{0,0,1},{0,1,1},{0,1,1},{1,1,1},
    {1,1,1},{1,1,1},{1,1,1},{1,1,2},{1,1,2},{1,1,2},{1,1,2},{1,2,3},{1,2,3},{2,2,3},{2,2,4},{2,3,4},
    {2,3,4},{3,3,5},{3,4,6},{3,4,6},{4,5,7},{4,5,8},{4,6,9},{5,7,10},{6,8,11},{6,8,13},{7,10,14},{8,11,16},
    {9,12,18},{10,13,20},{11,15,23},{13,17,25} };

/* decoder/macroblock.c:602-603 as raster positions; core/macroblock.h:194-201 */
static const uint8_t zigzag[16] = { 0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15 };
static const uint8_t bx_of[16] = { 0,1,0,1, 2,3,2,3, 0,1,0,1, 2,3,2,3 };
static const uint8_t by_of[16] = { 0,0,1,1, 0,0,1,1, 2,2,3,3, 2,2,3,3 };
static const uint8_t blk_at[4][4] = { {0,1,4,5}, {2,3,6,7}, {8,9,12,13}, {10,11,14,15} };   /* [y][x] */

/* ---- dequant / transforms ----------------------------------------------------------- */
void oracle_dequant4x4(int16_t d[16], int qp)
{   /* core/quant.c:66-99 */
    int qbits = qp / 6 - 4;
    for (int i = 0; i < 16; i++) {
        int v = d[i] * dq_mf(qp, i);
        if (qbits >= 0 && (long long)v * (1 << qbits) != (int16_t)((unsigned)v << qbits)) g_stats[7]++;   /* the int16 store wraps (A-Q8) */
        if (qbits >= 0) d[i] = (int16_t)((unsigned)v << qbits);
        else            d[i] = (int16_t)((v + (1 << (-qbits - 1))) >> (-qbits));
    }
}

void oracle_idct4x4dc(int16_t d[16])
{   /* core/dct.c:104-136: columns into an int16 tmp, then rows */
    int16_t t[16];
    for (int i = 0; i < 4; i++) {
        int s01 = d[0*4+i] + d[1*4+i], d01 = d[0*4+i] - d[1*4+i];
        int s23 = d[2*4+i] + d[3*4+i], d23 = d[2*4+i] - d[3*4+i];
        t[0*4+i] = (int16_t)(s01 + s23); t[1*4+i] = (int16_t)(s01 - s23);
        t[2*4+i] = (int16_t)(d01 - d23); t[3*4+i] = (int16_t)(d01 + d23);
    }
    for (int i = 0; i < 4; i++) {
        int s01 = t[i*4+0] + t[i*4+1], d01 = t[i*4+0] - t[i*4+1];
        int s23 = t[i*4+2] + t[i*4+3], d23 = t[i*4+2] - t[i*4+3];
        d[i*4+0] = (int16_t)(s01 + s23); d[i*4+1] = (int16_t)(s01 - s23);
        d[i*4+2] = (int16_t)(d01 - d23); d[i*4+3] = (int16_t)(d01 + d23);
    }
}

void oracle_dequant4x4_dc(int16_t d[16], int qp)
{   /* core/quant.c:161-191: rounded when the shift is to the right */
    int qbits = qp / 6 - 6, mf = dq_mf(qp, 0);
    for (int i = 0; i < 16; i++) {
        if (qbits >= 0) d[i] = (int16_t)(d[i] * (int)((unsigned)mf << qbits));
        else            d[i] = (int16_t)((d[i] * mf + (1 << (-qbits - 1))) >> (-qbits));
    }
}

void oracle_idct2x2dc(int16_t d[4])
{   /* core/dct.c:55-68 (dct2x2dc doubles as its own inverse, :401-402) */
    int t0 = d[0] + d[1], t1 = d[0] - d[1], t2 = d[2] + d[3], t3 = d[2] - d[3];
    d[0] = (int16_t)(t0 + t2); d[1] = (int16_t)(t1 + t3);
    d[2] = (int16_t)(t0 - t2); d[3] = (int16_t)(t1 - t3);
}

void oracle_dequant2x2_dc(int16_t d[4], int qp)
{   /* core/quant.c:138-159: truncated, not rounded */
    int qbits = qp / 6 - 5, mf = dq_mf(qp, 0);
    for (int i = 0; i < 4; i++) {
        if (qbits >= 0) d[i] = (int16_t)(d[i] * (int)((unsigned)mf << qbits));
        else            d[i] = (int16_t)((d[i] * mf) >> (-qbits));
    }
}

void oracle_add4x4_idct(uint8_t *dst, int stride, const int16_t c[16])
{   /* core/dct.c:205-247: rows, then columns, int16 intermediates */
    int16_t t[16], r[16];
    for (int i = 0; i < 4; i++) {
        int s02 = c[i*4+0] + c[i*4+2], d02 = c[i*4+0] - c[i*4+2];
        int s13 = c[i*4+1] + (c[i*4+3] >> 1), d13 = (c[i*4+1] >> 1) - c[i*4+3];
        t[i*4+0] = (int16_t)(s02 + s13); t[i*4+1] = (int16_t)(d02 + d13);
        t[i*4+2] = (int16_t)(d02 - d13); t[i*4+3] = (int16_t)(s02 - s13);
    }
    for (int i = 0; i < 4; i++) {
        int s02 = t[0*4+i] + t[2*4+i], d02 = t[0*4+i] - t[2*4+i];
        int s13 = t[1*4+i] + (t[3*4+i] >> 1), d13 = (t[1*4+i] >> 1) - t[3*4+i];
        r[0*4+i] = (int16_t)((s02 + s13 + 32) >> 6); r[1*4+i] = (int16_t)((d02 + d13 + 32) >> 6);
        r[2*4+i] = (int16_t)((d02 - d13 + 32) >> 6); r[3*4+i] = (int16_t)((s02 - s13 + 32) >> 6);
    }
    for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++)
            dst[y*stride + x] = (uint8_t)clip255(dst[y*stride + x] + r[y*4 + x]);
}

/* ---- intra prediction from explicit neighbour arrays -------------------------------- */
static void fill(uint8_t *o, int n, int w, int v) { for (int y = 0; y < n; y++) memset(o + y*w, v, (size_t)n); }

/* core/predict.c:55-193; out is 16x16 with stride 16 */
static void pred16(int mode, const uint8_t *l, const uint8_t *t, int tl, uint8_t *o)
{
    int s = 0;
    switch (mode) {
    case 0: for (int y = 0; y < 16; y++) memcpy(o + y*16, t, 16); break;
    case 1: for (int y = 0; y < 16; y++) memset(o + y*16, l[y], 16); break;
    case 2: for (int i = 0; i < 16; i++) s += l[i] + t[i]; fill(o, 16, 16, (s + 16) >> 5); break;
    case 4: for (int i = 0; i < 16; i++) s += l[i]; fill(o, 16, 16, (s + 8) >> 4); break;
    case 5: for (int i = 0; i < 16; i++) s += t[i]; fill(o, 16, 16, (s + 8) >> 4); break;
    case 6: fill(o, 16, 16, 128); break;
    case 3: {
        int H = 0, V = 0;
        for (int i = 0; i <= 7; i++) {
            H += (i + 1) * (t[8 + i] - (6 - i >= 0 ? t[6 - i] : tl));
            V += (i + 1) * (l[8 + i] - (6 - i >= 0 ? l[6 - i] : tl));
        }
        int a = 16 * (l[15] + t[15]), b = (5 * H + 32) >> 6, c = (5 * V + 32) >> 6;
        int i00 = a - 7*b - 7*c + 16;
        for (int y = 0; y < 16; y++, i00 += c)
            for (int x = 0; x < 16; x++) o[y*16 + x] = (uint8_t)clip255((i00 + b*x) >> 5);
        break; }
    }
}

/* core/predict.c:199-361; out is 8x8 with stride 8 */
static void pred8c(int mode, const uint8_t *l, const uint8_t *t, int tl, uint8_t *o)
{
    int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int i = 0; i < 4; i++) { s0 += t[i]; s1 += t[4+i]; s2 += l[i]; s3 += l[4+i]; }
    int dc[4];
    switch (mode) {
    case 0: dc[0] = (s0 + s2 + 4) >> 3; dc[1] = (s1 + 2) >> 2; dc[2] = (s3 + 2) >> 2; dc[3] = (s1 + s3 + 4) >> 3; break;
    case 4: dc[0] = dc[1] = (s2 + 2) >> 2; dc[2] = dc[3] = (s3 + 2) >> 2; break;
    case 5: dc[0] = dc[2] = (s0 + 2) >> 2; dc[1] = dc[3] = (s1 + 2) >> 2; break;
    case 6: dc[0] = dc[1] = dc[2] = dc[3] = 128; break;
    case 1: for (int y = 0; y < 8; y++) memset(o + y*8, l[y], 8); return;
    case 2: for (int y = 0; y < 8; y++) memcpy(o + y*8, t, 8); return;
    case 3: {
        int H = 0, V = 0;
        for (int i = 0; i < 4; i++) {
            H += (i + 1) * (t[4 + i] - (2 - i >= 0 ? t[2 - i] : tl));
            V += (i + 1) * (l[4 + i] - (2 - i >= 0 ? l[2 - i] : tl));
        }
        int a = 16 * (l[7] + t[7]), b = (17 * H + 16) >> 5, c = (17 * V + 16) >> 5;
        int i00 = a - 3*b - 3*c + 16;
        for (int y = 0; y < 8; y++, i00 += c)
            for (int x = 0; x < 8; x++) o[y*8 + x] = (uint8_t)clip255((i00 + b*x) >> 5);
        return; }
    default: return;
    }
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) o[y*8 + x] = (uint8_t)dc[(y >> 2) * 2 + (x >> 2)];
}

/* core/predict.c:366-638; t has 8 entries (top + top-right); out is 4x4 with stride 4 */
static void pred4(int mode, const uint8_t *l, const uint8_t *t, int lt, uint8_t *o)
{
#define O(x, y) o[(y)*4 + (x)]
#define F3(a, b, c) (((a) + 2*(b) + (c) + 2) >> 2)
#define F2(a, b) (((a) + (b) + 1) >> 1)
    switch (mode) {
    case 0: for (int y = 0; y < 4; y++) memcpy(o + y*4, t, 4); break;
    case 1: for (int y = 0; y < 4; y++) memset(o + y*4, l[y], 4); break;
    case 2: fill(o, 4, 4, (l[0]+l[1]+l[2]+l[3]+t[0]+t[1]+t[2]+t[3]+4) >> 3); break;
    case 9: fill(o, 4, 4, (l[0]+l[1]+l[2]+l[3]+2) >> 2); break;
    case 10: fill(o, 4, 4, (t[0]+t[1]+t[2]+t[3]+2) >> 2); break;
    case 11: fill(o, 4, 4, 128); break;
    case 3:  /* diagonal down-left */
        for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++) {
                int k = x + y;
                O(x, y) = (uint8_t)(k == 6 ? (t[6] + 3*t[7] + 2) >> 2 : F3(t[k], t[k+1], t[k+2]));
            }
        break;
    case 4: { /* diagonal down-right: edge e[] = l3 l2 l1 l0 lt t0 t1 t2 t3 */
        int e[9] = { l[3], l[2], l[1], l[0], lt, t[0], t[1], t[2], t[3] };
        for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++) { int k = 4 + x - y; O(x, y) = (uint8_t)F3(e[k-1], e[k], e[k+1]); }
        break; }
    case 5:  /* vertical-right */
        O(0,0) = O(1,2) = (uint8_t)F2(lt, t[0]);   O(1,0) = O(2,2) = (uint8_t)F2(t[0], t[1]);
        O(2,0) = O(3,2) = (uint8_t)F2(t[1], t[2]); O(3,0) = (uint8_t)F2(t[2], t[3]);
        O(0,1) = O(1,3) = (uint8_t)F3(l[0], lt, t[0]);   O(1,1) = O(2,3) = (uint8_t)F3(lt, t[0], t[1]);
        O(2,1) = O(3,3) = (uint8_t)F3(t[0], t[1], t[2]); O(3,1) = (uint8_t)F3(t[1], t[2], t[3]);
        O(0,2) = (uint8_t)F3(lt, l[0], l[1]); O(0,3) = (uint8_t)F3(l[0], l[1], l[2]);
        break;
    case 6:  /* horizontal-down */
        O(0,0) = O(2,1) = (uint8_t)F2(lt, l[0]);   O(1,0) = O(3,1) = (uint8_t)F3(l[0], lt, t[0]);
        O(2,0) = (uint8_t)F3(lt, t[0], t[1]);      O(3,0) = (uint8_t)F3(t[0], t[1], t[2]);
        O(0,1) = O(2,2) = (uint8_t)F2(l[0], l[1]); O(1,1) = O(3,2) = (uint8_t)F3(lt, l[0], l[1]);
        O(0,2) = O(2,3) = (uint8_t)F2(l[1], l[2]); O(1,2) = O(3,3) = (uint8_t)F3(l[0], l[1], l[2]);
        O(0,3) = (uint8_t)F2(l[2], l[3]);          O(1,3) = (uint8_t)F3(l[1], l[2], l[3]);
        break;
    case 7:  /* vertical-left */
        for (int x = 0; x < 4; x++) {
            O(x,0) = (uint8_t)F2(t[x], t[x+1]);     O(x,2) = (uint8_t)F2(t[x+1], t[x+2]);
            O(x,1) = (uint8_t)F3(t[x], t[x+1], t[x+2]); O(x,3) = (uint8_t)F3(t[x+1], t[x+2], t[x+3]);
        }
        break;
    case 8:  /* horizontal-up */
        O(0,0) = (uint8_t)F2(l[0], l[1]);          O(1,0) = (uint8_t)F3(l[0], l[1], l[2]);
        O(2,0) = O(0,1) = (uint8_t)F2(l[1], l[2]); O(3,0) = O(1,1) = (uint8_t)F3(l[1], l[2], l[3]);
        O(2,1) = O(0,2) = (uint8_t)F2(l[2], l[3]); O(3,1) = O(1,2) = (uint8_t)F3(l[2], l[3], l[3]);
        O(2,2) = O(3,2) = O(0,3) = O(1,3) = O(2,3) = O(3,3) = l[3];
        break;
    }
#undef O
#undef F3
#undef F2
}

/* KAT wrappers: neighbours come from the picture like in the reference */
void oracle_pred16x16(uint8_t *dst, int stride, int mode)
{
    uint8_t l[16], t[16], o[256];
    for (int i = 0; i < 16; i++) { l[i] = dst[i*stride - 1]; t[i] = dst[i - stride]; }
    pred16(mode, l, t, dst[-stride - 1], o);
    for (int y = 0; y < 16; y++) memcpy(dst + y*stride, o + y*16, 16);
}
void oracle_pred8x8c(uint8_t *dst, int stride, int mode)
{
    uint8_t l[8], t[8], o[64];
    for (int i = 0; i < 8; i++) { l[i] = dst[i*stride - 1]; t[i] = dst[i - stride]; }
    pred8c(mode, l, t, dst[-stride - 1], o);
    for (int y = 0; y < 8; y++) memcpy(dst + y*stride, o + y*8, 8);
}
void oracle_pred4x4(uint8_t *dst, int stride, int mode)
{
    uint8_t l[4], t[8], o[16];
    for (int i = 0; i < 4; i++) l[i] = dst[i*stride - 1];
    for (int i = 0; i < 8; i++) t[i] = dst[i - stride];
    pred4(mode, l, t, dst[-stride - 1], o);
    for (int y = 0; y < 4; y++) memcpy(dst + y*stride, o + y*4, 4);
}

/* ---- motion compensation ------------------------------------------------------------ */
typedef struct { const uint8_t *p; int w, h; } plane_t;
static inline int px(const plane_t *f, int x, int y)
{   /* clamped read == reading the reference's replicated pads (core/frame.c:183-222; A-Q9) */
    return f->p[clip3(y, 0, f->h - 1) * f->w + clip3(x, 0, f->w - 1)];
}
static inline int tap_h(const plane_t *f, int x, int y)
{   /* core/mc.c:53-56 */
    return px(f,x-2,y) - 5*px(f,x-1,y) + 20*(px(f,x,y) + px(f,x+1,y)) - 5*px(f,x+2,y) + px(f,x+3,y);
}
static inline int tap_v(const plane_t *f, int x, int y)
{   /* core/mc.c:49-52 */
    return px(f,x,y-2) - 5*px(f,x,y-1) + 20*(px(f,x,y) + px(f,x,y+1)) - 5*px(f,x,y+2) + px(f,x,y+3);
}
/* the four planes of p264_frame_filter (core/mc.c:409-451): 0 = integer, 1 = mc_hh, 2 = mc_hv, 3 = mc_hc */
static int half_plane(const plane_t *f, int which, int x, int y)
{
    switch (which) {
    case 0: return px(f, x, y);
    case 1: return clip255((tap_h(f, x, y) + 16) >> 5);                     /* core/mc.c:180 */
    case 2: return clip255((tap_v(f, x, y) + 16) >> 5);                     /* core/mc.c:194 */
    default: {                                                                /* core/mc.c:213-223 */
        int t0 = tap_h(f,x,y-2), t1 = tap_h(f,x,y-1), t2 = tap_h(f,x,y), t3 = tap_h(f,x,y+1), t4 = tap_h(f,x,y+2), t5 = tap_h(f,x,y+3);
        int v = (t0 - 5*t1 + 20*t2 + 20*t3 - 5*t4 + t5 + 512) >> 10;
        g_stats[0]++; if (v < -80 || v > 335) g_stats[1]++;
        return clip255(v); }
    }
}

void oracle_mc_luma(const uint8_t *ref, int w, int h, int x0, int y0, int mvx, int mvy,
                    int bw, int bh, uint8_t *dst, int dst_stride)
{   /* core/mc.c:237-266 */
    plane_t f = { ref, w, h };
    int correction = (mvx & 1) && (mvy & 1) && ((mvx & 2) ^ (mvy & 2));
    int h1x = mvx >> 1, h1y = (mvy + 1 - correction) >> 1;
    int f1 = (h1x & 1) + ((h1y & 1) << 1);
    int qpel = (mvx | mvy) & 1;
    int h2x = (mvx + 1) >> 1, h2y = (mvy + correction) >> 1;
    int f2 = (h2x & 1) + ((h2y & 1) << 1);
    for (int y = 0; y < bh; y++)
        for (int x = 0; x < bw; x++) {
            int a = half_plane(&f, f1, x0 + x + (h1x >> 1), y0 + y + (h1y >> 1));
            if (qpel) {
                int b = half_plane(&f, f2, x0 + x + (h2x >> 1), y0 + y + (h2y >> 1));
                a = (a + b + 1) >> 1;                                       /* core/mc.c:58-74 */
            }
            dst[y*dst_stride + x] = (uint8_t)a;
        }
}

void oracle_mc_chroma(const uint8_t *ref, int w, int h, int x0, int y0, int mvx, int mvy,
                      int bw, int bh, uint8_t *dst, int dst_stride)
{   /* core/mc.c:303-334 */
    plane_t f = { ref, w, h };
    int dx = mvx & 7, dy = mvy & 7;
    int cA = (8-dx)*(8-dy), cB = dx*(8-dy), cC = (8-dx)*dy, cD = dx*dy;
    int ox = x0 + (mvx >> 3), oy = y0 + (mvy >> 3);
    for (int y = 0; y < bh; y++)
        for (int x = 0; x < bw; x++)
            dst[y*dst_stride + x] = (uint8_t)((cA*px(&f,ox+x,oy+y) + cB*px(&f,ox+x+1,oy+y) +
                                               cC*px(&f,ox+x,oy+y+1) + cD*px(&f,ox+x+1,oy+y+1) + 32) >> 6);
}

/* ---- bi-prediction: pixel_avg_wxh (core/mc.c:76-88), pixel_avg_weight_wxh (core/mc.c:106-132) ------------------------ */
void oracle_bipred_avg(uint8_t *dst, int ds, const uint8_t *src, int ss, int w, int h)
{
    for (int y = 0; y < h; y++, dst += ds, src += ss)
        for (int x = 0; x < w; x++) dst[x] = (uint8_t)((dst[x] + src[x] + 1) >> 1);
}
void oracle_bipred_weight(uint8_t *dst, int ds, const uint8_t *src, int ss, int w, int h, int w1)
{   /* implicit weights only: log2_denom 5, offset 0, w1 + w2 = 64 (:104-105) */
    const int w2 = 64 - w1;
    for (int y = 0; y < h; y++, dst += ds, src += ss)
        for (int x = 0; x < w; x++) dst[x] = (uint8_t)clip255((dst[x] * w1 + src[x] * w2 + 32) >> 6);
}

/* ---- deblocking sample filters ------------------------------------------------------- */
void oracle_deblock_luma(uint8_t *pix, int xs, int ys, int alpha, int beta, const int8_t tc0[4])
{   /* core/frame.c:302-341 */
    for (int i = 0; i < 4; i++) {
        if (tc0[i] < 0) { pix += 4*ys; continue; }
        for (int d = 0; d < 4; d++, pix += ys) {
            int p2 = pix[-3*xs], p1 = pix[-2*xs], p0 = pix[-xs], q0 = pix[0], q1 = pix[xs], q2 = pix[2*xs];
            if (iabs(p0 - q0) < alpha && iabs(p1 - p0) < beta && iabs(q1 - q0) < beta) {
                int tc = tc0[i];
                if (iabs(p2 - p0) < beta) { pix[-2*xs] = (uint8_t)(p1 + clip3(((p2 + ((p0 + q0 + 1) >> 1)) >> 1) - p1, -tc0[i], tc0[i])); tc++; }
                if (iabs(q2 - q0) < beta) { pix[xs]    = (uint8_t)(q1 + clip3(((q2 + ((p0 + q0 + 1) >> 1)) >> 1) - q1, -tc0[i], tc0[i])); tc++; }
                int delta = clip3((((q0 - p0) * 4) + (p1 - q1) + 4) >> 3, -tc, tc);
                pix[-xs] = (uint8_t)clip255(p0 + delta);
                pix[0]   = (uint8_t)clip255(q0 - delta);
            }
        }
    }
}

void oracle_deblock_chroma(uint8_t *pix, int xs, int ys, int alpha, int beta, const int8_t tcv[4])
{   /* core/frame.c:351-377 */
    for (int i = 0; i < 4; i++) {
        int tc = tcv[i];
        if (tc <= 0) { pix += 2*ys; continue; }
        for (int d = 0; d < 2; d++, pix += ys) {
            int p1 = pix[-2*xs], p0 = pix[-xs], q0 = pix[0], q1 = pix[xs];
            if (iabs(p0 - q0) < alpha && iabs(p1 - p0) < beta && iabs(q1 - q0) < beta) {
                int delta = clip3((((q0 - p0) * 4) + (p1 - q1) + 4) >> 3, -tc, tc);
                pix[-xs] = (uint8_t)clip255(p0 + delta);
                pix[0]   = (uint8_t)clip255(q0 - delta);
            }
        }
    }
}

void oracle_deblock_luma_intra(uint8_t *pix, int xs, int ys, int alpha, int beta)
{   /* core/frame.c:387-433 */
    for (int d = 0; d < 16; d++, pix += ys) {
        int p2 = pix[-3*xs], p1 = pix[-2*xs], p0 = pix[-xs], q0 = pix[0], q1 = pix[xs], q2 = pix[2*xs];
        if (!(iabs(p0 - q0) < alpha && iabs(p1 - p0) < beta && iabs(q1 - q0) < beta)) continue;
        if (iabs(p0 - q0) < ((alpha >> 2) + 2)) {
            if (iabs(p2 - p0) < beta) {
                int p3 = pix[-4*xs];
                pix[-xs]   = (uint8_t)((p2 + 2*p1 + 2*p0 + 2*q0 + q1 + 4) >> 3);
                pix[-2*xs] = (uint8_t)((p2 + p1 + p0 + q0 + 2) >> 2);
                pix[-3*xs] = (uint8_t)((2*p3 + 3*p2 + p1 + p0 + q0 + 4) >> 3);
            } else pix[-xs] = (uint8_t)((2*p1 + p0 + q1 + 2) >> 2);
            if (iabs(q2 - q0) < beta) {
                int q3 = pix[3*xs];
                pix[0]    = (uint8_t)((p1 + 2*p0 + 2*q0 + 2*q1 + q2 + 4) >> 3);
                pix[xs]   = (uint8_t)((p0 + q0 + q1 + q2 + 2) >> 2);
                pix[2*xs] = (uint8_t)((2*q3 + 3*q2 + q1 + q0 + p0 + 4) >> 3);
            } else pix[0] = (uint8_t)((2*q1 + q0 + p1 + 2) >> 2);
        } else {
            pix[-xs] = (uint8_t)((2*p1 + p0 + q1 + 2) >> 2);
            pix[0]   = (uint8_t)((2*q1 + q0 + p1 + 2) >> 2);
        }
    }
}

void oracle_deblock_chroma_intra(uint8_t *pix, int xs, int ys, int alpha, int beta)
{   /* core/frame.c:443-462 */
    for (int d = 0; d < 8; d++, pix += ys) {
        int p1 = pix[-2*xs], p0 = pix[-xs], q0 = pix[0], q1 = pix[xs];
        if (iabs(p0 - q0) < alpha && iabs(p1 - p0) < beta && iabs(q1 - q0) < beta) {
            pix[-xs] = (uint8_t)((2*p1 + p0 + q1 + 2) >> 2);
            pix[0]   = (uint8_t)((2*q1 + q0 + p1 + 2) >> 2);
        }
    }
}

/* ---- per-macroblock reconstruction (decoder/macroblock.c:755-934) ---------------------- */
typedef struct {
    const p264hip_picture_t *pic;
    uint8_t *y, *u, *v;          /* destination frame */
    int w, h, cw, ch;            /* luma / chroma plane sizes */
    uint8_t **planes;
} rc_t;

static const int16_t *coef_block(const p264hip_picture_t *pic, const p264hip_mb_t *m, uint32_t bit)
{   /* n-th packed block of this MB: order [luma DC][chroma DC][0..23] (include/p264hip.h) */
    uint32_t before = 0, mask = m->coef_mask;
    if (bit == P264_COEF_LUMA_DC) before = 0;
    else if (bit == P264_COEF_CHROMA_DC) before = (mask & P264_COEF_LUMA_DC) ? 1 : 0;
    else {
        before = ((mask & P264_COEF_LUMA_DC) ? 1 : 0) + ((mask & P264_COEF_CHROMA_DC) ? 1 : 0);
        before += (uint32_t)__builtin_popcount(mask & (bit - 1) & 0xffffff);
    }
    return pic->coefs + ((size_t)m->coef_index + before) * 16;
}

/* decoder/macroblock.c:605-622: scan order -> raster */
static void unscan_full(int16_t d[16], const int16_t *lv) { for (int i = 0; i < 16; i++) d[zigzag[i]] = lv[i]; }
static void unscan_ac(int16_t d[16], const int16_t *lv)   { d[0] = 0; for (int i = 1; i < 16; i++) d[zigzag[i]] = lv[i-1]; }

static void residual_luma_4x4(const rc_t *r, const p264hip_mb_t *m, int mbx, int mby, int blk)
{   /* decoder/macroblock.c:820-829 / :839-847 */
    int16_t d[16];
    unscan_full(d, coef_block(r->pic, m, 1u << blk));
    oracle_dequant4x4(d, m->qp);
    oracle_add4x4_idct(r->y + (mby*16 + by_of[blk]*4) * r->w + mbx*16 + bx_of[blk]*4, r->w, d);
}

static void recon_chroma_residual(const rc_t *r, const p264hip_mb_t *m, int mbx, int mby)
{   /* decoder/macroblock.c:851-890 */
    if (!(m->cbp >> 4)) return;
    int qpc = chroma_qp[clip3(m->qp + r->pic->chroma_qp_offset, 0, 51)];
    for (int ch = 0; ch < 2; ch++) {
        int16_t dc[4] = { 0, 0, 0, 0 };
        if (m->coef_mask & P264_COEF_CHROMA_DC) memcpy(dc, coef_block(r->pic, m, P264_COEF_CHROMA_DC) + ch*4, 8);
        oracle_idct2x2dc(dc);
        oracle_dequant2x2_dc(dc, qpc);
        uint8_t *plane = ch ? r->v : r->u;
        for (int i = 0; i < 4; i++) {
            int16_t d[16];
            int blk = 16 + ch*4 + i;
            if (m->coef_mask & (1u << blk)) { unscan_ac(d, coef_block(r->pic, m, 1u << blk)); oracle_dequant4x4(d, qpc); }
            else memset(d, 0, sizeof d);
            d[0] = dc[i];                                                    /* :886, raster 2x2 */
            oracle_add4x4_idct(plane + (mby*8 + (i >> 1)*4) * r->cw + mbx*8 + (i & 1)*4, r->cw, d);
        }
    }
}

static void recon_inter(const rc_t *r, const p264hip_mb_t *m, int mbx, int mby, int mbi)
{   /* p264_mb_mc: core/macroblock.c:506-524,633-676.  Done per 4x4 block: each sample depends only
       on its own motion vector, so this equals the reference's per-partition calls.
       B pictures: p264_mb_mc_1xywh / p264_mb_mc_01xywh (core/macroblock.c:525-583): a quadrant predicts from list X iff its
       list-X index is >= 0; with both, the list-0 prediction lands in the picture, the list-1 prediction in a temporary,
       and pf->avg / pf->avg_weight combine them (weight = bipred_weight[ref0][ref1], applied to the list-0 samples). */
    const p264hip_picture_t *pic = r->pic;
    const int isB = pic->slice_type == P264_SLICE_B;
    for (int b = 0; b < 16; b++) {
        int bx = b & 3, by = b >> 2, q = (by >> 1)*2 + (bx >> 1);
        int r0 = pic->ref_idx[mbi*4 + q], r1 = isB ? pic->ref_idx_l1[mbi*4 + q] : -1;
        int X = mbx*16 + bx*4, Y = mby*16 + by*4;
        uint8_t *dy = r->y + Y*r->w + X, *du = r->u + (Y/2)*r->cw + X/2, *dv = r->v + (Y/2)*r->cw + X/2;
        if (!isB || r0 >= 0) {
            int ri = r0;
            if (ri < 0 || ri >= pic->n_ref) ri = isB ? clip3(ri, 0, pic->n_ref - 1) : 0;
            int slot = pic->ref_slot[ri];
            int mvx = pic->mv[(mbi*16 + b)*2], mvy = pic->mv[(mbi*16 + b)*2 + 1];
            oracle_mc_luma(r->planes[slot*3], r->w, r->h, X, Y, mvx, mvy, 4, 4, dy, r->w);
            oracle_mc_chroma(r->planes[slot*3+1], r->cw, r->ch, X/2, Y/2, mvx, mvy, 2, 2, du, r->cw);
            oracle_mc_chroma(r->planes[slot*3+2], r->cw, r->ch, X/2, Y/2, mvx, mvy, 2, 2, dv, r->cw);
        }
        if (isB && r1 >= 0) {
            int ri = clip3(r1, 0, pic->n_ref_l1 - 1), slot = pic->ref_slot_l1[ri];
            int mvx = pic->mv_l1[(mbi*16 + b)*2], mvy = pic->mv_l1[(mbi*16 + b)*2 + 1];
            if (r0 < 0) {                                                    /* list 1 only: straight into the picture */
                oracle_mc_luma(r->planes[slot*3], r->w, r->h, X, Y, mvx, mvy, 4, 4, dy, r->w);
                oracle_mc_chroma(r->planes[slot*3+1], r->cw, r->ch, X/2, Y/2, mvx, mvy, 2, 2, du, r->cw);
                oracle_mc_chroma(r->planes[slot*3+2], r->cw, r->ch, X/2, Y/2, mvx, mvy, 2, 2, dv, r->cw);
            } else {
                uint8_t ty[16], tu[4], tv[4];
                int w1 = pic->bipred_weight[clip3(r0, 0, pic->n_ref - 1) * P264HIP_MAX_REFS + ri];
                oracle_mc_luma(r->planes[slot*3], r->w, r->h, X, Y, mvx, mvy, 4, 4, ty, 4);
                oracle_mc_chroma(r->planes[slot*3+1], r->cw, r->ch, X/2, Y/2, mvx, mvy, 2, 2, tu, 2);
                oracle_mc_chroma(r->planes[slot*3+2], r->cw, r->ch, X/2, Y/2, mvx, mvy, 2, 2, tv, 2);
                if (pic->weighted_bipred) {
                    oracle_bipred_weight(dy, r->w, ty, 4, 4, 4, w1);
                    oracle_bipred_weight(du, r->cw, tu, 2, 2, 2, w1);
                    oracle_bipred_weight(dv, r->cw, tv, 2, 2, 2, w1);
                } else {
                    oracle_bipred_avg(dy, r->w, ty, 4, 4, 4);
                    oracle_bipred_avg(du, r->cw, tu, 2, 2, 2);
                    oracle_bipred_avg(dv, r->cw, tv, 2, 2, 2);
                }
                g_bipred_blocks++;                                           /* a bi-predicted 4x4 block */
            }
        }
    }
    for (int i = 0; i < 16; i++)
        if (m->coef_mask & (1u << i)) residual_luma_4x4(r, m, mbx, mby, i);
    recon_chroma_residual(r, m, mbx, mby);
}

/* neighbour samples of a WxW block at (X,Y) in a plane, with the reference's substitutions
   (decoder/macroblock.c:697-713): missing left/top/top-left -> 128, missing top-right -> t[W-1] */
static void gather(const uint8_t *p, int stride, int X, int Y, int W, int nt,
                   int left, int top, int topright, int topleft, uint8_t *l, uint8_t *t, int *tl)
{
    for (int i = 0; i < W; i++) l[i] = left ? p[(Y + i)*stride + X - 1] : 128;
    for (int i = 0; i < W; i++) t[i] = top ? p[(Y - 1)*stride + X + i] : 128;
    for (int i = W; i < nt; i++) t[i] = topright ? p[(Y - 1)*stride + X + i] : t[W-1];
    *tl = topleft ? p[(Y - 1)*stride + X - 1] : 128;
}

static void recon_intra(const rc_t *r, const p264hip_mb_t *m, int mbx, int mby, int mbi)
{
    int L = m->avail & P264_AVAIL_LEFT, T = m->avail & P264_AVAIL_TOP;
    int TR = m->avail & P264_AVAIL_TOPRIGHT, TL = m->avail & P264_AVAIL_TOPLEFT;
    uint8_t l[16], t[16], o[256]; int tl;
    uint8_t *Y = r->y + mby*16*r->w + mbx*16;
    if (m->mb_type == P264_MB_I16x16) {
        /* decoder/macroblock.c:771-798 */
        int mode = m->intra_modes & 3;
        if (mode == 2) mode = TL ? 2 : L ? 4 : T ? 5 : 6;                   /* valid_intra16x16_mode :635-667 */
        gather(r->y, r->w, mbx*16, mby*16, 16, 16, L, T, 0, TL, l, t, &tl);
        pred16(mode, l, t, tl, o);
        for (int y = 0; y < 16; y++) memcpy(Y + y*r->w, o + y*16, 16);
        int16_t dc[16];
        memset(dc, 0, sizeof dc);
        if (m->coef_mask & P264_COEF_LUMA_DC) unscan_full(dc, coef_block(r->pic, m, P264_COEF_LUMA_DC));
        oracle_idct4x4dc(dc);
        oracle_dequant4x4_dc(dc, m->qp);
        for (int i = 0; i < 16; i++) {
            int16_t d[16];
            if (m->coef_mask & (1u << i)) { unscan_ac(d, coef_block(r->pic, m, 1u << i)); oracle_dequant4x4(d, m->qp); }
            else memset(d, 0, sizeof d);
            d[0] = dc[by_of[i]*4 + bx_of[i]];                               /* :793 */
            oracle_add4x4_idct(Y + by_of[i]*4*r->w + bx_of[i]*4, r->w, d);
        }
    } else {
        /* I4x4: decoder/macroblock.c:799-831; per-block availability core/macroblock.c:1210-1231 */
        static const uint16_t tr_inside = 0x5744;   /* bit i: top-right of block i (by>0) lies in an already decoded block */
        for (int i = 0; i < 16; i++) {
            int bx = bx_of[i], by = by_of[i];
            int left = bx > 0 || L, top = by > 0 || T;
            int topleft = (bx > 0 && by > 0) ? 1 : bx > 0 ? T : by > 0 ? L : TL;
            int topright = by == 0 ? (bx < 3 ? T : TR) : (tr_inside >> i) & 1;
            int mode = r->pic->i4modes[mbi*16 + i];
            if (mode == 2) mode = (left && top) ? 2 : left ? 9 : top ? 10 : 11;   /* valid_intra4x4_mode :669-719 */
            uint8_t l4[4], t8[8], o4[16];
            gather(r->y, r->w, mbx*16 + bx*4, mby*16 + by*4, 4, 8, left, top, topright, topleft, l4, t8, &tl);
            pred4(mode, l4, t8, tl, o4);
            uint8_t *d = Y + by*4*r->w + bx*4;
            for (int y = 0; y < 4; y++) memcpy(d + y*r->w, o4 + y*4, 4);
            if (m->coef_mask & (1u << i)) residual_luma_4x4(r, m, mbx, mby, i);
        }
    }
    /* chroma: decoder/macroblock.c:853-859 */
    int cmode = (m->intra_modes >> 4) & 3;
    if (cmode == 0) cmode = TL ? 0 : L ? 4 : T ? 5 : 6;                     /* valid_intra8x8c_mode :721-753 */
    for (int ch = 0; ch < 2; ch++) {
        uint8_t *plane = ch ? r->v : r->u;
        gather(plane, r->cw, mbx*8, mby*8, 8, 8, L, T, 0, TL, l, t, &tl);
        pred8c(cmode, l, t, tl, o);
        for (int y = 0; y < 8; y++) memcpy(plane + (mby*8 + y)*r->cw + mbx*8, o + y*8, 8);
    }
    recon_chroma_residual(r, m, mbx, mby);
}

int oracle_reconstruct_nodeblock(const p264hip_picture_t *pic, uint8_t **planes)
{
    rc_t r;
    r.pic = pic; r.planes = planes;
    r.w = pic->mb_w * 16; r.h = pic->mb_h * 16; r.cw = r.w / 2; r.ch = r.h / 2;
    r.y = planes[pic->dst_slot*3]; r.u = planes[pic->dst_slot*3 + 1]; r.v = planes[pic->dst_slot*3 + 2];
    for (int mby = 0; mby < pic->mb_h; mby++)
        for (int mbx = 0; mbx < pic->mb_w; mbx++) {
            int mbi = mby * pic->mb_w + mbx;
            const p264hip_mb_t *m = &pic->mb[mbi];
            if (P264_MB_IS_INTRA(m->mb_type)) recon_intra(&r, m, mbx, mby, mbi);
            else recon_inter(&r, m, mbx, mby, mbi);
        }
    return 0;
}

/* ---- picture-level loop filter (core/frame.c:472-643) ---------------------------------- */
static void edge(const p264hip_picture_t *pic, uint8_t *pix, int stride, int dir, const int bS[4], int qp, int chroma)
{   /* deblock_edge, core/frame.c:472-488; dir 0: vertical edge (filter across x) */
    int ia = clip3(qp + pic->alpha_c0_offset, 0, 51);
    int alpha = alpha_tab[ia], beta = beta_tab[clip3(qp + pic->beta_offset, 0, 51)];
    int xs = dir == 0 ? 1 : stride, ys = dir == 0 ? stride : 1;
    uint8_t before[16][8];
    int nl = chroma ? 8 : 16, nside = chroma ? 2 : 4;
    for (int l = 0; l < nl; l++) for (int k = 0; k < 2*nside; k++) before[l][k] = pix[l*ys + (k - nside)*xs];
    if (bS[0] < 4) {
        int8_t tc[4];
        for (int i = 0; i < 4; i++) tc[i] = (int8_t)((bS[i] ? tc0_tab[ia][bS[i] - 1] : -1) + chroma);
        if (chroma) oracle_deblock_chroma(pix, xs, ys, alpha, beta, tc);
        else        oracle_deblock_luma(pix, xs, ys, alpha, beta, tc);
    } else {
        if (chroma) oracle_deblock_chroma_intra(pix, xs, ys, alpha, beta);
        else        oracle_deblock_luma_intra(pix, xs, ys, alpha, beta);
    }
    for (int l = 0; l < nl; l++) {
        if (!bS[l * 4 / nl]) continue;
        int changed = 0;
        for (int k = 0; k < 2*nside; k++) changed |= before[l][k] != pix[l*ys + (k - nside)*xs];
        g_stats[chroma ? 4 : 2]++; g_stats[chroma ? 5 : 3] += changed;
    }
}

/* B pictures, H.264 8.7.2.1: the two blocks either side of an edge segment get strength 1 when they predict from different
 * reference PICTURES or a different number of vectors, or when vectors that belong to the same picture differ by >= 4
 * quarter-pels in a component.  The reference's loop (core/frame.c:565-577, encoder side: its decoder never gets here,
 * decoder/macroblock.c:168-171) compares list by list on the indices, which is only the same thing while no picture sits in
 * both lists.  Pictures are told apart by their frame-store slot; an unused list is slot -1 with a zero vector (the parser
 * keeps that), a quadrant without any list predicts from list 0, entry 0 (as the motion compensation reads it). */
static int b_slot(const p264hip_picture_t *pic, int list, int idx)
{
    if (idx < 0) return -1;
    if (list == 0) return pic->ref_slot[idx < pic->n_ref ? idx : 0];
    return pic->ref_slot_l1[idx < pic->n_ref_l1 ? idx : 0];
}
static int mv_far(const int16_t *a, const int16_t *b) { return iabs(a[0] - b[0]) >= 4 || iabs(a[1] - b[1]) >= 4; }
static int b_motion_strength(const p264hip_picture_t *pic, int mbi, int x, int y, int nbi, int xn, int yn)
{
    const int qp_ = mbi*4 + (y >> 1)*2 + (x >> 1), qn_ = nbi*4 + (yn >> 1)*2 + (xn >> 1);
    int p0 = b_slot(pic, 0, pic->ref_idx[qp_]), p1 = b_slot(pic, 1, pic->ref_idx_l1[qp_]);
    int q0 = b_slot(pic, 0, pic->ref_idx[qn_]), q1 = b_slot(pic, 1, pic->ref_idx_l1[qn_]);
    if (p0 < 0 && p1 < 0) p0 = pic->ref_slot[0];
    if (q0 < 0 && q1 < 0) q0 = pic->ref_slot[0];
    const int16_t *vp0 = pic->mv + (mbi*16 + y*4 + x)*2, *vq0 = pic->mv + (nbi*16 + yn*4 + xn)*2;
    const int16_t *vp1 = pic->mv_l1 + (mbi*16 + y*4 + x)*2, *vq1 = pic->mv_l1 + (nbi*16 + yn*4 + xn)*2;
    const int straight = p0 == q0 && p1 == q1 && !mv_far(vp0, vq0) && !mv_far(vp1, vq1);
    const int crossed  = p0 == q1 && p1 == q0 && !mv_far(vp0, vq1) && !mv_far(vp1, vq0);
    const int bs = !(straight || crossed);
    {   /* coverage: what the list-by-list comparison of indices would have said */
        int by_index = pic->ref_idx[qp_] != pic->ref_idx[qn_] || mv_far(vp0, vq0) || pic->ref_idx_l1[qp_] != pic->ref_idx_l1[qn_] || mv_far(vp1, vq1);
        if (by_index != bs) g_bs_by_picture++;
    }
    return bs;
}

int oracle_deblock_picture(const p264hip_picture_t *pic, uint8_t **planes)
{
    int w = pic->mb_w * 16, cw = w / 2;
    uint8_t *Y = planes[pic->dst_slot*3], *U = planes[pic->dst_slot*3 + 1], *V = planes[pic->dst_slot*3 + 2];
    for (int mby = 0; mby < pic->mb_h; mby++)
        for (int mbx = 0; mbx < pic->mb_w; mbx++) {
            int mbi = mby * pic->mb_w + mbx;
            const p264hip_mb_t *m = &pic->mb[mbi];
            if (!m->edges) continue;
            for (int dir = 0; dir < 2; dir++) {
                int first = (m->edges & (dir == 0 ? P264_EDGE_LEFT : P264_EDGE_TOP)) ? 0 : 1;   /* :524 */
                for (int e = first; e < 4; e++) {
                    int nbi = e > 0 ? mbi : dir == 0 ? mbi - 1 : mbi - pic->mb_w;
                    const p264hip_mb_t *n = &pic->mb[nbi];
                    int bS[4];
                    if (P264_MB_IS_INTRA(m->mb_type) || P264_MB_IS_INTRA(n->mb_type)) {
                        bS[0] = bS[1] = bS[2] = bS[3] = e == 0 ? 4 : 3;     /* :535-538 */
                    } else for (int i = 0; i < 4; i++) {                     /* :542-580 */
                        int x = dir == 0 ? e : i, y = dir == 0 ? i : e;
                        int xn = dir == 0 ? (x - 1) & 3 : x, yn = dir == 0 ? y : (y - 1) & 3;
                        if (((m->coef_mask >> blk_at[y][x]) & 1) || ((n->coef_mask >> blk_at[yn][xn]) & 1)) bS[i] = 2;
                        else {
                            int rp = pic->ref_idx[mbi*4 + (y >> 1)*2 + (x >> 1)], rq = pic->ref_idx[nbi*4 + (yn >> 1)*2 + (xn >> 1)];
                            const int16_t *vp = pic->mv + (mbi*16 + y*4 + x)*2, *vq = pic->mv + (nbi*16 + yn*4 + xn)*2;
                            bS[i] = (rp != rq || iabs(vp[0] - vq[0]) >= 4 || iabs(vp[1] - vq[1]) >= 4) ? 1 : 0;   /* :565-577, one list */
                            if (pic->slice_type == P264_SLICE_B) bS[i] = b_motion_strength(pic, mbi, x, y, nbi, xn, yn);
                        }
                    }
                    int qp = m->qp, qpn = n->qp;
                    if (qp != qpn && (bS[0] | bS[1] | bS[2] | bS[3])) g_stats[6]++;   /* an edge filtered with the mean of two QPs */
                    uint8_t *py = dir == 0 ? Y + mby*16*w + mbx*16 + 4*e : Y + (mby*16 + 4*e)*w + mbx*16;
                    edge(pic, py, w, dir, bS, (qp + qpn + 1) >> 1, 0);       /* :593-595, :615-617 */
                    if (!(e & 1)) {                                          /* :597-608, :620-630 */
                        int qc = (chroma_qp[clip3(qp + pic->chroma_qp_offset, 0, 51)] +
                                  chroma_qp[clip3(qpn + pic->chroma_qp_offset, 0, 51)] + 1) >> 1;
                        int off = dir == 0 ? mby*8*cw + mbx*8 + 2*e : (mby*8 + 2*e)*cw + mbx*8;
                        edge(pic, U + off, cw, dir, bS, qc, 1);
                        edge(pic, V + off, cw, dir, bS, qc, 1);
                    }
                }
            }
        }
    return 0;
}

int oracle_reconstruct(const p264hip_picture_t *pic, uint8_t **planes)
{   /* decoder/decoder.c:635-661: all MBs, then the loop filter (pads / half-pel planes are implicit) */
    oracle_reconstruct_nodeblock(pic, planes);
    if (pic->deblock) oracle_deblock_picture(pic, planes);
    return 0;
}
