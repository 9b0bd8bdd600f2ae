// kernel_inter.h - K1: inter prediction + residual for every non-intra macroblock of a batch.
//
// Replaces p264_mb_mc / p264_mb_mc_0xywh (core/macroblock.c:506-524,633-676), mc_luma /
// pixel_avg / mc_copy (core/mc.c:58-74,160-171,237-266), the half-pel plane generator
// p264_frame_filter (core/mc.c:172-235,409-451 - computed on the fly here, never stored),
// motion_compensation_chroma (core/mc.c:303-334), p264_macroblock_decode_skip
// (decoder/macroblock.c:895-934) and the inter half of p264_macroblock_decode
// (decoder/macroblock.c:832-890: unscan, dequant_4x4, add4x4_idct, chroma DC).
//
// Shape: one 64-lane wavefront per macroblock, four macroblocks (256 threads) per workgroup,
// all inter MBs of all pictures of the batch in one launch (they are independent).
//
// v2 structure (latency first, then instruction count):
//   1. header: MB record, 16 motion vectors, 4 reference indices;
//   2. ALL global loads of the macroblock are issued back to back - four 13x13 luma windows
//      (one per 8x8 quadrant, aligned dwords, rows clamped per lane), eight 5x5 chroma windows,
//      the coded coefficients (8 bytes per lane) - and land in LDS in one go;
//   3. every lane produces FOUR horizontally adjacent samples (one output dword): rows come out
//      of LDS as dwords, horizontal 6-tap = v_alignbyte + v_dot4_i32_i8 on sign-flipped bytes,
//      vertical 6-tap = packed 16-bit math, quarter-pel mean = byte-parallel rounding average;
//   4. residual added in registers; the lane stores its own dword (coalesced, no staging).
// Windows that cross the left/right picture edge, or quadrants whose four vectors differ
// (sub-8x8 partitions), take a per-lane clamped-read path.  Border padding
// (core/frame.c:183-222) is replaced by coordinate clamping, which is equivalent inside the
// reference's pads (SURVEY A-Q9).
#pragma once
#include "device_common.h"

#define YWIN_DW 56                // 13 rows x 4 dwords (+4 pad) per luma quadrant window
#define CWIN_DW 10                // 5 rows x 2 dwords per chroma quadrant window

struct InterLds {                 // per wavefront
    uint32_t ywin[4][YWIN_DW];
    uint32_t cwin[2][4][CWIN_DW];
    int16_t  coef[24 * 16];       // dequantised coefficients, raster order per block
};

typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t alignbyte(uint32_t hi, uint32_t lo, int sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }
__device__ __forceinline__ s16x2 as_s16x2(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ uint32_t as_u32(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
// byte-parallel (a + b + 1) >> 1
__device__ __forceinline__ uint32_t avg4(uint32_t a, uint32_t b) { return (a | b) - (((a ^ b) & 0xfefefefeu) >> 1); }
// hipcc (ROCm 7.2) fuses "arithmetic shift right -> clamp to 0..255 -> pack two bytes" into gfx950's
// v_ashr_pk_u8_i32 and then ORs further bytes into the upper half of its result, which the hardware
// does not leave zero (measured: wrong upper bytes in hv4).  An empty asm on the shifted value keeps
// the shift and the clamp apart; it emits no instruction.
__device__ __forceinline__ int no_fuse(int v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ uint32_t pack4(int a, int b, int c, int d) { return (uint32_t)a | ((uint32_t)b << 8) | ((uint32_t)c << 16) | ((uint32_t)d << 24); }

// ---- four samples from a 16-byte-per-row LDS window ---------------------------------------
// 4 bytes at row r, byte b (b + 3 <= 15)
__device__ __forceinline__ uint32_t row4(const uint32_t *w, int r, int b)
{
    const uint32_t *p = w + r * 4 + (b >> 2);
    return alignbyte(p[1], p[0], b & 3);
}
// 12 bytes at row r starting at byte `start` (only the first 9 are meaningful)
__device__ __forceinline__ void row12(const uint32_t *w, int r, int start, uint32_t &n0, uint32_t &n1, uint32_t &n2)
{
    const uint32_t *p = w + r * 4 + (start >> 2);
    uint32_t e0 = p[0], e1 = p[1], e2 = p[2];
    int s = start & 3;
    n0 = alignbyte(e1, e0, s); n1 = alignbyte(e2, e1, s); n2 = e2 >> (8 * s);
}
// horizontal 6-tap sums (core/mc.c:53-56) for 4 adjacent samples; n* hold bytes x-2 .. x+9
__device__ __forceinline__ void tap_h4(uint32_t n0, uint32_t n1, uint32_t n2, int t[4])
{
    const int C0 = 0x1414fb01, C1 = 0x000001fb;            // (1,-5,20,20) and (-5,1,0,0) as int8
    n0 ^= 0x80808080u; n1 ^= 0x80808080u; n2 ^= 0x80808080u;   // sample - 128 as int8; sum of taps = 32 -> bias 4096
    t[0] = __builtin_amdgcn_sdot4((int)n0, C0, __builtin_amdgcn_sdot4((int)n1, C1, 4096, false), false);
    t[1] = __builtin_amdgcn_sdot4((int)alignbyte(n1, n0, 1), C0, __builtin_amdgcn_sdot4((int)alignbyte(n2, n1, 1), C1, 4096, false), false);
    t[2] = __builtin_amdgcn_sdot4((int)alignbyte(n1, n0, 2), C0, __builtin_amdgcn_sdot4((int)alignbyte(n2, n1, 2), C1, 4096, false), false);
    t[3] = __builtin_amdgcn_sdot4((int)alignbyte(n1, n0, 3), C0, __builtin_amdgcn_sdot4((int)alignbyte(n2, n1, 3), C1, 4096, false), false);
}
__device__ __forceinline__ uint32_t h4(const uint32_t *w, int r, int b)       // mc_hh, core/mc.c:172-185
{
    uint32_t n0, n1, n2; int t[4];
    row12(w, r, b - 2, n0, n1, n2);
    tap_h4(n0, n1, n2, t);
    return pack4(clip255(no_fuse((t[0] + 16) >> 5)), clip255(no_fuse((t[1] + 16) >> 5)), clip255(no_fuse((t[2] + 16) >> 5)), clip255(no_fuse((t[3] + 16) >> 5)));
}
__device__ __forceinline__ uint32_t v4(const uint32_t *w, int r, int b)       // mc_hv, core/mc.c:186-199
{
    s16x2 lo[6], hi[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        uint32_t d = row4(w, r - 2 + k, b);
        lo[k] = as_s16x2(d & 0x00ff00ffu); hi[k] = as_s16x2((d >> 8) & 0x00ff00ffu);
    }
    const s16x2 c20 = { 20, 20 }, c5 = { 5, 5 }, c16 = { 16, 16 }, z = { 0, 0 }, m = { 255, 255 };
    s16x2 a = (lo[0] + lo[5]) + c20 * (lo[2] + lo[3]) - c5 * (lo[1] + lo[4]);
    s16x2 b2 = (hi[0] + hi[5]) + c20 * (hi[2] + hi[3]) - c5 * (hi[1] + hi[4]);
    a = (a + c16) >> 5; b2 = (b2 + c16) >> 5;
    a = __builtin_elementwise_min(__builtin_elementwise_max(a, z), m);
    b2 = __builtin_elementwise_min(__builtin_elementwise_max(b2, z), m);
    return as_u32(a) | (as_u32(b2) << 8);
}
__device__ __forceinline__ uint32_t hv4(const uint32_t *w, int r, int b)      // mc_hc, core/mc.c:200-235
{
    int acc[4] = { 512, 512, 512, 512 };
    const int cv[6] = { 1, -5, 20, 20, -5, 1 };
#pragma unroll
    for (int k = 0; k < 6; k++) {
        uint32_t n0, n1, n2; int t[4];
        row12(w, r - 2 + k, b - 2, n0, n1, n2);
        tap_h4(n0, n1, n2, t);
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] += cv[k] * t[i];
    }
    return pack4(clip255(no_fuse(acc[0] >> 10)), clip255(no_fuse(acc[1] >> 10)), clip255(no_fuse(acc[2] >> 10)), clip255(no_fuse(acc[3] >> 10)));
}
// four samples at half-pel coordinate (2x + hx, 2y + hy); plane choice as in core/mc.c:244-257
__device__ __forceinline__ uint32_t half4(const uint32_t *w, int r, int b, int hx, int hy)
{
    r += hy >> 1; b += hx >> 1;
    int which = (hx & 1) | ((hy & 1) << 1);
    if (which == 0) return row4(w, r, b);
    if (which == 1) return h4(w, r, b);
    if (which == 2) return v4(w, r, b);
    return hv4(w, r, b);
}
__device__ __forceinline__ uint32_t qpel4(const uint32_t *w, int r, int b, int fx, int fy)
{
    int corr = (fx & 1) && (fy & 1) && ((fx & 2) ^ (fy & 2));
    uint32_t a = half4(w, r, b, fx >> 1, (fy + 1 - corr) >> 1);
    if ((fx | fy) & 1) a = avg4(a, half4(w, r, b, (fx + 1) >> 1, (fy + corr) >> 1));
    return a;
}

// ---- per-sample fallback on the clamped plane (same arithmetic, one sample at a time) -------
struct ClampedPlane {
    const uint8_t *p; int w, h;
    __device__ __forceinline__ int operator()(int x, int y) const
    { return p[clip3i(y, 0, h - 1) * w + clip3i(x, 0, w - 1)]; }
};
template <class F> __device__ __forceinline__ int tap_h(const F &f, int x, int y)
{ return f(x-2, y) - 5*f(x-1, y) + 20*(f(x, y) + f(x+1, y)) - 5*f(x+2, y) + f(x+3, y); }
template <class F> __device__ __forceinline__ int tap_v(const F &f, int x, int y)
{ return f(x, y-2) - 5*f(x, y-1) + 20*(f(x, y) + f(x, y+1)) - 5*f(x, y+2) + f(x, y+3); }
template <class F> __device__ int half_sample(const F &f, int x, int y, int hx, int hy)
{
    x += hx >> 1; y += hy >> 1;
    int which = (hx & 1) | ((hy & 1) << 1);
    if (which == 0) return f(x, y);
    if (which == 1) return clip255((tap_h(f, x, y) + 16) >> 5);
    if (which == 2) return clip255((tap_v(f, x, y) + 16) >> 5);
    int t = tap_h(f, x, y-2) - 5*tap_h(f, x, y-1) + 20*(tap_h(f, x, y) + tap_h(f, x, y+1)) - 5*tap_h(f, x, y+2) + tap_h(f, x, y+3);
    return clip255((t + 512) >> 10);
}
template <class F> __device__ int qpel_sample(const F &f, int x, int y, int fx, int fy)
{
    int corr = (fx & 1) && (fy & 1) && ((fx & 2) ^ (fy & 2));
    int a = half_sample(f, x, y, fx >> 1, (fy + 1 - corr) >> 1);
    if ((fx | fy) & 1) a = (a + half_sample(f, x, y, (fx + 1) >> 1, (fy + corr) >> 1) + 1) >> 1;
    return a;
}
__device__ __forceinline__ int chroma_sample(const ClampedPlane &c, int sx, int sy, int mvx, int mvy)
{   // core/mc.c:303-334
    int dx = mvx & 7, dy = mvy & 7, x = sx + (mvx >> 3), y = sy + (mvy >> 3);
    return ((8 - dx) * (8 - dy) * c(x, y) + dx * (8 - dy) * c(x + 1, y) + (8 - dx) * dy * c(x, y + 1) + dx * dy * c(x + 1, y + 1) + 32) >> 6;
}

__device__ __forceinline__ int mv_x(int packed) { return (int)(int16_t)(packed & 0xffff); }
__device__ __forceinline__ int mv_y(int packed) { return packed >> 16; }

// four residual samples of row y of a 4x4 block (core/dct.c:205-247), c = 16 dequantised coefficients
__device__ __forceinline__ void idct4x4_row(const int16_t *c, int y, int r[4])
{
    int t[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int c0 = c[i*4], c1 = c[i*4+1], c2 = c[i*4+2], c3 = c[i*4+3];
        int s02 = c0 + c2, d02 = c0 - c2, s13 = c1 + (c3 >> 1), d13 = (c1 >> 1) - c3;
        t[i][0] = (int)(int16_t)(s02 + s13); t[i][1] = (int)(int16_t)(d02 + d13);
        t[i][2] = (int)(int16_t)(d02 - d13); t[i][3] = (int)(int16_t)(s02 - s13);
    }
#pragma unroll
    for (int x = 0; x < 4; x++) {
        int s02 = t[0][x] + t[2][x], d02 = t[0][x] - t[2][x], s13 = t[1][x] + (t[3][x] >> 1), d13 = (t[1][x] >> 1) - t[3][x];
        int v = y == 0 ? s02 + s13 : y == 1 ? d02 + d13 : y == 2 ? d02 - d13 : s02 - s13;
        r[x] = (int)(int16_t)((v + 32) >> 6);
    }
}
__device__ __forceinline__ uint32_t add_residual4(uint32_t pred, const int16_t *c, int y)
{
    int r[4];
    idct4x4_row(c, y, r);
    return pack4(clip255((int)(pred & 255) + r[0]), clip255((int)((pred >> 8) & 255) + r[1]),
                 clip255((int)((pred >> 16) & 255) + r[2]), clip255((int)(pred >> 24) + r[3]));
}

__global__ __launch_bounds__(256)
void k_inter(const PicDev *__restrict__ pics, Geom g, int blocks_per_pic, int n_blocks)
{
    __shared__ InterLds lds[4];
    // XCD-aware remap: the dispatcher deals workgroups round-robin over the 8 XCDs; give every XCD
    // one contiguous eighth of the batch so that overlapping reference windows share an L2.
    int per_xcd = gridDim.x >> 3;
    int logical = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (logical >= n_blocks) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pic = logical / blocks_per_pic;
    const int mbi = rfl((logical - pic * blocks_per_pic) * 4 + wave);
    if (mbi >= g.n_mb) return;
    const PicDev *pd = pics + pic;
    const p264hip_mb_t m = pd->mb[mbi];
    if (P264_MB_IS_INTRA(m.mb_type)) return;

    InterLds &L = lds[wave];
    const int mbx = mbi % g.mb_w, mby = mbi / g.mb_w;
    const int X0 = mbx * 16, Y0 = mby * 16;
    const int mvreg = lane < 16 ? pd->mv[mbi * 16 + lane] : 0;
    const int refs4 = *(const int *)(pd->ref_idx + mbi * 4);
    const unsigned mask = m.coef_mask;
    const int16_t *cf = pd->coefs + (size_t)m.coef_index * 16;

    // ---------------- per-quadrant set-up (wave-uniform) ----------------
    int qmv[4]; const uint8_t *qref[4]; unsigned fast = 0;      // fully unrolled below: stay in SGPRs
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int b0 = (q >> 1) * 8 + (q & 1) * 2;          // raster 4x4 index of the quadrant's first block
        const int mv0 = __builtin_amdgcn_readlane(mvreg, b0), mv1 = __builtin_amdgcn_readlane(mvreg, b0 + 1);
        const int mv2 = __builtin_amdgcn_readlane(mvreg, b0 + 4), mv3 = __builtin_amdgcn_readlane(mvreg, b0 + 5);
        int ri = (int)(int8_t)(refs4 >> (8 * q));
        if (ri < 0 || ri >= pd->n_ref) ri = 0;
        qref[q] = pd->ref[ri];
        qmv[q] = mv0;
        const int wx0 = X0 + (q & 1) * 8 + (mv_x(mv0) >> 2) - 2;             // luma window column 0
        const int cx0 = X0 / 2 + (q & 1) * 4 + (mv_x(mv0) >> 3);             // chroma window column 0
        if (mv0 == mv1 && mv0 == mv2 && mv0 == mv3 && wx0 >= 0 && wx0 + 12 < g.w && cx0 >= 0 && cx0 + 4 < g.cw) fast |= 1u << q;
    }

    // ---------------- issue every global load of this macroblock ----------------
    uint32_t yv[4], cv[2];
#pragma unroll
    for (int k = 0; k < 4; k++) {                            // 4 windows x 13 rows x 4 dwords = 208 dwords
        int i = lane + 64 * k;
        yv[k] = 0;
        if (i < 208) {
            int q = i / 52, rem = i - q * 52, r = rem >> 2, d = rem & 3;
            if ((fast >> q) & 1) {
                int mv = q == 0 ? qmv[0] : q == 1 ? qmv[1] : q == 2 ? qmv[2] : qmv[3];
                const uint8_t *rf = q == 0 ? qref[0] : q == 1 ? qref[1] : q == 2 ? qref[2] : qref[3];
                int wx0 = X0 + (q & 1) * 8 + (mv_x(mv) >> 2) - 2;
                int yy = clip3i(Y0 + (q >> 1) * 8 + (mv_y(mv) >> 2) - 2 + r, 0, g.h - 1);
                yv[k] = *(const uint32_t *)(rf + (size_t)yy * g.w + (wx0 & ~3) + d * 4);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {                            // 2 planes x 4 quadrants x 5 rows x 2 dwords = 80 dwords
        int i = lane + 64 * k;
        cv[k] = 0;
        if (i < 80) {
            int p = i / 40, rem = i - p * 40, q = rem / 10, rr = rem - q * 10, r = rr >> 1, d = rr & 1;
            if ((fast >> q) & 1) {
                int mv = q == 0 ? qmv[0] : q == 1 ? qmv[1] : q == 2 ? qmv[2] : qmv[3];
                const uint8_t *rf = q == 0 ? qref[0] : q == 1 ? qref[1] : q == 2 ? qref[2] : qref[3];
                int cx0 = X0 / 2 + (q & 1) * 4 + (mv_x(mv) >> 3);
                int yy = clip3i(Y0 / 2 + (q >> 1) * 4 + (mv_y(mv) >> 3) + r, 0, g.ch - 1);
                cv[k] = *(const uint32_t *)(rf + (p ? g.off_v : g.off_u) + (size_t)yy * g.cw + (cx0 & ~3) + d * 4);
            }
        }
    }
    // coded coefficients: luma block lane>>2, levels 4*(lane&3)..+3 ; chroma block 16+(lane>>3), levels 2*(lane&7)..+1
    uint2 lc = make_uint2(0, 0); uint32_t cc = 0; int cdc = 0;
    if (mask) {
        int lb = lane >> 2;
        if ((mask >> lb) & 1) lc = *(const uint2 *)(cf + coef_slot(mask, lb) * 16 + (lane & 3) * 4);
        int cb = 16 + (lane >> 3);
        if ((mask >> cb) & 1) cc = *(const uint32_t *)(cf + coef_slot(mask, cb) * 16 + (lane & 7) * 2);
        if ((mask & P264_COEF_CHROMA_DC) && lane < 8) cdc = cf[((mask >> 24) & 1) * 16 + lane];
    }

    // ---------------- land them in LDS ----------------
#pragma unroll
    for (int k = 0; k < 4; k++) { int i = lane + 64 * k; if (i < 208) { int q = i / 52; L.ywin[q][i - q * 52] = yv[k]; } }
#pragma unroll
    for (int k = 0; k < 2; k++) { int i = lane + 64 * k; if (i < 80) { int p = i / 40, rem = i - p * 40, q = rem / 10; L.cwin[p][q][rem - q * 10] = cv[k]; } }
    if (mask) {
        const int qp = m.qp;
        if (mask & 0xffff) {                                  // luma: unscan + dequant (decoder/macroblock.c:839-843)
            int lb = lane >> 2;
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                int k = (lane & 3) * 4 + kk, pos = c_zigzag[k];
                int c = (int)(int16_t)((kk & 2 ? lc.y : lc.x) >> (16 * (kk & 1)));
                L.coef[lb * 16 + pos] = (int16_t)dequant_coef(c, pos, qp);
            }
        }
        if (m.cbp >> 4) {                                     // chroma: DC (core/dct.c:55-68, core/quant.c:138-159) + AC
            const int qpc = c_chroma_qp[clip3i(qp + pd->chroma_qp_offset, 0, 51)];
            int cb = 16 + (lane >> 3), i2 = lane & 7;
#pragma unroll
            for (int kk = 0; kk < 2; kk++) {                  // level index 2*i2+kk sits at scan position 2*i2+kk+1
                int k = 2 * i2 + kk + 1;
                if (k < 16) { int pos = c_zigzag[k]; L.coef[cb * 16 + pos] = (int16_t)dequant_coef((int)(int16_t)(cc >> (16 * kk)), pos, qpc); }
            }
            // DC of chroma block j of plane p: lanes 0..7 hold the parsed DC levels of (p = lane>>2, index lane&3)
            int d0 = __shfl(cdc, (lane >> 5) * 4 + 0), d1 = __shfl(cdc, (lane >> 5) * 4 + 1);
            int d2 = __shfl(cdc, (lane >> 5) * 4 + 2), d3 = __shfl(cdc, (lane >> 5) * 4 + 3);
            if ((lane & 7) == 0) {
                int j = (lane >> 3) & 3;
                int t0 = d0 + d1, t1 = d0 - d1, t2 = d2 + d3, t3 = d2 - d3;
                int f = j == 0 ? t0 + t2 : j == 1 ? t1 + t3 : j == 2 ? t0 - t2 : t1 - t3;
                f = (int)(int16_t)f;
                int qbits = qpc / 6 - 5, mf = c_dqmf[qpc % 6][0];
                int v = qbits >= 0 ? f * (int)((unsigned)mf << qbits) : (f * mf) >> (-qbits);
                L.coef[cb * 16] = (int16_t)v;
            }
        }
    }
    wave_lds_fence();

    // ---------------- luma: lane = (row, dword) of the 16x16 block ----------------
    const int row = lane >> 2, dw = lane & 3;
    uint32_t outY;
    {
        const int q = (row >> 3) * 2 + (dw >> 1);
        const int mvl = __shfl(mvreg, (row >> 2) * 4 + dw);
        const int lx = mv_x(mvl), ly = mv_y(mvl);
        if ((fast >> q) & 1) {
            const int wx0 = X0 + (q & 1) * 8 + (lx >> 2) - 2;
            outY = qpel4(L.ywin[q], (row & 7) + 2, (wx0 & 3) + 2 + (dw & 1) * 4, lx & 3, ly & 3);
        } else {
            int ri = (int)(int8_t)(refs4 >> (8 * q));
            if (ri < 0 || ri >= pd->n_ref) ri = 0;
            ClampedPlane f = { pd->ref[ri], g.w, g.h };
            int v[4];
            for (int i = 0; i < 4; i++) v[i] = qpel_sample(f, X0 + dw * 4 + i + (lx >> 2), Y0 + row + (ly >> 2), lx & 3, ly & 3);
            outY = pack4(v[0], v[1], v[2], v[3]);
        }
        const int blk = blk_at(dw, row >> 2);
        if ((mask >> blk) & 1) outY = add_residual4(outY, L.coef + blk * 16, row & 3);
    }
    *(uint32_t *)(pd->dst + (size_t)(Y0 + row) * g.w + X0 + dw * 4) = outY;

    // ---------------- chroma: lanes 0..31 = (plane, row, dword) of the two 8x8 blocks ----------------
    if (lane < 32) {
        const int p = lane >> 4, crow = (lane >> 1) & 7, cdw = lane & 1;
        const int q = (crow >> 2) * 2 + cdw;
        uint32_t outC;
        // vectors of the two 4x4 luma blocks this dword spans (equal on the fast path); shuffled here,
        // outside the divergent branch, so that the source lanes 0..15 are active
        const int mvA = __shfl(mvreg, (crow >> 1) * 4 + cdw * 2), mvB = __shfl(mvreg, (crow >> 1) * 4 + cdw * 2 + 1);
        if ((fast >> q) & 1) {
            const int lx = mv_x(mvA), ly = mv_y(mvA);
            const int cx0 = X0 / 2 + (q & 1) * 4 + (lx >> 3);
            const uint32_t *w = L.cwin[p][q] + (crow & 3) * 2;
            const int off = cx0 & 3;
            const unsigned long long r0 = ((unsigned long long)w[1] << 32) | w[0], r1 = ((unsigned long long)w[3] << 32) | w[2];
            const uint32_t a = (uint32_t)(r0 >> (8 * off)), b = (uint32_t)(r0 >> (8 * off + 8));
            const uint32_t c = (uint32_t)(r1 >> (8 * off)), d = (uint32_t)(r1 >> (8 * off + 8));
            const int dx = lx & 7, dy = ly & 7;
            const short cA = (short)((8 - dx) * (8 - dy)), cB = (short)(dx * (8 - dy)), cC = (short)((8 - dx) * dy), cD = (short)(dx * dy);
            const s16x2 kA = { cA, cA }, kB = { cB, cB }, kC = { cC, cC }, kD = { cD, cD }, k32 = { 32, 32 };
            s16x2 lo = kA * as_s16x2(a & 0x00ff00ffu) + kB * as_s16x2(b & 0x00ff00ffu) + kC * as_s16x2(c & 0x00ff00ffu) + kD * as_s16x2(d & 0x00ff00ffu) + k32;
            s16x2 hi = kA * as_s16x2((a >> 8) & 0x00ff00ffu) + kB * as_s16x2((b >> 8) & 0x00ff00ffu) + kC * as_s16x2((c >> 8) & 0x00ff00ffu) + kD * as_s16x2((d >> 8) & 0x00ff00ffu) + k32;
            // sums reach 64*255+32 = 16352 < 32768: the 16-bit lanes never overflow; logical shift of non-negative values
            outC = ((as_u32(lo) >> 6) & 0x00ff00ffu) | (((as_u32(hi) >> 6) & 0x00ff00ffu) << 8);
        } else {
            int ri = (int)(int8_t)(refs4 >> (8 * q));
            if (ri < 0 || ri >= pd->n_ref) ri = 0;
            ClampedPlane f = { pd->ref[ri] + (p ? g.off_v : g.off_u), g.cw, g.ch };
            int v[4];
            for (int i = 0; i < 4; i++) {
                int mv = i < 2 ? mvA : mvB;
                v[i] = chroma_sample(f, X0 / 2 + cdw * 4 + i, Y0 / 2 + crow, mv_x(mv), mv_y(mv));
            }
            outC = pack4(v[0], v[1], v[2], v[3]);
        }
        if (mask && (m.cbp >> 4)) outC = add_residual4(outC, L.coef + (16 + p * 4 + (crow >> 2) * 2 + cdw) * 16, crow & 3);
        *(uint32_t *)(pd->dst + (p ? g.off_v : g.off_u) + (size_t)(Y0 / 2 + crow) * g.cw + X0 / 2 + cdw * 4) = outC;
    }
}
