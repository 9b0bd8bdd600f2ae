// kernel_inter.h - K1: inter prediction + residual for every non-intra macroblock of a batch.
//
// Replaces p264_mb_mc / p264_mb_mc_0xywh (core/macroblock.c:506-524,633-676), mc_luma /
// pixel_avg / mc_copy (core/mc.c:58-74,160-171,237-266), the half-pel plane generator
// p264_frame_filter (core/mc.c:172-235,409-451 - computed on the fly here, never stored),
// motion_compensation_chroma (core/mc.c:303-334), p264_macroblock_decode_skip
// (decoder/macroblock.c:895-934) and the inter half of p264_macroblock_decode
// (decoder/macroblock.c:832-890: unscan, dequant_4x4, add4x4_idct, chroma DC).
//
// Two kernels (both instruction-issue bound, see DESIGN.md section 4 - the structure is about instructions and round
// trips per macroblock, the frames are macroblock-tiled so every access covers whole cache lines):
//
// k_inter        one 64-lane wavefront per macroblock, four macroblocks per workgroup, all pictures of the batch in
//                one launch.  Macroblocks with ONE vector (16x16, P_SKIP) take the whole-MB path:
//   1. descriptor (one scalar load), then MB record + 16 vectors + 4 reference indices in one round trip;
//   2. coded coefficients (8 bytes per lane) and the reference window - 21x24 luma + 2 x 9x12 chroma bytes as aligned
//      dwords, three load instructions - issued back to back, landing in LDS; rows are clamped through y, a dword
//      outside the picture is the replicated border byte (core/frame.c:183-222 without storing the pads, SURVEY A-Q9);
//   3. every lane produces FOUR horizontally adjacent samples (one dword of the tile): horizontal 6-tap =
//      v_alignbyte + v_dot4_i32_i8 on sign-flipped bytes, vertical 6-tap and chroma bilinear = packed 16-bit math,
//      the centre position from a table of horizontal sums shared by the wavefront, quarter-pel mean = byte-parallel
//      rounding average;
//   4. residual: levels dequantised into LDS, inverse transform shared between the four lanes of a block, added in
//      registers; the macroblock leaves as three whole cache lines.
//                Macroblocks flagged P264_MBF_QUADS are skipped (k_inter_quads); other multi-vector macroblocks
//                (no quadrant list, or sub-8x8 partitions with differing vectors) take the quadrant path below with
//                one interpolation pass per distinct vector, and a rolled per-sample path for non-uniform quadrants.
// k_inter_quads  at the end of this file: the quadrants of multi-vector macroblocks, four of one quarter-pel phase
//                per wavefront, from the parser's phase-sorted list.
#pragma once
#include "device_common.h"
#ifndef EXP_NOCOMPUTE
#define EXP_NOCOMPUTE 0
#endif
#ifndef EXP_NT
#define EXP_NT 0
#endif
#if EXP_NT
typedef const __attribute__((address_space(1))) uint32_t *gu32p;
#define WLOAD(p) __builtin_nontemporal_load((gu32p)(const uint32_t *)(p))
#else
#define WLOAD(p) gload1(p)
#endif
#ifndef EXP_NORESID
#define EXP_NORESID 0
#endif
#ifndef EXP_NOSTORE
#define EXP_NOSTORE 0
#endif
#ifndef EXP_NOLOAD
#define EXP_NOLOAD 0
#endif
#ifndef EXP_HDRONLY
#define EXP_HDRONLY 0
#endif
#ifndef EXP_FORCE            // 1: every MB takes the whole-MB path, 2: every MB the quadrant path (timing experiments only)
#define EXP_FORCE 0
#endif

#define YWIN_DW 56                // 13 rows x 4 dwords (+4 pad) per luma quadrant window
#define CWIN_DW 10                // 5 rows x 2 dwords per chroma quadrant window

struct InterLds {                 // per wavefront
    uint32_t ywin[4][YWIN_DW];    // whole-MB path: 21 rows x 6 dwords (+2 pad); quadrant path: 13 x 4 luma + 20 chroma dwords
    uint32_t cwin[2][4][16];      // whole-MB path: 2 planes x 9 rows x 3 dwords (+pad)
    int16_t  coef[24 * 16];       // dequantised coefficients, raster order per block
    uint32_t hrow[224];           // unrounded horizontal 6-tap sums for the centre half-pel position, 4 x int16 per (row, dword column)
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

// ---- four samples from an LDS window with a row pitch of P dwords ---------------------------
// 4 bytes at row r, byte b
template <int P> __device__ __forceinline__ uint32_t row4(const uint32_t *w, int r, int b)
{
    const uint32_t *p = w + r * P + (b >> 2);
    return alignbyte(p[1], p[0], b & 3);
}
// 12 bytes at row r starting at byte `start` (only the first 9 are meaningful)
template <int P> __device__ __forceinline__ void row12(const uint32_t *w, int r, int start, uint32_t &n0, uint32_t &n1, uint32_t &n2)
{
    const uint32_t *p = w + r * P + (start >> 2);
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
template <int P> __device__ __forceinline__ uint32_t h4(const uint32_t *w, int r, int b)       // mc_hh, core/mc.c:172-185
{
    uint32_t n0, n1, n2; int t[4];
    row12<P>(w, r, b - 2, n0, n1, n2);
    tap_h4(n0, n1, n2, t);
    return pack4(clip255(no_fuse((t[0] + 16) >> 5)), clip255(no_fuse((t[1] + 16) >> 5)), clip255(no_fuse((t[2] + 16) >> 5)), clip255(no_fuse((t[3] + 16) >> 5)));
}
template <int P> __device__ __forceinline__ uint32_t v4(const uint32_t *w, int r, int b)       // mc_hv, core/mc.c:186-199
{
    s16x2 lo[6], hi[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        uint32_t d = row4<P>(w, r - 2 + k, b);
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
template <int P> __device__ __forceinline__ uint32_t hv4(const uint32_t *w, int r, int b)      // mc_hc, core/mc.c:200-235
{
    int acc[4] = { 512, 512, 512, 512 };
    const int cv[6] = { 1, -5, 20, 20, -5, 1 };
#pragma unroll 1
    for (int k = 0; k < 6; k++) {
        uint32_t n0, n1, n2; int t[4];
        row12<P>(w, r - 2 + k, b - 2, n0, n1, n2);
        tap_h4(n0, n1, n2, t);
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] += cv[k] * t[i];
    }
    return pack4(clip255(no_fuse(acc[0] >> 10)), clip255(no_fuse(acc[1] >> 10)), clip255(no_fuse(acc[2] >> 10)), clip255(no_fuse(acc[3] >> 10)));
}
// The centre position shared between lanes: hv4 above runs the horizontal filter on six rows per lane, and vertically
// adjacent lanes repeat five of them.  When a whole wavefront needs the centre position (the phase is wave-uniform) the
// lanes first fill a table of horizontal sums, one (row, dword column) each - h4_raw - and then every lane runs only the
// vertical filter over six table rows - hv4_from_rows.  Same arithmetic: unrounded 16-bit sums, (sum + 512) >> 10.
template <int P> __device__ __forceinline__ uint2 h4_raw(const uint32_t *w, int r, int b)
{
    uint32_t n0, n1, n2; int t[4];
    row12<P>(w, r, b - 2, n0, n1, n2);
    tap_h4(n0, n1, n2, t);
    return make_uint2(((uint32_t)t[0] & 0xffffu) | ((uint32_t)t[1] << 16), ((uint32_t)t[2] & 0xffffu) | ((uint32_t)t[3] << 16));
}
__device__ __forceinline__ uint32_t hv4_from_rows(const uint32_t *h, int pitch)
{
    int acc[4] = { 512, 512, 512, 512 };
    const int cv[6] = { 1, -5, 20, 20, -5, 1 };
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const uint2 v = *(const uint2 *)(h + k * pitch);
        acc[0] += cv[k] * (int)(int16_t)(v.x & 0xffff); acc[1] += cv[k] * ((int)v.x >> 16);
        acc[2] += cv[k] * (int)(int16_t)(v.y & 0xffff); acc[3] += cv[k] * ((int)v.y >> 16);
    }
    return pack4(clip255(no_fuse(acc[0] >> 10)), clip255(no_fuse(acc[1] >> 10)), clip255(no_fuse(acc[2] >> 10)), clip255(no_fuse(acc[3] >> 10)));
}
// the phases that use the centre position (core/mc.c:244-257): always at half-pel offset (1,1) of the lane's own samples
__device__ __forceinline__ bool phase_uses_centre(int fx, int fy) { return (fx == 2 && fy != 0) || (fy == 2 && fx != 0); }

// four samples at half-pel coordinate (2x + hx, 2y + hy); plane choice as in core/mc.c:244-257.  hrows: this lane's six rows
// of the shared table (pitch hp dwords), or null
template <int P> __device__ __forceinline__ uint32_t half4(const uint32_t *w, int r, int b, int hx, int hy, const uint32_t *hrows = nullptr, int hp = 0)
{
    r += hy >> 1; b += hx >> 1;
    int which = (hx & 1) | ((hy & 1) << 1);
    if (which == 0) return row4<P>(w, r, b);
    if (which == 1) return h4<P>(w, r, b);
    if (which == 2) return v4<P>(w, r, b);
    return hrows ? hv4_from_rows(hrows, hp) : hv4<P>(w, r, b);
}
template <int P> __device__ __forceinline__ uint32_t qpel4(const uint32_t *w, int r, int b, int fx, int fy, const uint32_t *hrows = nullptr, int hp = 0)
{
    int corr = (fx & 1) && (fy & 1) && ((fx & 2) ^ (fy & 2));
    uint32_t a = half4<P>(w, r, b, fx >> 1, (fy + 1 - corr) >> 1, hrows, hp);
    if ((fx | fy) & 1) a = avg4(a, half4<P>(w, r, b, (fx + 1) >> 1, (fy + corr) >> 1, hrows, hp));
    return a;
}

// ---- per-sample fallback on the clamped plane (same arithmetic, one sample at a time) -------
struct ClampedPlane {
    const uint8_t *p; int w, h, mb_w, plane;          // plane: -1 luma, 0 U, 1 V (tiled frame, device_common.h)
    __device__ __forceinline__ int operator()(int x, int y) const
    {
        x = clip3i(x, 0, w - 1); y = clip3i(y, 0, h - 1);
        if (plane < 0) return glob(p)[(uint32_t)((y >> 4) * mb_w + (x >> 4)) * MB_TILE + (y & 15) * 16 + (x & 15)];
        return glob(p)[(uint32_t)((y >> 3) * mb_w + (x >> 3)) * MB_TILE + MB_TILE_U + plane * 64 + (y & 7) * 8 + (x & 7)];
    }
};
// Deliberately rolled loops: this path is rare (sub-8x8 partitions, windows crossing the left/right
// picture edge) and must not set the register budget of the kernel.
__device__ __forceinline__ int tap_coef(int k) { return (k == 0 || k == 5) ? 1 : (k == 1 || k == 4) ? -5 : 20; }
__device__ __noinline__ int half_sample(const uint8_t *p, int w, int h, int mb_w, int x, int y, int hx, int hy)
{
    ClampedPlane f = { p, w, h, mb_w, -1 };
    x += hx >> 1; y += hy >> 1;
    int which = (hx & 1) | ((hy & 1) << 1);
    if (which == 0) return f(x, y);
    int s = 0;
    if (which == 1) {
#pragma unroll 1
        for (int k = 0; k < 6; k++) s += tap_coef(k) * f(x - 2 + k, y);
        return clip255((s + 16) >> 5);
    }
    if (which == 2) {
#pragma unroll 1
        for (int k = 0; k < 6; k++) s += tap_coef(k) * f(x, y - 2 + k);
        return clip255((s + 16) >> 5);
    }
#pragma unroll 1
    for (int j = 0; j < 6; j++) {
        int t = 0;
#pragma unroll 1
        for (int k = 0; k < 6; k++) t += tap_coef(k) * f(x - 2 + k, y - 2 + j);
        s += tap_coef(j) * t;
    }
    return clip255((s + 512) >> 10);
}
__device__ __forceinline__ int qpel_sample(const uint8_t *p, int w, int h, int mb_w, int x, int y, int fx, int fy)
{
    int corr = (fx & 1) && (fy & 1) && ((fx & 2) ^ (fy & 2));
    int a = half_sample(p, w, h, mb_w, x, y, fx >> 1, (fy + 1 - corr) >> 1);
    if ((fx | fy) & 1) a = (a + half_sample(p, w, h, mb_w, x, y, (fx + 1) >> 1, (fy + corr) >> 1) + 1) >> 1;
    return a;
}
__device__ __forceinline__ int chroma_sample(const ClampedPlane &c, int sx, int sy, int mvx, int mvy)
{   // core/mc.c:303-334
    int dx = mvx & 7, dy = mvy & 7, x = sx + (mvx >> 3), y = sy + (mvy >> 3);
    return ((8 - dx) * (8 - dy) * c(x, y) + dx * (8 - dy) * c(x + 1, y) + (8 - dx) * dy * c(x, y + 1) + dx * dy * c(x + 1, y + 1) + 32) >> 6;
}

// Window dwords on the padded plane.  The reference extends the picture by replicating its border
// samples (core/frame.c:183-222, A-Q9); rows are clamped through y, and because dword loads are 4-aligned
// and the plane widths are multiples of 8, a dword is either entirely inside the picture or entirely
// outside: outside, it is the replicated first (last) byte of the row's first (last) dword.
__device__ __forceinline__ uint32_t edge_fix(uint32_t v, int xa, int w)
{
    const uint32_t sel = xa < 0 ? 0x00000000u : xa >= w ? 0x03030303u : 0x03020100u;
    return __builtin_amdgcn_perm(v, v, sel);
}
__device__ __forceinline__ uint32_t luma_dword(const uint8_t *ref, const Geom &g, int xa, int y)
{
    return edge_fix(WLOAD(ref + luma_off(g, clip3i(xa, 0, g.w - 4), clip3i(y, 0, g.h - 1))), xa, g.w);
}
__device__ __forceinline__ uint32_t chroma_dword(const uint8_t *ref, const Geom &g, int plane, int xa, int y)
{
    return edge_fix(WLOAD(ref + chroma_off(g, plane, clip3i(xa, 0, g.cw - 4), clip3i(y, 0, g.ch - 1))), xa, g.cw);
}

__device__ __forceinline__ int mv_x(int packed) { return (int)(int16_t)(packed & 0xffff); }
__device__ __forceinline__ int mv_y(int packed) { return packed >> 16; }

__device__ __forceinline__ uint32_t idct4x4_colpass_add(uint32_t pred, const int16_t *c, int y)
{
    const uint4 a = *(const uint4 *)c, b = *(const uint4 *)(c + 8);     // rows 0,1 | rows 2,3 of the intermediate
    const uint32_t row[4][2] = { { a.x, a.y }, { a.z, a.w }, { b.x, b.y }, { b.z, b.w } };
    int r[4];
#pragma unroll
    for (int x = 0; x < 4; x++) {
        int t[4];
#pragma unroll
        for (int i = 0; i < 4; i++) t[i] = (x & 1) ? (int)row[i][x >> 1] >> 16 : (int)(int16_t)(row[i][x >> 1] & 0xffff);
        const int s02 = t[0] + t[2], d02 = t[0] - t[2], s13 = t[1] + (t[3] >> 1), d13 = (t[1] >> 1) - t[3];
        r[x] = (int)(int16_t)((butterfly_pick(s02, d02, s13, d13, y) + 32) >> 6);
    }
    return pack4(clip255((int)(pred & 255) + r[0]), clip255((int)((pred >> 8) & 255) + r[1]),
                 clip255((int)((pred >> 16) & 255) + r[2]), clip255((int)(pred >> 24) + r[3]));
}

// chroma 1/8-pel bilinear, 4 samples (core/mc.c:303-334): w = the sample row inside a window of row pitch P dwords,
// b = byte offset of the first sample (b + 4 <= 4P - 1 is guaranteed by the staging); (dx,dy) wave-uniform
template <int P> __device__ __forceinline__ uint32_t chroma4(const uint32_t *w, int b, int dx, int dy)
{
    const uint32_t *p = w + (b >> 2);
    const int s = b & 3;
    const unsigned long long r0 = ((unsigned long long)p[1] << 32) | p[0], r1 = ((unsigned long long)p[P + 1] << 32) | p[P];
    const uint32_t a = (uint32_t)(r0 >> (8 * s)), bb = (uint32_t)(r0 >> (8 * s + 8));
    const uint32_t c = (uint32_t)(r1 >> (8 * s)), d = (uint32_t)(r1 >> (8 * s + 8));
    const short cA = (short)((8 - dx) * (8 - dy)), cB = (short)(dx * (8 - dy)), cC = (short)((8 - dx) * dy), cD = (short)(dx * dy);
    const s16x2 kA = { cA, cA }, kB = { cB, cB }, kC = { cC, cC }, kD = { cD, cD }, k32 = { 32, 32 };
    s16x2 lo = kA * as_s16x2(a & 0x00ff00ffu) + kB * as_s16x2(bb & 0x00ff00ffu) + kC * as_s16x2(c & 0x00ff00ffu) + kD * as_s16x2(d & 0x00ff00ffu) + k32;
    s16x2 hi = kA * as_s16x2((a >> 8) & 0x00ff00ffu) + kB * as_s16x2((bb >> 8) & 0x00ff00ffu) + kC * as_s16x2((c >> 8) & 0x00ff00ffu) + kD * as_s16x2((d >> 8) & 0x00ff00ffu) + k32;
    // sums reach 64*255+32 = 16352 < 32768: the 16-bit lanes never overflow
    return ((as_u32(lo) >> 6) & 0x00ff00ffu) | (((as_u32(hi) >> 6) & 0x00ff00ffu) << 8);
}

// rare paths kept out of line so that they do not set the register budget of the kernel
__device__ __noinline__ uint32_t slow_luma4(const uint8_t *ref, int w, int h, int mb_w, int x, int y, int mv)
{
    uint32_t out = 0;
#pragma unroll 1
    for (int i = 0; i < 4; i++) out |= (uint32_t)qpel_sample(ref, w, h, mb_w, x + i + (mv_x(mv) >> 2), y + (mv_y(mv) >> 2), mv_x(mv) & 3, mv_y(mv) & 3) << (8 * i);
    return out;
}
__device__ __noinline__ uint32_t slow_chroma4(const uint8_t *ref, int w, int h, int mb_w, int plane, int x, int y, int mvA, int mvB)
{
    ClampedPlane f = { ref, w, h, mb_w, plane };
    uint32_t out = 0;
#pragma unroll 1
    for (int i = 0; i < 4; i++) { int m2 = i < 2 ? mvA : mvB; out |= (uint32_t)chroma_sample(f, x + i, y, mv_x(m2), mv_y(m2)) << (8 * i); }
    return out;
}

#ifndef INTER_WAVES_PER_EU
#define INTER_WAVES_PER_EU 8
#endif
__global__ __launch_bounds__(256, INTER_WAVES_PER_EU)
void k_inter(const PicDev *__restrict__ pics, Geom g, int blocks_per_pic, int n_blocks, uint32_t inv_bpp, uint32_t inv_mbw)
{
    __shared__ InterLds lds[4];
    // XCD-aware remap: the dispatcher deals workgroups round-robin over the 8 XCDs; give every XCD
    // one contiguous eighth of the batch so that overlapping reference windows share an L2.
    int per_xcd = gridDim.x >> 3;
    int logical = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (logical >= n_blocks) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // divisions by launch constants: multiply by the host's floor(2^32/d) and correct the estimate (never more than one short)
    int pic = (int)__umulhi((unsigned)logical, inv_bpp);
    if (logical - pic * blocks_per_pic >= blocks_per_pic) pic++;
    const int mbi = rfl((logical - pic * blocks_per_pic) * 4 + wave);
    if (mbi >= g.n_mb) return;
    const PicDev *pd = pics + pic;

    // ---------------- header: everything wave-uniform goes to SGPRs ----------------
    // the three header loads go out together; nothing waits before all of them are in flight
    const PicHead ph = load_pic_head(pd);                      // one round trip for the whole descriptor
    const uint4 rec = gload4(ph.mb + mbi);
    const int mvreg_raw = glob(ph.mv)[mbi * 16 + (lane & 15)];
    const int refs4_raw = (int)gload1(ph.ref_idx + mbi * 4);
    const uint8_t *ref0 = ph.ref0;                             // by far the most common reference
    const unsigned w0 = (unsigned)rfl((int)rec.x), mask = (unsigned)rfl((int)rec.y);
    const int mb_type = w0 & 255, qp = (w0 >> 8) & 255, cbp = (w0 >> 16) & 255;
    // (the second term is never true; it makes the vector and the reference indices part of this branch, so that their
    // loads are issued with the record's instead of being sunk below it - one round trip instead of two)
    // macroblocks flagged P264_MBF_QUADS are motion-compensated by k_inter_quads
    if (P264_MB_IS_INTRA(mb_type) | (int)((rfl((int)rec.w) >> 16) & P264_MBF_QUADS) | ((ph.n_ref < 0) & (__ballot(mvreg_raw == 0x7fffffff) != 0) & (refs4_raw == 0x7fffffff))) return;
    const int16_t *cf = ph.coefs + (size_t)(unsigned)rfl((int)rec.z) * 16;
    const int mvreg = lane < 16 ? mvreg_raw : 0;
    const int refs4 = rfl(refs4_raw);
    const int n_ref = ph.n_ref;

    InterLds &L = lds[wave];
    int mby = (int)__umulhi((unsigned)mbi, inv_mbw);
    if (mbi - mby * g.mb_w >= g.mb_w) mby++;
    const int mbx = mbi - mby * g.mb_w;
    const int X0 = mbx * 16, Y0 = mby * 16;

    // coded coefficients are fetched now, whatever path the prediction takes:
    // luma block lane>>2, levels 4*(lane&3)..+3 ; chroma block 16+(lane>>3), levels 2*(lane&7)..+1
    uint2 lc = make_uint2(0, 0); uint32_t cc = 0, cdc_raw = 0;
    if (mask) {
        int lb = lane >> 2;
        if ((mask >> lb) & 1) lc = gload2(cf + coef_slot(mask, lb) * 16 + (lane & 3) * 4);
        int cb = 16 + (lane >> 3);
        if ((mask >> cb) & 1) cc = gload1(cf + coef_slot(mask, cb) * 16 + (lane & 7) * 2);
        if ((mask & P264_COEF_CHROMA_DC) && lane < 8) cdc_raw = glob((const uint16_t *)cf)[((mask >> 24) & 1) * 16 + lane];
    }

    if (EXP_HDRONLY) { if (mvreg == 0x7fffffff) ph.dst[0] = 1; return; }
    const int row = lane >> 2, dw = lane & 3;                 // luma: lane = (row, dword) of the 16x16 block
    const int crow = (lane >> 1) & 7, cdw = lane & 1, cp = (lane >> 4) & 1;   // chroma (lanes 0..31): (plane, row, dword)
    uint32_t outY = 0, outC = 0;

    const int mv0 = __builtin_amdgcn_readfirstlane(mvreg);
    int r0i = (int)(int8_t)refs4;
    if ((unsigned)r0i >= (unsigned)n_ref) r0i = 0;             // negative or past the list: entry 0, as the reference's flat lists
    const bool same_mv = __ballot(lane < 16 && mvreg != mv0) == 0 && (unsigned)refs4 == ((unsigned)(refs4 & 255) * 0x01010101u);
    const int ux0 = X0 + (mv_x(mv0) >> 2) - 2, ucx0 = X0 / 2 + (mv_x(mv0) >> 3);

    if (EXP_FORCE != 2 && (EXP_FORCE == 1 || same_mv)) {
        // ======== one vector for the whole macroblock (16x16 partitions and P_SKIP) ========
        // luma window 21 rows x 6 dwords, chroma windows 2 x 9 rows x 3 dwords: three load instructions
        const uint8_t *rf = r0i == 0 ? ref0 : pd->ref[r0i];
        const int lx = mv_x(mv0), ly = mv_y(mv0);
        // (lanes past the end of a window repeat its last dword and write it to unused LDS words: cheaper than masking them off)
        uint32_t yv[2], cvv = 0;
        const int wy0 = Y0 + (ly >> 2) - 2, wcy0 = Y0 / 2 + (ly >> 3);
        // windows that lie inside the picture (nearly all) need neither coordinate clamps nor border replication
        // (0 <= v <= limit for four values at once: v | (limit - v) keeps its sign bit clear exactly then - one test instead of
        // eight short-circuit branches; the coordinates are far below 2^30)
        const int uxa = ux0 & ~3, ucxa = ucx0 & ~3;
        const bool inside = ((uxa | (g.w - 24 - uxa)) | (wy0 | (g.h - 21 - wy0)) | (ucxa | (g.cw - 12 - ucxa)) | (wcy0 | (g.ch - 9 - wcy0))) >= 0;
        const int cl = min(lane, 53);
        const int cpl = cl >= 27, l2 = cl - 27 * cpl, cr = (l2 * 11) >> 5, cd = l2 - 3 * cr;   // l2 / 3 for l2 < 27
        if (inside) {
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int i = min(lane + 64 * k, 125), r = (i * 43) >> 8, d = i - r * 6;       // i / 6 for i < 128
                yv[k] = 0;
                if (!EXP_NOLOAD) yv[k] = WLOAD(rf + luma_off(g, (ux0 & ~3) + d * 4, wy0 + r));
            }
            if (!EXP_NOLOAD) cvv = WLOAD(rf + chroma_off(g, cpl, (ucx0 & ~3) + cd * 4, wcy0 + cr));
        } else {
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int i = min(lane + 64 * k, 125), r = (i * 43) >> 8, d = i - r * 6;
                yv[k] = 0;
                if (!EXP_NOLOAD) yv[k] = luma_dword(rf, g, (ux0 & ~3) + d * 4, wy0 + r);
            }
            if (!EXP_NOLOAD) cvv = chroma_dword(rf, g, cpl, (ucx0 & ~3) + cd * 4, wcy0 + cr);
        }
        uint32_t *yw = &L.ywin[0][0], *cw = &L.cwin[0][0][0];
        yw[lane] = yv[0];
        yw[64 + lane] = yv[1];
        cw[lane] = cvv;
        wave_lds_fence();
#if EXP_NOCOMPUTE
        outY = yw[lane]; outC = cw[lane & 31];
#else
        if (phase_uses_centre(lx & 3, ly & 3)) {
            // 21 window rows x 4 dword columns of horizontal sums: lane t takes entry t, lanes 0..19 also entry 64 + t
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int t = lane + 64 * k;
                if (t < 84) *(uint2 *)(L.hrow + t * 2) = h4_raw<6>(yw, t >> 2, (ux0 & 3) + 2 + (t & 3) * 4);
            }
            wave_lds_fence();
            outY = qpel4<6>(yw, row + 2, (ux0 & 3) + 2 + dw * 4, lx & 3, ly & 3, L.hrow + lane * 2, 8);
        } else
            outY = qpel4<6>(yw, row + 2, (ux0 & 3) + 2 + dw * 4, lx & 3, ly & 3);
        if (lane < 32) outC = chroma4<3>(cw + cp * 27 + crow * 3, (ucx0 & 3) + cdw * 4, lx & 7, ly & 7);
#endif
    } else {
        // ======== general case: four 8x8 quadrants with their own vectors ========
        // Everything per quadrant stays in vector registers: a lane acts as (a) window loader of quadrant
        // wq = lane>>4, (b) producer of a luma dword in quadrant lq, (c) producer of a chroma dword in quadrant cq.
        const int lq = (row >> 3) * 2 + (dw >> 1), cq = (crow >> 2) * 2 + cdw;
        const int wq = lane >> 4, wl = lane & 15;
        // quadrant q covers the 4x4 blocks b0, b0+1, b0+4, b0+5 (raster), b0 = (q>>1)*8 + (q&1)*2; lanes 0..15 hold the vectors
        // (all three shuffles before any comparison: a shuffle inside a short-circuit branch would read lanes that are masked off)
        const int mv_r = __shfl(mvreg, lane + 1), mv_d = __shfl(mvreg, lane + 4), mv_rd = __shfl(mvreg, lane + 5);
        const bool quad_uni = (mvreg == mv_r) & (mvreg == mv_d) & (mvreg == mv_rd);
        const unsigned uni_mask = (unsigned)__ballot(quad_uni);                      // bit b0(q) is meaningful
        const int wb0 = (wq >> 1) * 8 + (wq & 1) * 2;
        const int mvW = __shfl(mvreg, wb0);
        int riW = (int)(int8_t)(refs4 >> (8 * wq));
        if (riW < 0 || riW >= n_ref) riW = 0;
        const uint8_t *refW = (const uint8_t *)glob((const uint64_t *)pd->ref)[riW];
        const int wx0 = X0 + (wq & 1) * 8 + (mv_x(mvW) >> 2) - 2, wy0 = Y0 + (wq >> 1) * 8 + (mv_y(mvW) >> 2) - 2;
        const int cx0 = X0 / 2 + (wq & 1) * 4 + (mv_x(mvW) >> 3), cy0 = Y0 / 2 + (wq >> 1) * 4 + (mv_y(mvW) >> 3);
        const bool fastW = (uni_mask >> wb0) & 1;
        const unsigned long long fast_lanes = __ballot(fastW);                       // 16 equal bits per quadrant
        // ---- all windows at once: luma 13 rows x 4 dwords per quadrant (lane: dword wl&3 of rows (wl>>2) + 4k),
        // chroma 5 rows x 2 dwords per quadrant and plane ----
        {
            // Every lane loads, unconditionally: lanes past the end of a window repeat its last row, quadrants that will take
            // the per-sample path fetch a window they do not use.  Predicated loads would cost exec-mask bookkeeping and, worse,
            // make every load wait for the previous one (the wait for the reference pointer would sit inside the predicate).
            uint32_t yv[4], cv[2];
            const int xa = (wx0 & ~3) + (wl & 3) * 4, xc = clip3i(xa, 0, g.w - 4);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int r = min((wl >> 2) + 4 * k, 12);
                yv[k] = WLOAD(refW + luma_off(g, xc, clip3i(wy0 + r, 0, g.h - 1)));
            }
            const int cr = min(wl, 9) >> 1, cxa = (cx0 & ~3) + (wl & 1) * 4;
            const uint32_t coff = chroma_off(g, 0, clip3i(cxa, 0, g.cw - 4), clip3i(cy0 + cr, 0, g.ch - 1));
            cv[0] = WLOAD(refW + coff); cv[1] = WLOAD(refW + coff + 64);
#pragma unroll
            for (int k = 0; k < 4; k++) yv[k] = edge_fix(yv[k], xa, g.w);
            cv[0] = edge_fix(cv[0], cxa, g.cw); cv[1] = edge_fix(cv[1], cxa, g.cw);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int r = (wl >> 2) + 4 * k;
                if (r < 13) L.ywin[wq][r * 4 + (wl & 3)] = yv[k];
            }
            if (wl < 10) { L.cwin[0][wq][wl] = cv[0]; L.cwin[1][wq][wl] = cv[1]; }
        }
        wave_lds_fence();
        // ---- one pass per distinct (vector, reference, path): the phase is wave-uniform inside a pass ----
        unsigned todo = 15;
#pragma unroll 1
        while (todo) {
            const int q0 = __ffs((int)todo) - 1;
            const int b0 = (q0 >> 1) * 8 + (q0 & 1) * 2;
            const int kmv = __builtin_amdgcn_readlane(mvreg, b0);
            const int kri = __builtin_amdgcn_readlane(riW, q0 * 16);
            const bool kfast = (fast_lanes >> (q0 * 16)) & 1;
            const unsigned long long same = __ballot(mvW == kmv && riW == kri && fastW == kfast);
            const unsigned group = (unsigned)(same & 1) | ((unsigned)(same >> 15) & 2) | ((unsigned)(same >> 30) & 4) | ((unsigned)(same >> 45) & 8);
            todo &= ~group;
            if (kfast) {
                const int lx = mv_x(kmv), ly = mv_y(kmv);
                if ((group >> lq) & 1)
                    outY = qpel4<4>(L.ywin[lq], (row & 7) + 2, ((X0 + (lx >> 2) - 2) & 3) + 2 + (dw & 1) * 4, lx & 3, ly & 3);
                if (lane < 32 && ((group >> cq) & 1))
                    outC = chroma4<2>(L.cwin[cp][cq] + (crow & 3) * 2, (X0 / 2 + (lx >> 3)) & 3, lx & 7, ly & 7);
            } else {
                // sub-8x8 partitions with differing vectors: every lane samples the clamped plane with its own vectors
                // (the branch is wave-uniform, so every lane takes part in these shuffles)
                const int mvl = __shfl(mvreg, (row >> 2) * 4 + dw);
                const int mvA = __shfl(mvreg, (crow >> 1) * 4 + cdw * 2), mvB = __shfl(mvreg, (crow >> 1) * 4 + cdw * 2 + 1);
                const uint8_t *rf = pd->ref[kri];
                if ((group >> lq) & 1) outY = slow_luma4(rf, g.w, g.h, g.mb_w, X0 + dw * 4, Y0 + row, mvl);
                if (lane < 32 && ((group >> cq) & 1)) outC = slow_chroma4(rf, g.cw, g.ch, g.mb_w, cp, X0 / 2 + cdw * 4, Y0 / 2 + crow, mvA, mvB);
            }
        }
    }

    // ---------------- residual (decoder/macroblock.c:832-890) ----------------
    if (mask && !EXP_NORESID) {
        if (mask & 0xffff) {                                  // luma: unscan + dequant
            const DqParams dq = dq_params(qp);
            int lb = lane >> 2;
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                int pos = zigzag_pos((lane & 3) * 4 + kk);
                int c = (int)(int16_t)((kk & 2 ? lc.y : lc.x) >> (16 * (kk & 1)));
                L.coef[lb * 16 + pos] = (int16_t)dequant_coef(c, pos, dq);
            }
        }
        if (cbp >> 4) {                                       // chroma: DC (core/dct.c:55-68, core/quant.c:138-159) + AC
            const int qpc = chroma_qp(clip3i(qp + ph.chroma_qp_offset, 0, 51));
            const DqParams dq = dq_params(qpc);
            int cb = 16 + (lane >> 3), i2 = lane & 7;
#pragma unroll
            for (int kk = 0; kk < 2; kk++) {                  // level index 2*i2+kk sits at scan position 2*i2+kk+1
                int k = 2 * i2 + kk + 1;
                if (k < 16) { int pos = zigzag_pos(k); L.coef[cb * 16 + pos] = (int16_t)dequant_coef((int)(int16_t)(cc >> (16 * kk)), pos, dq); }
            }
            // DC of chroma block j of plane p: lanes 0..7 hold the parsed DC levels of (p = lane>>2, index lane&3)
            const int cdc = (int)(int16_t)cdc_raw;
            int d0 = __shfl(cdc, (lane >> 5) * 4 + 0), d1 = __shfl(cdc, (lane >> 5) * 4 + 1);
            int d2 = __shfl(cdc, (lane >> 5) * 4 + 2), d3 = __shfl(cdc, (lane >> 5) * 4 + 3);
            if ((lane & 7) == 0) {
                int j = (lane >> 3) & 3;
                int t0 = d0 + d1, t1 = d0 - d1, t2 = d2 + d3, t3 = d2 - d3;
                int f = pick_addsub(t0, t1, t2, t3, j & 1, j & 2);            // {t0+t2, t1+t3, t0-t2, t1-t3}[j]
                f = (int)(int16_t)f;
                int qbits = dq.qbits - 1;                     // qpc/6 - 5
                int v = qbits >= 0 ? f * (int)((unsigned)dq.mf0 << qbits) : (f * dq.mf0) >> (-qbits);
                L.coef[cb * 16] = (int16_t)v;
            }
        }
        wave_lds_fence();
        const int blk = blk_at(dw, row >> 2);
        const bool resY = (mask >> blk) & 1, resC = lane < 32 && (cbp >> 4);
        int16_t *coefY = L.coef + blk * 16, *coefC = L.coef + (16 + cp * 4 + (crow >> 2) * 2 + cdw) * 16;
        if (resY) idct4x4_rowpass(coefY, row & 3);
        if (resC) idct4x4_rowpass(coefC, crow & 3);
        wave_lds_fence();
        if (resY) outY = idct4x4_colpass_add(outY, coefY, row & 3);
        if (resC) outC = idct4x4_colpass_add(outC, coefC, crow & 3);
    }

    // ---------------- the lane stores its own dword ----------------
    if (EXP_NOSTORE && outY != 0x12345678u) return;
    // (tiled frame: luma dword (row, dw) sits at lane*4, chroma dword (plane, row, dw) at 256 + lane*4 - the
    // macroblock goes out as three whole cache lines)
    uint8_t *tile = ph.dst + (size_t)mbi * MB_TILE;
    gstore1(tile + lane * 4, outY);
    if (lane < 32) gstore1(tile + MB_TILE_U + lane * 4, outC);
}


// ------------------------------------------------------------------------------------------
// K1b k_inter_quads - macroblocks with one vector per 8x8 quadrant (16x8, 8x16, P_8x8), quadrant by quadrant.
//
// In k_inter such a macroblock costs 2.5x a single-vector one: one interpolation pass per distinct vector with most lanes
// idle.  Here the work item is the QUADRANT (8x8 luma + two 4x4 chroma blocks with their residuals), and the host parser
// hands the quadrants over sorted by quarter-pel phase (p264hip_picture_t.quads): a wavefront takes four list entries -
// four quadrants of the same phase from whatever macroblocks - so the phase is wave-uniform again, every lane works in
// the one interpolation pass, and nothing is sorted on the device.  16 lanes per quadrant: lane l produces luma dword
// (row l>>1, dword l&1) and, for l < 8, chroma row l&3 of plane l>>2.
// Same arithmetic and the same helpers as k_inter; only the bookkeeping is per 16-lane group instead of per wavefront.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 8)
void k_inter_quads(const PicDev *__restrict__ pics, Geom g, uint32_t inv_mbw, int chunks_per_pic, int n_chunks, uint32_t inv_chunks)
{
    __shared__ InterLds lds[4];
    // XCD-aware remap as in k_inter: every XCD gets a contiguous range of (picture, chunk) pairs, so that the quadrants of
    // one picture - which reach the device sorted by phase, not by position - meet their reference tiles in ONE L2
    // (a 1080p reference frame is 3.1 MB, an XCD's L2 4 MB).  Without it the same tiles were fetched by all eight XCDs.
    const int per_xcd = gridDim.x >> 3;
    const int logical = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (logical >= n_chunks) return;
    int pic = (int)__umulhi((unsigned)logical, inv_chunks);
    if (logical - pic * chunks_per_pic >= chunks_per_pic) pic++;
    const int chunk = logical - pic * chunks_per_pic;
    const PicDev *pd = pics + pic;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n_quads = pd->n_quads;
    const int item0 = (chunk * 4 + wave) * 4;
    if (item0 >= n_quads) return;
    const PicHead ph = load_pic_head(pd);
    const int grp = lane >> 4, l = lane & 15;
    InterLds &L = lds[wave];

    // ---- the group's work item; padding entries repeat the wave's first (always real) entry and store nothing ----
    const uint32_t item_raw = glob(pd->quads)[item0 + grp];
    const bool valid = item_raw != 0xffffffffu;
    const uint32_t item = valid ? item_raw : glob(pd->quads)[item0];     // (a uniform load: the wave's first entry)
    const int mbi = (int)(item >> 2), q = (int)(item & 3);
    const uint4 rec = gload4(ph.mb + mbi);
    const int b0 = (q >> 1) * 8 + (q & 1) * 2;
    const int mv = glob(ph.mv)[mbi * 16 + b0];
    int ri = glob(ph.ref_idx)[mbi * 4 + q];
    if (ri < 0 || ri >= ph.n_ref) ri = 0;
    const uint8_t *ref = (const uint8_t *)glob((const uint64_t *)pd->ref)[ri];
    const unsigned mask = rec.y;
    const int qp = (rec.x >> 8) & 255, cbp = (rec.x >> 16) & 255;
    const AS1 int16_t *cf = glob(ph.coefs) + (size_t)rec.z * 16;
    int mby = (int)__umulhi((unsigned)mbi, inv_mbw);
    if (mbi - mby * g.mb_w >= g.mb_w) mby++;
    const int mbx = mbi - mby * g.mb_w;
    const int X0 = mbx * 16 + (q & 1) * 8, Y0 = mby * 16 + (q >> 1) * 8;        // quadrant origin (luma)

    // ---- coded coefficients of the quadrant: luma blocks 4q .. 4q+3 (lane: block l>>2, levels 4*(l&3)..+3), chroma blocks
    //      16+q and 20+q (lane: plane l>>3, levels 2*(l&7)..+1), the macroblock's 8 chroma DC levels in lanes 0..7 ----
    uint2 lc = make_uint2(0, 0); uint32_t cc = 0, cdc_raw = 0;
    const int lb = q * 4 + (l >> 2), cb = 16 + q + 4 * (l >> 3);
    if (mask) {
        if ((mask >> lb) & 1) lc = gload2(cf + coef_slot(mask, lb) * 16 + (l & 3) * 4);
        if ((mask >> cb) & 1) cc = gload1(cf + coef_slot(mask, cb) * 16 + (l & 7) * 2);
        if ((mask & P264_COEF_CHROMA_DC) && l < 8) cdc_raw = glob((const uint16_t *)cf)[((mask >> 24) & 1) * 16 + l];
    }

    // ---- reference windows: luma 13 rows x 4 dwords (lane: dword l&3 of rows (l>>2) + 4k), chroma 5 rows x 2 dwords per
    //      plane (lanes 0..9); every lane loads, lanes past the end repeat the last row ----
    const int wx0 = X0 + (mv_x(mv) >> 2) - 2, wy0 = Y0 + (mv_y(mv) >> 2) - 2;
    const int cx0 = X0 / 2 + (mv_x(mv) >> 3), cy0 = Y0 / 2 + (mv_y(mv) >> 3);
    {
        uint32_t yv[4], cv[2];
        const int xa = (wx0 & ~3) + (l & 3) * 4, cxa = (cx0 & ~3) + (l & 1) * 4, cr = min(l, 9) >> 1;
        const int wxa = wx0 & ~3, cxa4 = cx0 & ~3;
        const bool inside = ((wxa | (g.w - 16 - wxa)) | (wy0 | (g.h - 13 - wy0)) | (cxa4 | (g.cw - 8 - cxa4)) | (cy0 | (g.ch - 5 - cy0))) >= 0;
        if (__ballot(!inside) == 0) {                           // all four windows inside the picture: no clamps, no border fix-up
#pragma unroll
            for (int k = 0; k < 4; k++) yv[k] = WLOAD(ref + luma_off(g, xa, wy0 + min((l >> 2) + 4 * k, 12)));
            const uint32_t coff = chroma_off(g, 0, cxa, cy0 + cr);
            cv[0] = WLOAD(ref + coff); cv[1] = WLOAD(ref + coff + 64);
        } else {
            const int xc = clip3i(xa, 0, g.w - 4);
#pragma unroll
            for (int k = 0; k < 4; k++) yv[k] = WLOAD(ref + luma_off(g, xc, clip3i(wy0 + min((l >> 2) + 4 * k, 12), 0, g.h - 1)));
            const uint32_t coff = chroma_off(g, 0, clip3i(cxa, 0, g.cw - 4), clip3i(cy0 + cr, 0, g.ch - 1));
            cv[0] = WLOAD(ref + coff); cv[1] = WLOAD(ref + coff + 64);
#pragma unroll
            for (int k = 0; k < 4; k++) yv[k] = edge_fix(yv[k], xa, g.w);
            cv[0] = edge_fix(cv[0], cxa, g.cw); cv[1] = edge_fix(cv[1], cxa, g.cw);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int r = (l >> 2) + 4 * k;
            if (r < 13) L.ywin[grp][r * 4 + (l & 3)] = yv[k];
        }
        if (l < 10) { L.cwin[0][grp][l] = cv[0]; L.cwin[1][grp][l] = cv[1]; }
    }
    wave_lds_fence();

    // ---- prediction: the quarter-pel phase is the same for all four quadrants of the wave (the list is sorted by it) ----
    const int fx = rfl(mv_x(mv) & 3), fy = rfl(mv_y(mv) & 3);
    const int row = l >> 1, dw = l & 1;                         // luma: (row, dword) inside the 8x8 quadrant
    const int cp = (l >> 2) & 1, crow = l & 3;                  // chroma (lanes 0..7): plane, row of the 4x4 block
    uint32_t outY, outC = 0;
    if (phase_uses_centre(fx, fy)) {
        // per quadrant 13 window rows x 2 dword columns of horizontal sums: lane l takes entry l, lanes 0..9 also entry 16 + l
        uint32_t *hq = L.hrow + grp * 52;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int t = l + 16 * k;
            if (t < 26) *(uint2 *)(hq + t * 2) = h4_raw<4>(L.ywin[grp], t >> 1, (wx0 & 3) + 2 + (t & 1) * 4);
        }
        wave_lds_fence();
        outY = qpel4<4>(L.ywin[grp], row + 2, (wx0 & 3) + 2 + dw * 4, fx, fy, hq + l * 2, 4);
    } else
        outY = qpel4<4>(L.ywin[grp], row + 2, (wx0 & 3) + 2 + dw * 4, fx, fy);
    if (l < 8) outC = chroma4<2>(L.cwin[cp][grp] + crow * 2, cx0 & 3, mv_x(mv) & 7, mv_y(mv) & 7);

    // ---- residual (decoder/macroblock.c:832-890): six blocks per quadrant, L.coef[(6*grp + block)*16 + raster position] ----
    if (__ballot(mask != 0)) {
        int16_t *co = L.coef + grp * 96;
        // the four macroblocks of a wave nearly always share their QP: then the dequantisation parameters are scalars
        const int qp0 = rfl(qp);
        const bool qp_uniform = __ballot(qp != qp0) == 0;
        {
            const DqParams dq = qp_uniform ? dq_params(qp0) : dq_params(qp);
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const int pos = zigzag_pos((l & 3) * 4 + kk);
                const int c = (int)(int16_t)((kk & 2 ? lc.y : lc.x) >> (16 * (kk & 1)));
                co[(l >> 2) * 16 + pos] = (int16_t)dequant_coef(c, pos, dq);
            }
        }
        {   // chroma: DC (core/dct.c:55-68, core/quant.c:138-159) + AC of block q of both planes
            const int qpc0 = chroma_qp(clip3i(qp0 + ph.chroma_qp_offset, 0, 51));
            const DqParams dq = qp_uniform ? dq_params(qpc0) : dq_params(chroma_qp(clip3i(qp + ph.chroma_qp_offset, 0, 51)));
            const int pl = l >> 3, i2 = l & 7;
#pragma unroll
            for (int kk = 0; kk < 2; kk++) {                  // level index 2*i2+kk sits at scan position 2*i2+kk+1
                const int k = 2 * i2 + kk + 1;
                if (k < 16) { const int pos = zigzag_pos(k); co[(4 + pl) * 16 + pos] = (int16_t)dequant_coef((int)(int16_t)(cc >> (16 * kk)), pos, dq); }
            }
            // lanes 0..7 of the group hold the DC levels of (plane = lane>>2, index lane&3); all shuffles before any use
            const int cdc = (int)(int16_t)cdc_raw, base = (lane & 48) + pl * 4;
            const int d0 = __shfl(cdc, base), d1 = __shfl(cdc, base + 1), d2 = __shfl(cdc, base + 2), d3 = __shfl(cdc, base + 3);
            if (i2 == 0) {
                const int t0 = d0 + d1, t1 = d0 - d1, t2 = d2 + d3, t3 = d2 - d3;
                int f = pick_addsub(t0, t1, t2, t3, q & 1, q & 2);            // {t0+t2, t1+t3, t0-t2, t1-t3}[q]
                f = (int)(int16_t)f;
                const int qbits = dq.qbits - 1;               // qpc/6 - 5
                const int v = qbits >= 0 ? f * (int)((unsigned)dq.mf0 << qbits) : (f * dq.mf0) >> (-qbits);
                co[(4 + pl) * 16] = (int16_t)((cbp >> 4) ? v : 0);
            }
        }
        wave_lds_fence();
        const int blk = (row >> 2) * 2 + dw;                  // block inside the quadrant, decode order
        const bool resY = (mask >> (q * 4 + blk)) & 1, resC = l < 8 && (cbp >> 4);
        if (resY) idct4x4_rowpass(co + blk * 16, row & 3);
        if (resC) idct4x4_rowpass(co + (4 + cp) * 16, crow);
        wave_lds_fence();
        if (resY) outY = idct4x4_colpass_add(outY, co + blk * 16, row & 3);
        if (resC) outC = idct4x4_colpass_add(outC, co + (4 + cp) * 16, crow);
    }

    // ---- store into the macroblock's tile ----
    if (valid) {
        uint8_t *tile = ph.dst + (size_t)mbi * MB_TILE;
        gstore1(tile + ((q >> 1) * 8 + row) * 16 + (q & 1) * 8 + dw * 4, outY);
        if (l < 8) gstore1(tile + MB_TILE_U + cp * 64 + ((q >> 1) * 4 + crow) * 8 + (q & 1) * 4, outC);
    }
}
