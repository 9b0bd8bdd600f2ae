// device_common.h - shared device-side definitions for the gfx950 reconstruction kernels.
//
// All arithmetic here is 8/16/32-bit integer; the kernels are HBM/latency bound, there is
// no dense contraction anywhere on this path (no MFMA on purpose).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "p264hip.h"

#define WAVE 64

// One entry per picture of a batch; built on the host for every reconstruct call.  Wave-uniform in every kernel
// (a wavefront never mixes pictures), so the fields are fetched with scalar loads.
struct PicDev {
    const p264hip_mb_t *mb;
    const int          *mv;        // packed (mvy << 16) | (mvx & 0xffff) per 4x4 block, [mb][16]
    const int8_t       *ref_idx;   // [mb][4]
    const uint8_t      *i4modes;   // [mb][16]
    const int16_t      *coefs;     // [blocks][16]
    uint8_t            *dst;       // destination frame (strip layout, see below)
    uint8_t            *store;     // the stream's frame store: slot 0; every reference and dst lie inside [store, store + store_bytes)
    uint32_t store_bytes, dst_off; // dst - store
    int32_t n_ref, slice_type, chroma_qp_offset, deblock, alpha_off, beta_off;
    uint32_t ref_off[P264HIP_MAX_REFS];   // reference frame k - store, list-0 order (entries >= n_ref repeat entry 0)
    // B pictures (slice_type == P264_SLICE_B) only:
    const int          *mv_l1;     // packed like mv, [mb][16]
    const int8_t       *ref_idx_l1;// [mb][4]; a quadrant predicts from list X iff its list-X index is >= 0
    const int16_t      *bipred_w;  // [16][16] weight of the list-0 prediction (used when weighted != 0)
    int32_t n_ref_l1, weighted;
    uint32_t ref_off_l1[P264HIP_MAX_REFS];
};

// Geometry shared by every picture of a context.
struct Geom {
    int mb_w, mb_h, n_mb;
    int w, h, cw, ch;              // luma / chroma plane sizes in samples
    uint32_t ystrip, cstrip, coff; // frame layout below: bytes per luma strip, per chroma strip, offset of the chroma part
};

// ---- frame layout in HBM -------------------------------------------------------------------
// Frames are stored as vertical STRIPS, one macroblock wide, each strip contiguous from the top of the picture to the
// bottom with a row pitch of 16 bytes:
//   luma   sample (x,y)  at  (x >> 4) * ystrip + y * 16 + (x & 15)                      ystrip = 16 * h
//   chroma sample (x,y)  at  coff + (x >> 3) * cstrip + y * 16 + plane * 8 + (x & 7)    cstrip = 16 * ch, coff = w * h
// (a chroma row holds 8 bytes of U then 8 bytes of V).  A macroblock is still whole cache lines - 256 contiguous luma
// bytes, 128 contiguous chroma bytes - which is what the two row-wavefront kernels need (a planar frame costs one line
// per 16-byte row piece: measured 3-5x HBM write amplification), and consecutive ROWS of a motion-compensation window
// are exactly 16 bytes apart wherever the window lies: its loads are one register offset per dword column plus an
// immediate per row, no per-row address arithmetic (kernel_mc.h).  Planar views exist only at the host boundary
// (p264hip_read_frame / p264hip_write_frame).
#define MB_LUMA_BYTES   256
// strip index x strip size: both below 2^24 (a strip is 16 bytes x the picture height), so the full-rate 24-bit multiply
// does it (v_mul_lo_u32 runs at a quarter of the rate and sits in every address computation of the kernels)
__device__ __forceinline__ uint32_t strip_mul(int strip, uint32_t strip_bytes)  { return __umul24((unsigned)strip, strip_bytes); }
#define MB_CHROMA_BYTES 128
__device__ __forceinline__ uint32_t luma_off(const Geom &g, int x, int y)
{
    return strip_mul(x >> 4, g.ystrip) + (uint32_t)(y * 16 + (x & 15));
}
__device__ __forceinline__ uint32_t chroma_off(const Geom &g, int plane, int x, int y)
{
    return g.coff + strip_mul(x >> 3, g.cstrip) + (uint32_t)(y * 16 + plane * 8 + (x & 7));
}
// first byte of a macroblock's luma / chroma part
__device__ __forceinline__ uint32_t mb_luma_off(const Geom &g, int mbx, int mby) { return strip_mul(mbx, g.ystrip) + (uint32_t)mby * MB_LUMA_BYTES; }
__device__ __forceinline__ uint32_t mb_chroma_off(const Geom &g, int mbx, int mby) { return g.coff + strip_mul(mbx, g.cstrip) + (uint32_t)mby * MB_CHROMA_BYTES; }

// ---- global-memory accessors ------------------------------------------------------------
// Pointers that reach a kernel through memory (the fields of PicDev) are "flat" to the compiler:
// it cannot prove they are not LDS or scratch, emits flat_load/flat_store, and has to wait for
// vmcnt AND lgkmcnt on every use, which ties the global and LDS pipelines together.  Everything
// those pointers address is HBM, so all accesses go through address-space-1 views.
#define AS1 __attribute__((address_space(1)))
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <class T> __device__ __forceinline__ AS1 T *glob(T *p) { return (AS1 T *)p; }
__device__ __forceinline__ uint4 gload4(const void *p) { u32x4 v = *(const AS1 u32x4 *)p; return make_uint4(v.x, v.y, v.z, v.w); }
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ uint3 gload3(const void *p) { u32x3 v = *(const AS1 u32x3 *)p; return make_uint3(v.x, v.y, v.z); }
__device__ __forceinline__ uint2 gload2(const void *p) { u32x2 v = *(const AS1 u32x2 *)p; return make_uint2(v.x, v.y); }
__device__ __forceinline__ uint2 gload2(const AS1 void *p) { u32x2 v = *(const AS1 u32x2 *)p; return make_uint2(v.x, v.y); }
__device__ __forceinline__ uint32_t gload1(const void *p) { return *(const AS1 uint32_t *)p; }
__device__ __forceinline__ uint32_t gload1(const AS1 void *p) { return *(const AS1 uint32_t *)p; }
__device__ __forceinline__ void gstore4(void *p, uint4 v) { u32x4 t = { v.x, v.y, v.z, v.w }; *(AS1 u32x4 *)p = t; }
__device__ __forceinline__ void gstore2(void *p, uint2 v) { u32x2 t = { v.x, v.y }; *(AS1 u32x2 *)p = t; }
__device__ __forceinline__ void gstore1(void *p, uint32_t v) { *(AS1 uint32_t *)p = v; }

__device__ __forceinline__ int clip3i(int v, int lo, int hi) { return min(max(v, lo), hi); }
__device__ __forceinline__ int clip255(int v) { return min(max(v, 0), 255); }

// LDS traffic between the lanes of ONE wave needs no barrier in hardware (a wave's DS
// operations execute in order); this only stops the compiler from moving them.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// v_perm_b32: result byte i = byte sel[i] of {hi (4..7), lo (0..3)}; 0x0c = zero
__device__ __forceinline__ uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
// v_sat_pk_u8_i16: both signed 16-bit halves of v saturated to 0..255, into bytes 0 (low half) and 1 (high half); the upper
// bytes are not used by any caller (the compiler has no pattern for this instruction)
__device__ __forceinline__ uint32_t sat_pk_u8_i16(uint32_t v) { uint32_t r; asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(v)); return r; }
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned long long rfl64(unsigned long long v)
{
    return (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) | (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32;
}

// ---- tables (H.264 standard data; the reference holds them at the cited places) --------
// zig-zag scan -> raster position, decoder/macroblock.c:602-603
__device__ __constant__ const uint8_t c_zigzag[16] = { 0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15 };
// core/frame.c:262-291
__device__ __constant__ const uint8_t c_alpha[52] = {
    0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,4,4,5,6,7,8,9,10,12,13,15,17,20,22,
    25,28,32,36,40,45,50,56,63,71,80,90,101,113,127,144,162,182,203,226,255,255 };
__device__ __constant__ const uint8_t c_beta[52] = {
    0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,2,2,2,3,3,3,3,4,4,4,6,6,7,7,
    8,8,9,9,10,10,11,11,12,12,13,13,14,14,15,15,16,16,17,17,18,18 };
__device__ __constant__ const uint8_t c_tc0[52][3] = {
    {0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},
    {0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,1},{0,0,1},{0,0,1},{0,0,1},{0,1,1},{0,1,1},{1,1,1},
    {1,1,1},{1,1,1},{1,1,1},{1,1,2},{1,1,2},{1,1,2},{1,1,2},{1,2,3},{1,2,3},{2,2,3},{2,2,4},{2,3,4},
    {2,3,4},{3,3,5},{3,4,6},{3,4,6},{4,5,7},{4,5,8},{4,6,9},{5,7,10},{6,8,11},{6,8,13},{7,10,14},{8,11,16},
    {9,12,18},{10,13,20},{11,15,23},{13,17,25} };

// table-free forms for the hot kernels (no dependent constant-memory loads):
// zig-zag scan position k -> raster position (same data as c_zigzag, one nibble per entry)
__device__ __forceinline__ int zigzag_pos(int k) { return (int)(((k & 8) ? 0xFEB7ADC9u : 0x63258410u) >> (4 * (k & 7))) & 15; }

// dequantisation parameters of one macroblock / plane: wave-uniform, so they live in SGPRs
struct DqParams { int mf0, mf1, mf2, qbits; };
// table-free: the six scales of a position class are 5-bit fields of one constant (scale / 16 <= 29), so a
// macroblock's parameters cost a handful of scalar ALU operations instead of dependent constant-memory loads
__device__ __forceinline__ int dq_scale(int cls, int rem)
{   // core/set.c:27-35: {10,11,13,14,16,18}, {13,14,16,18,20,23}, {16,18,20,23,25,29}
    const uint32_t k = cls == 0 ? (10u | 11u << 5 | 13u << 10 | 14u << 15 | 16u << 20 | 18u << 25)
                     : cls == 1 ? (13u | 14u << 5 | 16u << 10 | 18u << 15 | 20u << 20 | 23u << 25)
                                : (16u | 18u << 5 | 20u << 10 | 23u << 15 | 25u << 20 | 29u << 25);
    return (int)((k >> (5 * rem)) & 31u) * 16;
}
__device__ __forceinline__ DqParams dq_params(int qp)
{
    const int per = (qp * 43) >> 8, rem = qp - per * 6;          // qp / 6 and qp % 6 for 0 <= qp < 64
    DqParams d = { dq_scale(0, rem), dq_scale(1, rem), dq_scale(2, rem), per - 4 };
    return d;
}
// chroma QP of a luma QP index 0..51 (core/macroblock.h:210-218 = H.264 table 8-15): identity below 30, then
// qp minus a 4-bit correction
__device__ __forceinline__ int chroma_qp(int qi)
{
    const int k = qi - 30;
    if (k < 0) return qi;
    const int corr = k < 16 ? (int)((0x7765544332221111ull >> (4 * k)) & 15) : (int)((0xCBA998u >> (4 * (k - 16))) & 15);
    return qi - corr;
}
__device__ __forceinline__ int dequant_coef(int c, int pos, const DqParams &d)
{   // core/quant.c:66-99; position class (pos&1) + ((pos>>2)&1); int16 store wrap = A-Q8
    int cls = (pos & 1) + ((pos >> 2) & 1);
    const int m1 = -(int)(cls == 1), m2 = -(int)(cls == 2);            // lane-varying class: masks, not branches
    int mf = (d.mf0 & ~(m1 | m2)) | (d.mf1 & m1) | (d.mf2 & m2);
    int v = c * mf;
    v = d.qbits >= 0 ? (int)((unsigned)v << d.qbits) : (v + (1 << (-d.qbits - 1))) >> (-d.qbits);
    return (int)(int16_t)v;
}

// luma 4x4 block index (decode order, core/macroblock.h:194-201) <-> position
__device__ __forceinline__ int blk_x(int i) { return (i & 1) | ((i >> 1) & 2); }
__device__ __forceinline__ int blk_y(int i) { return ((i >> 1) & 1) | ((i >> 2) & 2); }
__device__ __forceinline__ int blk_at(int x, int y) { return (x & 1) | ((y & 1) << 1) | ((x & 2) << 1) | ((y & 2) << 2); }

// index of packed block `bit` (0..23) of an MB inside the coefficient stream (see p264hip.h)
__device__ __forceinline__ uint32_t coef_slot(uint32_t mask, int blk)
{
    return ((mask >> 24) & 1) + ((mask >> 25) & 1) + __popc(mask & ((1u << blk) - 1u) & 0xffffffu);
}

// ---- dequantisation, one coefficient (core/quant.c:66-99; int16 store wrap = A-Q8) -------
__device__ __forceinline__ int dequant_coef(int c, int pos, int qp)
{
    const int per = (qp * 43) >> 8, rem = qp - per * 6;
    int mf = dq_scale((pos & 1) + ((pos >> 2) & 1), rem);
    int qbits = per - 4;
    int v = c * mf;
    v = qbits >= 0 ? (int)((unsigned)v << qbits) : (v + (1 << (-qbits - 1))) >> (-qbits);
    return (int)(int16_t)v;
}

// a + b or a - b, with the pair picked by lane-varying flags (all-ones masks), branch-free
__device__ __forceinline__ int pick_addsub(int a0, int a1, int b0, int b1, bool second, bool minus)
{
    const int ms = -(int)second, neg = -(int)minus;
    const int a = (a1 & ms) | (a0 & ~ms), b = (b1 & ms) | (b0 & ~ms);
    return a + ((b ^ neg) - neg);
}

// Output k (0..3) of the transform's butterfly {s02+s13, d02+d13, d02-d13, s02-s13} with k varying per lane, without
// branches: nested ?: on a lane-varying index compiles to nested exec-mask regions (~16 scalar instructions per use).
__device__ __forceinline__ int butterfly_pick(int s02, int d02, int s13, int d13, int k)
{
    const int outer = -(int)(k == 0 || k == 3), neg = -(int)(k >= 2);     // all-ones masks
    const int a = (s02 & outer) | (d02 & ~outer), b = (s13 & outer) | (d13 & ~outer);
    return a + ((b ^ neg) - neg);
}

// The 4x4 inverse transform (core/dct.c:205-247), shared between the four lanes that own the four rows of a block: lane y
// runs the horizontal pass on ROW y only and puts it back in place (int16, as the reference stores its tmp), then - after
// a fence - every lane reads the 16 intermediate values and evaluates the vertical pass for its own output row.  (Each
// lane doing the whole horizontal pass for itself cost 4x the work of that pass.)
__device__ __forceinline__ void idct4x4_rowpass(int16_t *c, int y)
{
    uint2 *r = (uint2 *)(c + y * 4);
    const uint2 v = *r;
    const int c0 = (int)(int16_t)(v.x & 0xffff), c1 = (int)v.x >> 16, c2 = (int)(int16_t)(v.y & 0xffff), c3 = (int)v.y >> 16;
    const int s02 = c0 + c2, d02 = c0 - c2, s13 = c1 + (c3 >> 1), d13 = (c1 >> 1) - c3;
    const uint32_t t0 = (uint32_t)(s02 + s13) & 0xffffu, t1 = (uint32_t)(d02 + d13) << 16;
    const uint32_t t2 = (uint32_t)(d02 - d13) & 0xffffu, t3 = (uint32_t)(s02 - s13) << 16;
    *r = make_uint2(t0 | t1, t2 | t3);
}
// one output sample (x,y) of the vertical pass over the intermediate left by idct4x4_rowpass
__device__ __forceinline__ int idct4x4_col_sample(const int16_t *t, int x, int y)
{
    const int t0 = t[x], t1 = t[4 + x], t2 = t[8 + x], t3 = t[12 + x];
    return (int)(int16_t)((butterfly_pick(t0 + t2, t0 - t2, t1 + (t3 >> 1), (t1 >> 1) - t3, y) + 32) >> 6);
}

