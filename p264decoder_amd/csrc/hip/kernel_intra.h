// kernel_intra.h - K3: intra macroblock reconstruction (I pictures, and the intra MBs of P / B pictures).
//
// Replaces the intra half of p264_macroblock_decode (decoder/macroblock.c:769-831,851-890), the
// mode fix-ups valid_intra16x16_mode / valid_intra4x4_mode / valid_intra8x8c_mode
// (decoder/macroblock.c:635-753), the predictors of core/predict.c:55-638, idct4x4dc +
// p264_mb_dequant_4x4_dc (core/dct.c:104-136, core/quant.c:161-191) and add16x16_idct.
//
// Intra prediction reads the UNFILTERED reconstruction of the left / top / top-left / top-right
// neighbours, so macroblocks form a 2-MB-lag wavefront over the picture (wavefront_sync.h).  Per
// picture one workgroup for luma and one for chroma (independent chains).  A wavefront owns a BAND
// of four macroblock rows and works on FOUR MACROBLOCKS AT A TIME, sixteen lanes each (group g of
// the wavefront = row g of the band): the sixteen dependent 4x4 blocks of an Intra4x4 macroblock
// never had work for more than sixteen lanes, and with one macroblock per wavefront (rounds 1-2)
// the kernel was bound by the instructions and the memory round trip per macroblock, not by the
// dependencies.  Every iteration each group takes the next intra macroblock of its row if that
// macroblock's neighbours are final (the row above is this wavefront's own group g-1, or - for
// group 0 - the last row of the band above, published through an LDS counter): in an I picture the
// groups fall into the 2-MB staircase by themselves, in P pictures (sparse intra macroblocks, whose
// inter neighbours were finished by the motion-compensation kernels) they mostly run independently.
// A group never waits inside an iteration - the wavefront polls only when NO group can go - so a
// wavefront cannot block itself.  Missing neighbours are substituted in registers (128 /
// replicated t3) instead of being written into the frame as the reference does
// (decoder/macroblock.c:697-713, SURVEY A-Q7).
#pragma once
#include <stddef.h>
#include "device_common.h"
#include "wavefront_sync.h"
#include "kernel_deblock.h"             // (edge_info_of: the edge-info role of k_intra_sparse)

#define IT_STRIDE 24               // luma tile: row -1..15, byte 3 = left column, 4..19 = MB, 20..23 = top-right
#define CT_STRIDE 12               // chroma tile (one per plane): byte 3 = left column, 4..11 = MB

struct IntraGrp {                  // per group of sixteen lanes (= one macroblock in flight)
    uint8_t  tile[17 * IT_STRIDE]; // luma: the tile above; chroma: two tiles of 9 x CT_STRIDE
    int16_t  res[16 * 16];         // residual of every 4x4 block (luma: decode order; chroma: plane * 4 + block), raster inside
    int16_t  dc[16];
    uint8_t  pad[8];
};
struct IntraLds { IntraGrp g[4]; };                        // per wavefront
static_assert(sizeof(IntraGrp) % 16 == 0, "group areas stay 16-byte aligned");

__device__ __forceinline__ int f3(int a, int b, int c) { return (a + 2 * b + c + 2) >> 2; }
__device__ __forceinline__ int f2(int a, int b) { return (a + b + 1) >> 1; }

// Intra 4x4 prediction (core/predict.c:366-638) over an EDGE ARRAY: S[0..14] = l3 l3 l2 l1 l0 lt t0 .. t7 t7 (left
// column bottom-up, corner, top and top-right row; the ends replicated).  Every directional mode is then one of
//   copy S[c],  (S[c] + S[c+1] + 1) >> 1,  (S[c-1] + 2 S[c] + S[c+1] + 2) >> 2
// with an index c that is linear in (x,y) per mode (checked against the per-mode formulas for all modes and positions).
// The four macroblocks of a wavefront have four different modes: per (mode, sample) the places of S[c-1], S[c], S[c+1] come
// out of a 576-byte table in LDS, filled from this function when the kernel starts.  All three kinds are ONE formula over
// three places: (A + 2 B + D + 2) >> 2 is B for A = D = B and (B + D + 1) >> 1 for A = D - the table repeats a place
// instead of naming a kind, and the sixteen steps have no branch on it.
enum { P4_COPY = 0, P4_F2 = 1, P4_F3 = 2, P4_DC = 3 };
__device__ __forceinline__ void pred4x4_where(int mode, int x, int y, int &c, int &kind)
{
    switch (mode) {
    case 0: c = 6 + x; kind = P4_COPY; break;                                                   // vertical
    case 1: c = 4 - y; kind = P4_COPY; break;                                                   // horizontal
    case 2: c = 1; kind = P4_DC; break;
    case 3: c = 7 + x + y; kind = P4_F3; break;                                                 // diagonal down-left
    case 4: c = 5 + x - y; kind = P4_F3; break;                                                 // diagonal down-right
    case 5: { int z = 2 * x - y;                                                                // vertical-right
              c = z >= 0 ? 5 + x - (y >> 1) : z == -1 ? 5 : 6 - y; kind = (z >= 0 && !(z & 1)) ? P4_F2 : P4_F3; break; }
    case 6: { int z = 2 * y - x, i = y - (x >> 1);                                              // horizontal-down
              c = z >= 0 ? ((z & 1) ? 5 - i : 4 - i) : z == -1 ? 5 : 4 + x; kind = (z >= 0 && !(z & 1)) ? P4_F2 : P4_F3; break; }
    case 7: { int i = x + (y >> 1); c = (y & 1) ? 7 + i : 6 + i; kind = (y & 1) ? P4_F3 : P4_F2; break; }   // vertical-left
    default: { int z = x + 2 * y;                                                               // 8: horizontal-up
              c = z >= 5 ? 1 : 3 - y - (x >> 1); kind = z > 5 ? P4_COPY : (z == 5 || (z & 1)) ? P4_F3 : P4_F2; break; }
    }
}
#define INTRA_LUT_ENTRIES (9 * 16)         // uint32: place of A | place of B << 8 | place of D << 16 | DC << 24 (lut_entry)
#define INTRA_LUT_BIAS (IT_STRIDE + 1)     // places are offsets from the block origin + this: 0 (the corner) .. 4 IT_STRIDE
__device__ __forceinline__ int edge_offset(int k) { return k <= 4 ? (4 - max(k, 1)) * IT_STRIDE - 1 : -IT_STRIDE + min(k, 13) - 6; }
__device__ __forceinline__ uint32_t lut_entry(int mode, int x, int y)
{
    int c, kind;
    pred4x4_where(mode, x, y, c, kind);
    const int pb = edge_offset(c) + INTRA_LUT_BIAS, pd = edge_offset(c + 1) + INTRA_LUT_BIAS;
    const int pa = kind == P4_F3 ? edge_offset(c - 1) + INTRA_LUT_BIAS : kind == P4_F2 ? pd : pb;
    return (uint32_t)pa | (uint32_t)pb << 8 | (uint32_t)(kind == P4_COPY ? pb : pd) << 16 | (uint32_t)(kind == P4_DC) << 24;
}

// sum over the sixteen lanes of a group / over aligned groups of 2^k lanes (xor butterflies stay inside the group)
__device__ __forceinline__ int sum_lanes(int v, int n)
{
    for (int m = 1; m < n; m <<= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ int byte_sum(uint32_t v) { return (int)__builtin_amdgcn_sad_u8(v, 0u, 0u); }

// add4x4_idct (core/dct.c:205-247) without the prediction: r[y][0] = (d[y][0], d[y][1]), r[y][1] = (d[y][2], d[y][3]) as the
// reference's int16 d[][] = (sum + 32) >> 6; first pass in packed 16-bit (its int16 tmp), second pass in 32-bit
__device__ __forceinline__ void idct_res(const uint32_t (&col)[4][2], uint32_t (&r)[4][2])
{
    s16x2 T[4][2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const s16x2 c0 = as_s16x2(col[0][j]), c1 = as_s16x2(col[1][j]), c2 = as_s16x2(col[2][j]), c3 = as_s16x2(col[3][j]);
        const s16x2 s02 = c0 + c2, d02 = c0 - c2, s13 = c1 + (c3 >> 1), d13 = (c1 >> 1) - c3;
        T[0][j] = s02 + s13; T[1][j] = d02 + d13; T[2][j] = d02 - d13; T[3][j] = s02 - s13;
    }
    int res[4][4];
#pragma unroll
    for (int x = 0; x < 4; x++) {
        const int a0 = T[x][0].x, a1 = T[x][0].y, a2 = T[x][1].x, a3 = T[x][1].y;
        const int s02 = a0 + a2 + 32, d02 = a0 - a2 + 32, s13 = a1 + (a3 >> 1), d13 = (a1 >> 1) - a3;
        res[0][x] = (s02 + s13) >> 6; res[1][x] = (d02 + d13) >> 6; res[2][x] = (d02 - d13) >> 6; res[3][x] = (s02 - s13) >> 6;
    }
#pragma unroll
    for (int y = 0; y < 4; y++) {
        r[y][0] = ((uint32_t)res[y][0] & 0xffffu) | (uint32_t)res[y][1] << 16;
        r[y][1] = ((uint32_t)res[y][2] & 0xffffu) | (uint32_t)res[y][3] << 16;
    }
}
// four samples + four int16 residuals, clipped to bytes
__device__ __forceinline__ uint32_t add_res4(uint32_t px, uint32_t r01, uint32_t r23)
{
    const s16x2 zero = { 0, 0 }, top = { 255, 255 };
    // (saturating: a residual near the int16 limits must not wrap - the reference adds in int)
    s16x2 lo = __builtin_elementwise_add_sat(as_s16x2(perm(0u, px, 0x0c010c00u)), as_s16x2(r01)), hi = __builtin_elementwise_add_sat(as_s16x2(perm(0u, px, 0x0c030c02u)), as_s16x2(r23));
    lo = __builtin_elementwise_min(__builtin_elementwise_max(lo, zero), top);
    hi = __builtin_elementwise_min(__builtin_elementwise_max(hi, zero), top);
    return perm(as_u32(hi), as_u32(lo), 0x06040200u);
}

// ------------------------------------------------------------------------------------------
// luma of one intra macroblock per group of sixteen lanes (l = lane inside the group).  Every global load the macroblock
// needs is issued at the top, before anything waits: one memory round trip per iteration, everything after that runs out
// of registers and LDS.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void intra_luma4(const PicDev *pd, const Geom &g, IntraGrp &L, const uint32_t *lut, int mbx, int mby, const uint4 rec, int l)
{
    // Everything below that depends on the lane number alone (roles, tile offsets, edge slots) would otherwise be hoisted
    // out of the caller's loops and kept - or spilled - across them: recomputing it per macroblock is a few dozen
    // instructions, holding it is dozens of registers.
    asm volatile("" : "+v"(l));
    const int mbi = mby * g.mb_w + mbx, X0 = mbx * 16, Y0 = mby * 16;
    const int mb_type = (int)(rec.x & 255u), qp = (int)((rec.x >> 8) & 255u), modes = (int)(rec.x >> 24), avail = (int)(rec.w & 255u);
    const bool aL = avail & P264_AVAIL_LEFT, aT = avail & P264_AVAIL_TOP, aTR = avail & P264_AVAIL_TOPRIGHT, aTL = avail & P264_AVAIL_TOPLEFT;
    uint8_t *F = pd->dst;                                  // strip frame layout (device_common.h)
    const AS1 uint8_t *Fg = glob(F);
    const unsigned mask = rec.y;
    const int16_t *cf = pd->coefs + (size_t)rec.z * 16;
    const bool is16 = mb_type == P264_MB_I16x16;

    // ---- (a) neighbour samples, 128 where the neighbour does not exist: lane l the left sample of row l; lanes 0..3 the
    //      four dwords of the row above, lane 4 the top-right dword, lane 5 the corner ----
    int vL = 128;
    if (aL) vL = Fg[luma_off(g, X0 - 1, Y0 + l)];
    uint32_t vT = 0x80808080u;
    if (l < 4)       { if (aT)  vT = gload1(F + luma_off(g, X0 + 4 * l, Y0 - 1)); }
    else if (l == 4) { if (aTR) vT = gload1(F + luma_off(g, X0 + 16, Y0 - 1)); }
    else if (l == 5) { if (aTL) vT = Fg[luma_off(g, X0 - 1, Y0 - 1)]; }
    // ---- (b) levels: lane l = block l (decode order), all 16 slots of it ----
    const bool coded = (mask >> l) & 1;
    uint4 la = make_uint4(0, 0, 0, 0), lb = la;
    if (coded) { const int16_t *c = cf + coef_slot(mask, l) * 16; la = gload4(c); lb = gload4(c + 8); }
    const bool dcflag = is16 && (mask & P264_COEF_LUMA_DC);
    int ldc = 0;                                           // Intra16x16 DC level l (scan order)
    if (dcflag) ldc = glob(cf)[l];
    // ---- (c) Intra4x4 prediction mode of block l ----
    int modebyte = 0;
    if (!is16) modebyte = glob(pd->i4modes)[mbi * 16 + l];

    // ---- land the neighbours ----
    L.tile[(l + 1) * IT_STRIDE + 3] = (uint8_t)vL;
    if (l < 5) *(uint32_t *)(L.tile + 4 + 4 * l) = vT;
    else if (l == 5) L.tile[3] = (uint8_t)vT;
    wave_lds_fence();
    if (!aTR && l == 4) *(uint32_t *)(L.tile + 20) = (uint32_t)L.tile[19] * 0x01010101u;   // top-right of the MB missing: replicate t15 (:706-709)
    wave_lds_fence();

    // ---- residuals of all sixteen blocks into LDS (lane = block) ----
    const bool grp_res = (mask & 0xffffu) != 0 || dcflag;  // (the same for the sixteen lanes)
    if (__ballot(grp_res)) {
        int dcv = 0;
        if (__ballot(dcflag)) {
            // luma DC: unscan, idct4x4dc (core/dct.c:104-136), rounded dequant (core/quant.c:161-191)
            if (dcflag) L.dc[zigzag_pos(l)] = (int16_t)ldc;
            wave_lds_fence();
            int v = 0;
            if (dcflag) {
                const int i = l >> 2, j = l & 3;
                int tcol[4];
#pragma unroll
                for (int c = 0; c < 4; c++) {                                   // column pass for tmp[i][c]
                    const int d0 = L.dc[c], d1 = L.dc[4 + c], d2 = L.dc[8 + c], d3 = L.dc[12 + c];
                    const int s01 = d0 + d1, d01 = d0 - d1, s23 = d2 + d3, d23 = d2 - d3;
                    tcol[c] = (int)(int16_t)pick_addsub(s01, d01, s23, d23, i >= 2, i == 1 || i == 2);   // {s01+s23, s01-s23, d01-d23, d01+d23}[i]
                }
                const int s01 = tcol[0] + tcol[1], d01 = tcol[0] - tcol[1], s23 = tcol[2] + tcol[3], d23 = tcol[2] - tcol[3];
                v = (int)(int16_t)pick_addsub(s01, d01, s23, d23, j >= 2, j == 1 || j == 2);
                const int per = (qp * 43) >> 8, rem = qp - per * 6, qbits = per - 6, mf = dq_scale(0, rem);
                v = qbits >= 0 ? v * (int)((unsigned)mf << qbits) : (v * mf + (1 << (-qbits - 1))) >> (-qbits);
                v = (int)(int16_t)v;
            }
            wave_lds_fence();
            if (dcflag) L.dc[l] = (int16_t)v;                                   // raster (y*4+x) of the 4x4 block grid
            wave_lds_fence();
            if (dcflag) dcv = L.dc[blk_y(l) * 4 + blk_x(l)];
        }
        const uint32_t lv[8] = { la.x, la.y, la.z, la.w, lb.x, lb.y, lb.z, lb.w };
        uint32_t col[4][2], r[4][2];
        // Intra4x4: 16 levels; Intra16x16: 15 AC levels at scan positions 1..15 (:787-794).  The macroblocks of a wavefront are
        // mostly of one type (the lists of P / B pictures are): both orders only where they are mixed
        if (!__ballot(is16)) unscan_cols<false>(lv, col);
        else if (!__ballot(!is16)) unscan_cols<true>(lv, col);
        else {
            uint32_t c4[4][2], c16[4][2];
            unscan_cols<false>(lv, c4);
            unscan_cols<true>(lv, c16);
#pragma unroll
            for (int x = 0; x < 4; x++) { col[x][0] = is16 ? c16[x][0] : c4[x][0]; col[x][1] = is16 ? c16[x][1] : c4[x][1]; }
        }
        dequant_cols(col, qp);
        if (is16) col[0][0] = (col[0][0] & 0xffff0000u) | ((uint32_t)dcv & 0xffffu);
        idct_res(col, r);
        const bool mine = coded || dcflag;
        uint4 *out = (uint4 *)(L.res + l * 16);
        out[0] = mine ? make_uint4(r[0][0], r[0][1], r[1][0], r[1][1]) : make_uint4(0, 0, 0, 0);
        out[1] = mine ? make_uint4(r[2][0], r[2][1], r[3][0], r[3][1]) : make_uint4(0, 0, 0, 0);
        wave_lds_fence();
    }

    if (__ballot(is16)) {
        if (is16) {
            // ---- Intra16x16 (core/predict.c:55-193): lane l = row l ----
            int mode = modes & 3;
            if (mode == 2) mode = aTL ? 2 : aL ? 4 : aT ? 5 : 6;                // :635-667
            const uint32_t *trow = (const uint32_t *)(L.tile + 4);
            const uint32_t t0 = trow[0], t1 = trow[1], t2 = trow[2], t3 = trow[3];
            uint32_t pv[4];
            const int sumT = byte_sum(t0) + byte_sum(t1) + byte_sum(t2) + byte_sum(t3), sumL = sum_lanes(vL, 16);
            const int dc = mode == 2 ? (sumT + sumL + 16) >> 5 : mode == 4 ? (sumL + 8) >> 4 : mode == 5 ? (sumT + 8) >> 4 : 128;
            const uint32_t flat = (uint32_t)(mode == 1 ? vL : dc) * 0x01010101u;
            pv[0] = mode == 0 ? t0 : flat; pv[1] = mode == 0 ? t1 : flat; pv[2] = mode == 0 ? t2 : flat; pv[3] = mode == 0 ? t3 : flat;
            if (__ballot(mode == 3)) {                                           // plane, core/predict.c:159-193
                // H = sum (i+1) (top[8+i] - top[6-i]) = sum over x = -1..15 of (x - 7) top[x]; the same down the left column
                const int corner = L.tile[3], topl = L.tile[4 + l];
                const int H = sum_lanes((l - 7) * topl, 16) - 8 * corner, V = sum_lanes((l - 7) * vL, 16) - 8 * corner;
                const int a = 16 * (__shfl(vL, (int)(threadIdx.x & 48) + 15) + (int)L.tile[19]), b = (5 * H + 32) >> 6, c = (5 * V + 32) >> 6;
                const int i00 = a - 7 * b - 7 * c + 16 + c * l;
                if (mode == 3) {
                    // (round_pack4: written as clip255(v >> 5) | ... << 8 | ... the compiler picks v_ashr_pk_u8_i32 itself and ORs the
                    // other bytes onto its result, whose upper half is not zero for negative inputs)
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int t[4] = { i00 + b * (4 * k), i00 + b * (4 * k + 1), i00 + b * (4 * k + 2), i00 + b * (4 * k + 3) };
                        pv[k] = round_pack4<5>(t);
                    }
                }
            }
            if (grp_res) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint2 rr = *(const uint2 *)(L.res + blk_at(k, l >> 2) * 16 + (l & 3) * 4);
                    pv[k] = add_res4(pv[k], rr.x, rr.y);
                }
            }
            gstore4(F + mb_luma_off(g, mbx, mby) + l * 16, make_uint4(pv[0], pv[1], pv[2], pv[3]));
        }
    }
    if (__ballot(!is16)) {
        if (!is16) {
            // ---- Intra4x4: sixteen dependent blocks (decoder/macroblock.c:799-831), lane l = sample (x, y) of the current block.
            // The samples a prediction reads are the edge S = l3 l3 l2 l1 l0 lt t0..t7 t7 around the block (pred4x4_where); they
            // are read straight out of the tile: cells of neighbours that do not exist already hold their substitute (128, put
            // there with the macroblock's neighbours; decoder/macroblock.c:697-713), and where a block has a row above but no
            // top-right neighbour (blocks 3, 7, 11, 13, 15 - and 5 without the macroblock's top-right, done above) the last
            // sample of the block above is replicated into the four cells to its right by the step that produced it: those cells
            // are padding, or belong to a block that is decoded later and overwrites them.  One LDS round trip per step. ----
            const int x = l & 3, y = l >> 2;
            // which of the 16 blocks (decode order) have their left / top neighbour (for the DC fall-backs): one bit per block
            const unsigned left_m = 0xFAFAu | (aL ? 0x0505u : 0u), top_m = 0xFFCCu | (aT ? 0x0033u : 0u);
            const int grp_base = (int)(threadIdx.x & 48);
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int bx = blk_x(i), by = blk_y(i);
                const bool left = (left_m >> i) & 1, top = (top_m >> i) & 1;
                const int mode = __shfl(modebyte, grp_base + i);
                uint8_t *o = L.tile + (by * 4 + 1) * IT_STRIDE + 4 + bx * 4;           // block origin inside the tile
                // the lane's three edge samples: their places come out of the table (pred4x4_where / lut_entry)
                const uint32_t wk = lut[min(mode, 8) * 16 + l];
                const uint8_t *e = o - INTRA_LUT_BIAS;
                const int a = e[wk & 255u], b = e[(wk >> 8) & 255u], d = e[(wk >> 16) & 255u];
                int v = (a + 2 * b + d + 2) >> 2;
                const bool is_dc = (wk >> 24) != 0;
                if (__ballot(is_dc)) {                                                  // DC and its fall-backs, :677-695
                    const int sl = o[-1] + o[IT_STRIDE - 1] + o[2 * IT_STRIDE - 1] + o[3 * IT_STRIDE - 1], st = byte_sum(*(const uint32_t *)(o - IT_STRIDE));
                    const int dcv = (left && top) ? (sl + st + 4) >> 3 : left ? (sl + 2) >> 2 : top ? (st + 2) >> 2 : 128;
                    v = is_dc ? dcv : v;
                }
                if ((mask >> i) & 1) v = clip255(v + (int)L.res[i * 16 + l]);
                o[y * IT_STRIDE + x] = (uint8_t)v;
                // the block below has no top-right neighbour: its t4..t7 = this block's last sample
                if ((i == 1 || i == 5 || i == 7 || i == 9 || i == 13) && l == 15) *(uint32_t *)(o + 3 * IT_STRIDE + 4) = (uint32_t)v * 0x01010101u;
                wave_lds_fence();
            }
            const uint32_t *row = (const uint32_t *)(L.tile + (l + 1) * IT_STRIDE + 4);
            gstore4(F + mb_luma_off(g, mbx, mby) + l * 16, make_uint4(row[0], row[1], row[2], row[3]));
        }
    }
}

// ------------------------------------------------------------------------------------------
// chroma of one intra macroblock per group of sixteen lanes: lane l = row r = l & 7 of plane p = l >> 3
// (core/predict.c:199-361, decoder/macroblock.c:721-753,851-890)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void intra_chroma4(const PicDev *pd, const Geom &g, IntraGrp &L, int mbx, int mby, const uint4 rec, int l)
{
    asm volatile("" : "+v"(l));                            // (as in intra_luma4)
    const int X0 = mbx * 8, Y0 = mby * 8;
    const int qp = (int)((rec.x >> 8) & 255u), cbp = (int)((rec.x >> 16) & 255u), modes = (int)(rec.x >> 24), avail = (int)(rec.w & 255u);
    const bool aL = avail & P264_AVAIL_LEFT, aT = avail & P264_AVAIL_TOP, aTL = avail & P264_AVAIL_TOPLEFT;
    uint8_t *F = pd->dst;
    const AS1 uint8_t *Fg = glob(F);
    const unsigned mask = rec.y;
    const int16_t *cf = pd->coefs + (size_t)rec.z * 16;
    const int p = l >> 3, r = l & 7;
    uint8_t *tile = L.tile + p * (9 * CT_STRIDE);

    // ---- neighbours: lane (p, r) the left sample of its row; r = 0, 1 the two dwords of the row above, r = 2 the corner ----
    int vL = 128;
    if (aL) vL = Fg[chroma_off(g, p, X0 - 1, Y0 + r)];
    uint32_t vT = 0x80808080u;
    if (r < 2)       { if (aT)  vT = gload1(F + chroma_off(g, p, X0 + 4 * r, Y0 - 1)); }
    else if (r == 2) { if (aTL) vT = Fg[chroma_off(g, p, X0 - 1, Y0 - 1)]; }
    // ---- levels: lanes 0..7 = block (plane l >> 2, block l & 3): 15 AC levels; the plane's four DC levels ----
    const bool has_chroma = (cbp >> 4) != 0;
    const int blk = 16 + l;                                // (l < 8)
    const bool coded = has_chroma && l < 8 && ((mask >> blk) & 1);
    uint4 la = make_uint4(0, 0, 0, 0), lb = la; uint2 dcl = make_uint2(0, 0);
    if (coded) { const int16_t *c = cf + coef_slot(mask, blk) * 16; la = gload4(c); lb = gload4(c + 8); }
    if (has_chroma && l < 8 && (mask & P264_COEF_CHROMA_DC)) dcl = gload2(cf + ((mask >> 24) & 1) * 16 + (l >> 2) * 4);

    tile[(r + 1) * CT_STRIDE + 3] = (uint8_t)vL;
    if (r < 2) *(uint32_t *)(tile + 4 + 4 * r) = vT;
    else if (r == 2) tile[3] = (uint8_t)vT;
    wave_lds_fence();

    // ---- residuals of the eight blocks (same arithmetic as kernel_mc.h) ----
    if (__ballot(has_chroma)) {
        if (l < 8) {
            const int qpc = chroma_qp(clip3i(qp + pd->chroma_qp_offset, 0, 51));
            const uint32_t lv[8] = { la.x, la.y, la.z, la.w, lb.x, lb.y, lb.z, lb.w };
            uint32_t col[4][2], rr[4][2];
            unscan_cols<true>(lv, col);
            dequant_cols(col, qpc);
            // DC of block j: idct2x2dc (core/dct.c:55-68, int16 stores), then p264_mb_dequant_2x2_dc (core/quant.c:138-159)
            const int j = l & 3;
            const int d0 = (int)(int16_t)(dcl.x & 0xffff), d1 = (int)dcl.x >> 16, d2 = (int)(int16_t)(dcl.y & 0xffff), d3 = (int)dcl.y >> 16;
            const int t0 = d0 + d1, t1 = d0 - d1, t2 = d2 + d3, t3 = d2 - d3;
            int f = pick_addsub(t0, t1, t2, t3, j & 1, j & 2);                    // {t0+t2, t1+t3, t0-t2, t1-t3}[j]
            f = (int)(int16_t)f;
            const int per = (qpc * 43) >> 8, rem = qpc - per * 6;
            const int dc = (f * (int)(dq_s(0, rem) << per)) >> 1;
            col[0][0] = (col[0][0] & 0xffff0000u) | ((uint32_t)dc & 0xffffu);
            idct_res(col, rr);
            uint4 *out = (uint4 *)(L.res + l * 16);
            out[0] = has_chroma ? make_uint4(rr[0][0], rr[0][1], rr[1][0], rr[1][1]) : make_uint4(0, 0, 0, 0);
            out[1] = has_chroma ? make_uint4(rr[2][0], rr[2][1], rr[3][0], rr[3][1]) : make_uint4(0, 0, 0, 0);
        }
        wave_lds_fence();
    }

    // ---- prediction: 8 samples of row r ----
    int mode = (modes >> 4) & 3;
    if (mode == 0) mode = aTL ? 0 : aL ? 4 : aT ? 5 : 6;                         // :721-753
    const uint32_t ta = *(const uint32_t *)(tile + 4), tb = *(const uint32_t *)(tile + 8);
    const int s0 = byte_sum(ta), s1 = byte_sum(tb);
    const int sq = sum_lanes(vL, 4);                                             // left sum of this lane's half (rows 0..3 or 4..7)
    const int so = __shfl_xor(sq, 4);
    const int s2 = r < 4 ? sq : so, s3 = r < 4 ? so : sq;
    uint32_t pa, pb;                                                             // samples 0..3 and 4..7
    {
        const int up = r < 4;
        int da, db;                                                              // DC of the left and right 4x4 of this row
        if (mode == 0)      { da = up ? (s0 + s2 + 4) >> 3 : (s3 + 2) >> 2; db = up ? (s1 + 2) >> 2 : (s1 + s3 + 4) >> 3; }
        else if (mode == 4) { da = db = ((up ? s2 : s3) + 2) >> 2; }
        else if (mode == 5) { da = (s0 + 2) >> 2; db = (s1 + 2) >> 2; }
        else                { da = db = 128; }
        if (mode == 1) da = db = vL;
        pa = (uint32_t)da * 0x01010101u; pb = (uint32_t)db * 0x01010101u;
        if (mode == 2) { pa = ta; pb = tb; }
    }
    if (__ballot(mode == 3)) {
        // H = sum (i+1) (top[4+i] - top[2-i]) = sum over x = -1..7 of (x - 3) top[x]; the same down the left column
        const int corner = tile[3], topl = tile[4 + r];
        const int H = sum_lanes((r - 3) * topl, 8) - 4 * corner, V = sum_lanes((r - 3) * vL, 8) - 4 * corner;
        const int a = 16 * (__shfl(vL, (int)(threadIdx.x & 56) + 7) + (int)tile[11]), b = (17 * H + 16) >> 5, c = (17 * V + 16) >> 5;
        const int i00 = a - 3 * b - 3 * c + 16 + c * r;
        if (mode == 3) {
            const int ta4[4] = { i00, i00 + b, i00 + 2 * b, i00 + 3 * b }, tb4[4] = { i00 + 4 * b, i00 + 5 * b, i00 + 6 * b, i00 + 7 * b };
            pa = round_pack4<5>(ta4); pb = round_pack4<5>(tb4);
        }
    }
    if (has_chroma) {
        const int16_t *rs = L.res + (p * 4 + (r >> 2) * 2) * 16 + (r & 3) * 4;
        const uint2 ra = *(const uint2 *)rs, rb = *(const uint2 *)(rs + 16);
        pa = add_res4(pa, ra.x, ra.y); pb = add_res4(pb, rb.x, rb.y);
    }
    gstore2(F + mb_chroma_off(g, mbx, mby) + r * 16 + p * 8, make_uint2(pa, pb));
}

#ifndef INTRA_ROW_WAVES
#define INTRA_ROW_WAVES 16          // most wavefronts per picture workgroup
#endif
#ifndef INTRA_WAVES_PER_EU
#define INTRA_WAVES_PER_EU 8        // (64 registers.  Round 3's kernel spilled at that and was built for 5 wavefronts per SIMD - 96 registers; since
                                    //  the band walk's state is scalar it needs 75 and fits 64 without scratch: config 2 at 5 / 6 / 7 / 8 wavefronts
                                    //  per SIMD 235 / 234 / 242 / 254 k frames/s, scratch/r4_intraocc.sh)
#endif
#define INTRA_BAND 4                // macroblock rows per wavefront = groups of sixteen lanes
// P / B pictures: intra macroblocks without an intra neighbour to the left or above depend on nothing this kernel writes.
// They are collected first (two lists, by macroblock type, so that the four macroblocks of an iteration run the same code)
// and reconstructed four at a time in any order; only the rest goes through the ordered band walk below.
#ifndef INTRA_ROUNDS
#define INTRA_ROUNDS     2          // rounds of ready macroblocks before the ordered band walk takes what is left (measured on the
                                    // bench stream: 1 / 2 / 3 and more rounds 0.46 / 0.33 / 0.37 ms per launch)
#endif
#define INTRA_FREE_CAP   256        // entries per list (uint16 macroblock index); macroblocks beyond it stay in the band walk
#define INTRA_MASKS      160        // pictures of up to this many row windows (rows x windows of 64 macroblocks): 1080p has 136
struct IntraSync { int progress[MAX_MB_ROWS / INTRA_BAND + 1]; };            // per band: columns of its last row that are final
struct IntraShared {
    IntraSync sync;
    uint32_t lut[INTRA_LUT_ENTRIES];
    uint16_t free_list[2][INTRA_FREE_CAP];
    unsigned long long m_intra[INTRA_MASKS], m_walk[INTRA_MASKS];   // per row window: intra macroblocks / those left to the band walk
    int free_n[2];
};
__device__ __forceinline__ void intra_picture(IntraShared &sh, const PicDev *__restrict__ pics, const Geom &g, int *status, const uint8_t *__restrict__ is_intra_all,
                                              const int pic_index, const bool chroma_role)
{
    IntraSync &sync = sh.sync;
    uint32_t *lut = sh.lut;
    uint16_t (*free_list)[INTRA_FREE_CAP] = sh.free_list;
    unsigned long long *m_intra = sh.m_intra, *m_walk = sh.m_walk;
    int *free_n = sh.free_n;
    // one tile set per wavefront, sized by the launch (dynamic shared memory = wavefronts x sizeof(IntraLds))
    extern __shared__ __attribute__((aligned(16))) uint8_t intra_dyn_lds[];
    IntraLds *lds = (IntraLds *)intra_dyn_lds;
    const PicDev *pd = pics + pic_index;                    // (luma and chroma of a picture in separate workgroups)
    // (the wavefront's number through readfirstlane: everything the band walk derives from it - rows, windows, masks, the
    // progress it waits for - is then scalar work for the compiler instead of vector instructions on uniform values)
    const int wave = rfl((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63, grp = lane >> 4, l = lane & 15;
    const int n_waves = blockDim.x >> 6;                       // 16 per picture, or fewer when pictures share a CU (host's choice)
    const int n_bands = (g.mb_h + INTRA_BAND - 1) / INTRA_BAND;
    for (int i = threadIdx.x; i < INTRA_LUT_ENTRIES; i += blockDim.x) lut[i] = lut_entry(i >> 4, i & 3, (i >> 2) & 3);
    for (int i = threadIdx.x; i <= n_bands; i += blockDim.x) sync.progress[i] = 0;
    const int wins = (g.mb_w + 63) / 64, n_win = g.mb_h * wins;
    const bool use_free = pd->slice_type != P264_SLICE_I && n_win <= INTRA_MASKS && g.n_mb < 65536;     // (scalar)
    const uint8_t *is_intra = is_intra_all + (size_t)pic_index * g.n_mb;                                 // written by k_mc_sort[_b] for P / B pictures
    if (threadIdx.x < 2) free_n[threadIdx.x] = 0;
    __syncthreads();
    IntraGrp &L = lds[wave].g[grp];
    bool ok = true;
    if (use_free) {
        // ---- collect, pass 1: which macroblocks are intra (one byte per lane, four row windows per pass: the loads fly together) ----
        for (int w0 = wave * 4; w0 < n_win; w0 += n_waves * 4) {
            bool in[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int w = w0 + j, row = w / wins, x = (w - row * wins) * 64 + lane;
                // (one byte per macroblock from the work-list sort, which has seen every record anyway: 64 contiguous bytes per
                // wavefront here - out of the 16-byte records it was one byte per cache line sector, 130 KB per picture and role)
                in[j] = w < n_win && x < g.mb_w && glob(is_intra)[row * g.mb_w + x] != 0;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) { const unsigned long long m = __ballot(in[j]); if (w0 + j < n_win && lane == 0) m_intra[w0 + j] = m; }
        }
        __syncthreads();
        // ---- rounds: the macroblocks of m_intra are PENDING.  Ready = no pending macroblock to the left, above-left, above,
        //      above-right (mask arithmetic); the ready ones go to the list of their type (what the lists cannot take stays
        //      pending), are reconstructed four per iteration in any order, and leave the pending set.  Round 0 takes every
        //      macroblock without an intra neighbour (85 % in the bench stream), the next rounds peel the small clusters layer by
        //      layer; what is still pending after INTRA_ROUNDS rounds (large intra areas) is left to the ordered band walk. ----
        const uint32_t inv_mbw = 0xffffffffu / (uint32_t)g.mb_w;
#pragma unroll 1
        for (int round = 0; round < INTRA_ROUNDS; round++) {
            for (int w = wave; w < n_win; w += n_waves) {
                const int row = w / wins, win = w - row * wins;
                const unsigned long long m = m_intra[w];
                if (m == 0) { if (lane == 0) m_walk[w] = 0; continue; }
                unsigned long long blocked = m << 1;
                if (win > 0) blocked |= m_intra[w - 1] >> 63;
                if (row > 0) {
                    const unsigned long long u = m_intra[w - wins];
                    blocked |= u | u << 1 | u >> 1;
                    if (win > 0) blocked |= m_intra[w - wins - 1] >> 63;
                    if (win + 1 < wins) blocked |= m_intra[w - wins + 1] << 63;
                }
                const unsigned long long fr = m & ~blocked;
                unsigned long long taken = 0;
                if (fr) {
                    const int mbi = row * g.mb_w + win * 64 + lane;
                    const bool mine = (fr >> lane) & 1;
                    int type = 0;
                    if (mine) type = glob(pd->mb)[mbi].mb_type;
#pragma unroll
                    for (int t = 0; t < 2; t++) {
                        const bool me = mine && (type == P264_MB_I16x16) == (t == 1);
                        const unsigned long long mt = __ballot(me);
                        if (mt == 0) continue;
                        int at = 0;
                        if (lane == 0) at = atomicAdd(&free_n[t], __popcll(mt));
                        at = rfl(at) + __popcll(mt & ((1ull << lane) - 1ull));
                        const bool put = me && at < INTRA_FREE_CAP;
                        if (put) free_list[t][at] = (uint16_t)mbi;
                        taken |= __ballot(put);
                    }
                }
                if (lane == 0) m_walk[w] = taken;              // (this round's share; the pending set changes behind the barrier)
            }
            __syncthreads();
            const int n0 = min(free_n[0], INTRA_FREE_CAP), n1 = min(free_n[1], INTRA_FREE_CAP);
            if (n0 + n1 == 0) break;                           // (scalar: nothing was ready - nothing is pending)
            // ---- reconstruct the listed macroblocks; the records of the next four are requested a round trip ahead ----
#pragma unroll 1
            for (int t = 0; t < 2; t++) {
                const int n = t ? n1 : n0;
                int k = wave * 4;
                int mbi = k + grp < n ? (int)free_list[t][k + grp] : 0;
                uint4 rec = gload4(pd->mb + mbi);
                while (k < n) {
                    const bool active = k + grp < n;
                    const int kn = k + n_waves * 4;
                    const int mbi_n = kn + grp < n ? (int)free_list[t][kn + grp] : 0;
                    const uint4 rec_n = gload4(pd->mb + mbi_n);
                    int mby = (int)__umulhi((unsigned)mbi, inv_mbw);
                    if (mbi - mby * g.mb_w >= g.mb_w) mby++;
                    const int mbx = mbi - mby * g.mb_w;
                    if (active) {
                        if (chroma_role) intra_chroma4(pd, g, L, mbx, mby, rec, l);
                        else             intra_luma4(pd, g, L, lut, mbx, mby, rec, l);
                    }
                    k = kn; mbi = mbi_n; rec = rec_n;
                }
            }
            // their samples are neighbours of what is still pending: stores drained, then the sets updated
            __syncthreads();
            // (thread number hidden from the optimiser: it keeps &m_intra[threadIdx.x] alive from the top of the kernel otherwise -
            // one register too many for the 64 this build has)
            int tid_a = (int)threadIdx.x;
            asm volatile("" : "+v"(tid_a));
            for (int w = tid_a; w < n_win; w += blockDim.x) m_intra[w] &= ~m_walk[w];
            if (threadIdx.x < 2) free_n[threadIdx.x] = 0;
            __syncthreads();
        }
        int tid_b = (int)threadIdx.x;
        asm volatile("" : "+v"(tid_b));
        for (int w = tid_b; w < n_win; w += blockDim.x) m_walk[w] = m_intra[w];             // the band walk's share
        __syncthreads();
    }
    for (int band = wave; band < n_bands; band += n_waves) {
        const int R0 = band * INTRA_BAND;
        const bool feeds = R0 + INTRA_BAND < g.mb_h;          // a band below reads this band's last row
        // Per row of the band: a window of 64 macroblocks (one record per lane), which of them are intra (todo) and which of
        // those touch an intra macroblock of the row above (deps: only those can still be in flight there - inter MBs were
        // finished by the motion-compensation kernels).  Every row moves through its windows on its own.
        unsigned long long todo[INTRA_BAND], deps[INTRA_BAND];
        int base[INTRA_BAND];
        bool fin[INTRA_BAND];
#pragma unroll
        for (int r = 0; r < INTRA_BAND; r++) { base[r] = -64; todo[r] = deps[r] = 0; fin[r] = R0 + r >= g.mb_h; }
        int spins = 0, published = -1;
        for (;;) {
            // (uniform by construction; said here so that the compiler keeps the walk's state in scalar registers)
#pragma unroll
            for (int r = 0; r < INTRA_BAND; r++) { todo[r] = rfl64(todo[r]); deps[r] = rfl64(deps[r]); base[r] = rfl(base[r]); fin[r] = rfl((int)fin[r]) != 0; }
            spins = rfl(spins); published = rfl(published); ok = rfl((int)ok) != 0;
            // ---- rows whose window is used up take the next one (all of them in one batch of loads) ----
            for (;;) {
                bool need[INTRA_BAND], any_need = false;
#pragma unroll
                for (int r = 0; r < INTRA_BAND; r++) {
                    need[r] = !fin[r] && todo[r] == 0;
                    if (need[r]) { base[r] += 64; if (base[r] >= g.mb_w) { fin[r] = true; need[r] = false; } }
                    any_need |= need[r];
                }
                if (!any_need) break;
                if (use_free) {
                    // (sparse pictures: the masks are in LDS; a macroblock waits only for macroblocks of the band walk above it)
#pragma unroll
                    for (int r = 0; r < INTRA_BAND; r++) {
                        if (!need[r]) continue;
                        const int row = R0 + r, win = base[r] >> 6, w = row * wins + win;
                        todo[r] = rfl64(m_walk[w]);                    // (LDS reads land in vector registers: back to scalars)
                        unsigned long long d = 0;
                        if (row > 0) {
                            const unsigned long long u = rfl64(m_walk[w - wins]);
                            d = u | u << 1 | u >> 1;
                            if (win > 0) d |= rfl64(m_walk[w - wins - 1]) >> 63;
                            if (win + 1 < wins) d |= rfl64(m_walk[w - wins + 1]) << 63;
                        }
                        deps[r] = d;
                    }
                    continue;
                }
                bool intra[INTRA_BAND], dep[INTRA_BAND];
#pragma unroll
                for (int r = 0; r < INTRA_BAND; r++) {
                    const int row = R0 + r, x = base[r] + lane;
                    intra[r] = dep[r] = false;
                    if (need[r] && x < g.mb_w) {
                        const AS1 p264hip_mb_t *cur = glob(pd->mb) + row * g.mb_w;
                        intra[r] = P264_MB_IS_INTRA(cur[x].mb_type);
                        if (row > 0) {
                            const AS1 p264hip_mb_t *up = cur - g.mb_w;
                            dep[r] = (int)P264_MB_IS_INTRA(up[x].mb_type) | (int)P264_MB_IS_INTRA(up[max(x - 1, 0)].mb_type) |
                                     (int)P264_MB_IS_INTRA(up[min(x + 1, g.mb_w - 1)].mb_type);
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < INTRA_BAND; r++) if (need[r]) { todo[r] = __ballot(intra[r]); deps[r] = __ballot(dep[r]); }
            }
            // next macroblock of every row (the row's end when it has none left): everything in front of it is final
            int col[INTRA_BAND];
#pragma unroll
            for (int r = 0; r < INTRA_BAND; r++) col[r] = fin[r] ? g.mb_w : base[r] + __ffsll((long long)todo[r]) - 1;
            // their records: wave-uniform addresses, read-only data - scalar loads, no vector registers
            uint4 srec[INTRA_BAND];
#pragma unroll
            for (int r = 0; r < INTRA_BAND; r++) {
                const int idx = (R0 + r) * g.mb_w + col[r];
                srec[r] = make_uint4(0, 0, 0, 0);
                if (!fin[r]) {
                    const __attribute__((address_space(4))) uint32_t *w = (const __attribute__((address_space(4))) uint32_t *)(pd->mb + rfl(idx));
                    srec[r] = make_uint4(w[0], w[1], w[2], w[3]);
                }
            }
            // the band's last row feeds the band below (release: this wavefront's stores have landed)
            if (feeds && col[INTRA_BAND - 1] != published) { published = col[INTRA_BAND - 1]; __hip_atomic_store(&sync.progress[band], published, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
            if (fin[0] && fin[1] && fin[2] && fin[3]) break;
            // whose neighbours are final?  The row above is this wavefront's own previous row, or the band above (acquire)
            bool go[INTRA_BAND], any = false;
            int above = band > 0 ? __hip_atomic_load(&sync.progress[band - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) : g.mb_w;
            above = rfl(above);
#pragma unroll
            for (int r = 0; r < INTRA_BAND; r++) {
                const bool dep = !fin[r] && ((deps[r] >> (col[r] - base[r])) & 1);
                go[r] = !fin[r] && (!dep || !ok || min(col[r] + 2, g.mb_w) <= above);
                any |= go[r];
                above = col[r];
            }
            if (!any) {                                        // (only ever waits for another wavefront: the band above)
                __builtin_amdgcn_s_sleep(WAIT_SLEEP);
                if (++spins > SPIN_LIMIT) { if (lane == 0) atomicOr(status, 1); ok = false; }
                continue;
            }
            // this group's macroblock
            // (selected with lane masks: written as grp == 0 ? a : grp == 1 ? b : ... the compiler builds a table in scratch memory)
            uint32_t m1 = 0u - (uint32_t)(grp == 1), m2 = 0u - (uint32_t)(grp == 2), m3 = 0u - (uint32_t)(grp == 3);
            asm volatile("" : "+v"(m1), "+v"(m2), "+v"(m3));
            const uint32_t m0 = ~(m1 | m2 | m3);
            auto pick = [&](uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return (a & m0) | (b & m1) | (c & m2) | (d & m3); };
            const bool active = pick(go[0], go[1], go[2], go[3]) != 0;
            const int mbx = (int)pick((uint32_t)col[0], (uint32_t)col[1], (uint32_t)col[2], (uint32_t)col[3]);
            const uint4 rec = make_uint4(pick(srec[0].x, srec[1].x, srec[2].x, srec[3].x), pick(srec[0].y, srec[1].y, srec[2].y, srec[3].y),
                                         pick(srec[0].z, srec[1].z, srec[2].z, srec[3].z), pick(srec[0].w, srec[1].w, srec[2].w, srec[3].w));
            if (active) {
                if (chroma_role) intra_chroma4(pd, g, L, mbx, R0 + grp, rec, l);
                else             intra_luma4(pd, g, L, lut, mbx, R0 + grp, rec, l);
            }
#pragma unroll
            for (int r = 0; r < INTRA_BAND; r++) if (go[r]) todo[r] &= todo[r] - 1;
        }
    }
}
// Two kernels of the same code, both at 64 registers (8 wavefronts per SIMD) since round 4: k_intra for batches with I pictures,
// k_intra_sparse - with the edge-info role below - for batches of P / B pictures only.
__global__ __launch_bounds__(INTRA_ROW_WAVES * 64, INTRA_WAVES_PER_EU)
void k_intra(const PicDev *__restrict__ pics, Geom g, int *status, const uint8_t *__restrict__ is_intra)
{
    __shared__ IntraShared sh;
    intra_picture(sh, pics, g, status, is_intra, (int)blockIdx.x, blockIdx.y != 0);
}
#ifndef INTRA_SPARSE_WAVES_PER_EU
#define INTRA_SPARSE_WAVES_PER_EU 8
#endif
// The sparse build also carries the loop filter's EDGE INFO pass (kernel_deblock.h, K4a) as a third role: that pass is bound
// by memory (84 bytes of records and vectors per macroblock in, 16 out) and depends on nothing but the parsed input, this
// kernel is bound by vector-instruction issue and leaves the memory pipes idle - side by side in ONE launch they overlap
// (two streams did not: the events between them cost more than the overlap).  Workgroup w of the
// launch: picture w / roles, role w % roles = luma, chroma, then INTRA_BS_WGS edge-info workgroups - a picture's
// workgroups are neighbours in the dispatch order, so the roles are in flight together all through the launch.
// Measured at 2048 pictures per launch (scratch/r4_bsfused2.sh, three runs each): own launch 0.59 + (0.37 + 4.78) ms for
// k_intra_sparse + (k_deblock_bs + k_deblock), fused 0.87 + 4.72 ms - 186.1 k -> 189.1 k frames/s; 2 / 4 / 8 edge-info
// workgroups per picture: 5.76 / 5.57 / 5.56 ms for the pair against 5.46 with one.
#define INTRA_BS_WGS 1
__global__ __launch_bounds__(INTRA_ROW_WAVES * 64, INTRA_SPARSE_WAVES_PER_EU)
void k_intra_sparse(const PicDev *__restrict__ pics, Geom g, int *status, const uint8_t *__restrict__ is_intra, EdgeInfo *__restrict__ info, uint32_t inv_mbw, int bs_wgs)
{
    const int roles = 2 + bs_wgs;
    const int pic = rfl((int)(blockIdx.x / (unsigned)roles)), role = (int)blockIdx.x - pic * roles;      // (the division runs on the vector unit: back to a scalar)
    if (role >= 2) {
        const PicDev *pd = pics + pic;
        if (!pd->deblock) return;
        // (stores through a buffer descriptor of the picture's edge-info array: 32-bit offsets, no 64-bit address per lane)
        const rsrc_t info_rs = make_rsrc(info + (size_t)pic * g.n_mb, (uint32_t)g.n_mb * (uint32_t)sizeof(EdgeInfo));
        // Round 6: with ONE edge-info workgroup per picture (the default) a thread walks a COLUMN of macroblocks downwards - lanes are
        // neighbouring columns of one row: the loads stay whole cache lines, the left neighbour still comes out of the lane below
        // (DPP) - and what edge_info_of needs of the macroblock above (record, bottom row of its vectors, reference indices) is what
        // the thread itself held one row earlier: no load.  As three loads per macroblock of lines another thread had fetched 120
        // macroblocks earlier they were 96 bytes of 32-byte sectors per macroblock that often had to come from memory again (the
        // launch streams ~ 4 TB/s through 32 MB of L2): 3.49 GB for 1.75 GB of algorithmic bytes in round 5.  The workgroup's threads
        // cover blockDim / pitch segments of rows (pitch = the picture's width rounded up to wavefronts); a segment's first row loads.
        const int pitch = (g.mb_w + 63) & ~63;
        if (bs_wgs == 1 && pitch <= (int)blockDim.x) {
            const int segs = (int)blockDim.x / pitch, seg = rfl((int)threadIdx.x / pitch), rows = (g.mb_h + segs - 1) / segs;      // (a wavefront lies inside one segment: scalars)
            int col = (int)threadIdx.x - seg * pitch;
            asm volatile("" : "+v"(col));
            const int y0 = seg * rows, y1 = min(y0 + rows, g.mb_h);
            const bool mine = seg < segs && col < g.mb_w;
            // (the carried words live in the launch's dynamic LDS - the intra roles' tiles, which this workgroup does not use: 32 bytes
            // per thread {record word 0, coded-block mask, reference indices, -} {bottom row of vectors}; in registers they cost this
            // 64-register build five spills)
            extern __shared__ __attribute__((aligned(16))) uint8_t intra_dyn_lds[];
            uint4 *carry = (uint4 *)intra_dyn_lds + 2 * threadIdx.x;
            const int *mvs = pd->mv;
            if (mine && y0 < y1) {
                uint4 r = make_uint4(0, 0, 0, 0), m = r; uint32_t rf = 0;
                if (y0 > 0) {
                    const uint32_t ti = (uint32_t)((y0 - 1) * g.mb_w + col);
                    r = gload4(ubase(pd->mb, ti * 16u)); m = gload4(ubase(mvs, ti * 64u + 48u)); rf = gload1(ubase(pd->ref_idx, ti * 4u));
                }
                carry[0] = make_uint4(r.x, r.y, rf, 0u); carry[1] = m;
            }
            for (int y = y0; y < y1; y++) {                    // (uniform trip count per wavefront: a wavefront lies inside one segment)
                if (!mine) continue;
                // (column and LDS slot derived from the thread number again per row: two instructions, no register held across the loop -
                // the build has 64)
                int tid = (int)threadIdx.x;
                asm volatile("" : "+v"(tid));
                const int col = tid - seg * pitch;
                uint4 *carry = (uint4 *)intra_dyn_lds + 2 * tid;
                int mbi = y * g.mb_w + col;
                asm volatile("" : "+v"(mbi));
                const uint4 rec = gload4(ubase(pd->mb, (uint32_t)mbi * 16u));
                const uint4 m0 = gload4(ubase(mvs, (uint32_t)mbi * 64u)), m1 = gload4(ubase(mvs, (uint32_t)mbi * 64u + 16u)), m2 = gload4(ubase(mvs, (uint32_t)mbi * 64u + 32u)), m3 = gload4(ubase(mvs, (uint32_t)mbi * 64u + 48u));
                const uint32_t refs = gload1(ubase(pd->ref_idx, (uint32_t)mbi * 4u));
                const uint4 c0 = carry[0];
                const EdgeTop top = { c0.x, c0.y, c0.z, carry[1] };
                carry[0] = make_uint4(rec.x, rec.y, refs, 0u); carry[1] = m3;
                const uint4 ei = edge_info_of<false>(pd, g, mbi, col, y, rec, m0, m1, m2, m3, refs, nullptr, &top);
                // (the store's offset from the thread number once more: kept alive across edge_info_of it was the build's one spill)
                int tid2 = (int)threadIdx.x;
                asm volatile("" : "+v"(tid2));
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{ ei.x, ei.y, ei.z, ei.w }, info_rs, (y * g.mb_w + tid2 - seg * pitch) * 16, 0, 0);
            }
            return;
        }
        for (int mbi_it = (role - 2) * (int)blockDim.x + (int)threadIdx.x; mbi_it < g.n_mb; mbi_it += bs_wgs * (int)blockDim.x) {
            // (every address below is a scalar base + this 32-bit index: hidden from the loop optimiser, which otherwise turns each
            // into a 64-bit induction variable per lane - this build has 64 registers)
            int mbi = mbi_it;
            asm volatile("" : "+v"(mbi));
            int mby = (int)__umulhi((unsigned)mbi, inv_mbw);
            if (mbi - mby * g.mb_w >= g.mb_w) mby++;
            const int mbx = mbi - mby * g.mb_w;
            const uint4 rec = gload4(ubase(pd->mb, (uint32_t)mbi * 16u));
            const int *mvs = pd->mv;
            const uint4 m0 = gload4(ubase(mvs, (uint32_t)mbi * 64u)), m1 = gload4(ubase(mvs, (uint32_t)mbi * 64u + 16u)), m2 = gload4(ubase(mvs, (uint32_t)mbi * 64u + 32u)), m3 = gload4(ubase(mvs, (uint32_t)mbi * 64u + 48u));
            const uint32_t refs = gload1(ubase(pd->ref_idx, (uint32_t)mbi * 4u));
            const uint4 ei = edge_info_of<false>(pd, g, mbi, mbx, mby, rec, m0, m1, m2, m3, refs, nullptr);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{ ei.x, ei.y, ei.z, ei.w }, info_rs, (int)((uint32_t)mbi_it * 16u), 0, 0);
        }
        return;
    }
    __shared__ IntraShared sh;
    intra_picture(sh, pics, g, status, is_intra, pic, role != 0);
}
