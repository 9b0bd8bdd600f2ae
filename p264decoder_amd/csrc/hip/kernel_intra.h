// kernel_intra.h - K3: intra macroblock reconstruction (I pictures, and the intra MBs of P pictures).
//
// Replaces the intra half of p264_macroblock_decode (decoder/macroblock.c:769-831,851-890), the
// mode fix-ups valid_intra16x16_mode / valid_intra4x4_mode / valid_intra8x8c_mode
// (decoder/macroblock.c:635-753), the predictors of core/predict.c:55-638, idct4x4dc +
// p264_mb_dequant_4x4_dc (core/dct.c:104-136, core/quant.c:161-191) and add16x16_idct.
//
// Intra prediction reads the UNFILTERED reconstruction of the left / top / top-left / top-right
// neighbours, so macroblocks form a 2-MB-lag wavefront over the picture (wavefront_sync.h): per
// picture one workgroup for luma and one for chroma (independent chains: the kernel is bound by the
// latency of the macroblock-to-macroblock chain per wavefront, so two shorter chains side by side
// and the registers of only one of them - 70 instead of 92, seven wavefronts per SIMD - beat one
// long one), one wavefront per macroblock row.  In P pictures the inter MBs were
// already written by k_mc_luma / k_mc_chroma, so only the sparse intra MBs are visited.  Missing neighbours are
// substituted in registers (128 / replicated t3) instead of being written into the frame as the
// reference does (decoder/macroblock.c:697-713, SURVEY A-Q7).
#pragma once
#include <stddef.h>
#include "device_common.h"
#include "wavefront_sync.h"

#define IT_STRIDE 24               // luma tile: row -1..15, byte 3 = left column, 4..19 = MB, 20..23 = top-right
#define CT_STRIDE 12               // chroma tile: byte 3 = left column, 4..11 = MB

struct IntraLds {                  // per wavefront
    uint8_t  y[17 * IT_STRIDE];
    uint8_t  c[2][9 * CT_STRIDE];
    int16_t  coef[16 * 16];
    int16_t  dc[16];
    uint8_t  edge[16];             // Intra4x4: l3 l3 l2 l1 l0 lt t0..t7 t7 of the current block
};

__device__ __forceinline__ int f3(int a, int b, int c) { return (a + 2 * b + c + 2) >> 2; }
__device__ __forceinline__ int f2(int a, int b) { return (a + b + 1) >> 1; }

// Intra 4x4 prediction (core/predict.c:366-638) over an EDGE ARRAY: S[0..14] = l3 l3 l2 l1 l0 lt t0 .. t7 t7 (left
// column bottom-up, corner, top and top-right row; the ends replicated).  Every directional mode is then one of
//   copy S[c],  (S[c] + S[c+1] + 1) >> 1,  (S[c-1] + 2 S[c] + S[c+1] + 2) >> 2
// with an index c that is linear in (x,y) per mode - the mode is wave-uniform, so this is a small scalar switch and
// no per-mode sample code (checked against the per-mode formulas for all modes and positions).
enum { P4_COPY = 0, P4_F2 = 1, P4_F3 = 2 };
__device__ __forceinline__ void pred4x4_where(int mode, int x, int y, int &c, int &kind)
{
    switch (mode) {
    case 0: c = 6 + x; kind = P4_COPY; break;                                                   // vertical
    case 1: c = 4 - y; kind = P4_COPY; break;                                                   // horizontal
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

// Reconstruct one intra macroblock with one wavefront.  Every global load the macroblock needs (neighbour samples,
// prediction modes, coefficients of all three planes) is issued at the top, before anything waits: one memory round
// trip per macroblock, everything after that runs out of registers and LDS.
// LUMA / CHROMA: which planes this wavefront reconstructs (k_intra runs the two as separate workgroups of a picture)
template <bool LUMA, bool CHROMA>
__device__ void intra_mb(const PicDev *pd, const Geom &g, IntraLds &L, int mbi, const p264hip_mb_t m, int lane, RowSync &sync, int row_publish_as)
{
    // Everything below that depends on the lane number alone (roles, tile offsets, scan positions) would otherwise be
    // hoisted out of the caller's macroblock loop and kept - or spilled - across it: recomputing it per macroblock is a few
    // dozen instructions, holding it is ~50 registers.
    asm volatile("" : "+v"(lane));
    const int mbx = mbi % g.mb_w, mby = mbi / g.mb_w, X0 = mbx * 16, Y0 = mby * 16;
    const bool aL = m.avail & P264_AVAIL_LEFT, aT = m.avail & P264_AVAIL_TOP;
    const bool aTR = m.avail & P264_AVAIL_TOPRIGHT, aTL = m.avail & P264_AVAIL_TOPLEFT;
    uint8_t *F = pd->dst;                                  // strip frame layout (device_common.h)
    const AS1 uint8_t *Fg = glob(F);
    const unsigned mask = m.coef_mask;
    const AS1 int16_t *cf = glob(pd->coefs) + (size_t)m.coef_index * 16;
    const bool is16 = m.mb_type == P264_MB_I16x16;

    // ---- (a) neighbour samples; 128 where the neighbour does not exist.  Role A: lanes 0..20 top row (corner, 16 top,
    //      4 top-right), 21..36 left column, 37..54 chroma top rows (corner + 8) of both planes; role B: lanes 0..15
    //      chroma left columns ----
    uint32_t offA = 0, offB = 0; bool okA = false, okB = false;
    uint8_t *const lds8 = (uint8_t *)&L;                   // byte offsets, so that the stores stay LDS stores
    const int c_base = (int)offsetof(IntraLds, c), c_size = (int)sizeof(L.c[0]);
    int dstA = 0, dstB = 0;
    if (!LUMA && lane < 37) {
    } else if (lane < 21) {
        int x = lane - 1;
        okA = x < 0 ? aTL : x < 16 ? aT : aTR;
        offA = luma_off(g, X0 + x, Y0 - 1); dstA = 3 + lane;
    } else if (lane < 37) {
        int r = lane - 21;
        okA = aL; offA = luma_off(g, X0 - 1, Y0 + r); dstA = (r + 1) * IT_STRIDE + 3;
    } else if (CHROMA && lane < 55) {
        int p = (lane - 37) / 9, x = (lane - 37) % 9 - 1;
        okA = x < 0 ? aTL : aT;
        offA = chroma_off(g, p, X0 / 2 + x, Y0 / 2 - 1); dstA = c_base + p * c_size + 3 + x + 1;
    }
    if (CHROMA && lane < 16) {
        int p = lane >> 3, r = lane & 7;
        okB = aL; offB = chroma_off(g, p, X0 / 2 - 1, Y0 / 2 + r); dstB = c_base + p * c_size + (r + 1) * CT_STRIDE + 3;
    }
    int vA = 128, vB = 128;
    if (okA) vA = Fg[offA];
    if (okB) vB = Fg[offB];
    // ---- (b) luma levels: lane = (block lane>>2 in decode order, levels 4*(lane&3) .. +3 of its 16 slots) ----
    const int lb = lane >> 2;
    uint2 lv = make_uint2(0, 0);
    if (LUMA && ((mask >> lb) & 1)) lv = gload2(cf + coef_slot(mask, lb) * 16 + (lane & 3) * 4);
    int ldc = 0;                                           // Intra16x16 DC levels, lanes 0..15
    if (LUMA && is16 && (mask & P264_COEF_LUMA_DC) && lane < 16) ldc = cf[lane];
    // ---- (c) Intra4x4 prediction modes, lanes 0..15 ----
    int modebyte = 0;
    if (LUMA && !is16) modebyte = glob(pd->i4modes)[mbi * 16 + (lane & 15)];
    // ---- (d) chroma: AC level k-1 of block j = lane>>4 of each plane (k = lane&15), DC levels in lanes 0..7 ----
    const bool has_chroma = CHROMA && (m.cbp >> 4) != 0;
    int cac[2] = { 0, 0 }, cdcv = 0;
    if (has_chroma) {
        const int j = lane >> 4, k = lane & 15;
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int blk = 16 + 4 * p + j;
            if (k > 0 && ((mask >> blk) & 1)) cac[p] = cf[coef_slot(mask, blk) * 16 + k - 1];
        }
        if ((mask & P264_COEF_CHROMA_DC) && lane < 8) cdcv = cf[((mask >> 24) & 1) * 16 + lane];
    }

    // Everything left of this macroblock is final once the stores of the previous one have landed: publishing here, where
    // the wave has to wait for its loads anyway, keeps the store latency off the macroblock-to-macroblock path.
    row_publish(sync, mby, row_publish_as);
    // ---- land the neighbours ----
    if (LUMA ? lane < (CHROMA ? 55 : 37) : (lane >= 37 && lane < 55)) lds8[dstA] = (uint8_t)vA;
    if (CHROMA && lane < 16) lds8[dstB] = (uint8_t)vB;
    wave_lds_fence();
    if (LUMA && !aTR && lane < 4) L.y[20 + lane] = L.y[19];  // top-right of the MB missing: replicate t15 (:706-709)
    wave_lds_fence();

    if (!LUMA) {
    } else if (is16) {
        // ---- prediction: each lane 4 samples of one row ----
        int mode = m.intra_modes & 3;
        if (mode == 2) mode = aTL ? 2 : aL ? 4 : aT ? 5 : 6;                    // :635-667
        const int row = lane >> 2, c0 = (lane & 3) * 4;
        const uint8_t *top = L.y + 4, *tile = L.y + IT_STRIDE;
        int pv[4];
        if (mode == 0)      { for (int i = 0; i < 4; i++) pv[i] = top[c0 + i]; }
        else if (mode == 1) { int v = tile[row * IT_STRIDE + 3]; for (int i = 0; i < 4; i++) pv[i] = v; }
        else if (mode == 3) {                                                     // plane, core/predict.c:159-193
            int H = 0, Vv = 0;
            for (int i = 0; i <= 7; i++) {
                H  += (i + 1) * (top[8 + i] - top[6 - i]);                       // top[-1] is the corner
                Vv += (i + 1) * (tile[(8 + i) * IT_STRIDE + 3] - L.y[(7 - i) * IT_STRIDE + 3]);
            }
            int a = 16 * (tile[15 * IT_STRIDE + 3] + top[15]), b = (5 * H + 32) >> 6, c = (5 * Vv + 32) >> 6;
            int i00 = a - 7 * b - 7 * c + 16 + c * row;
            for (int i = 0; i < 4; i++) pv[i] = clip255((i00 + b * (c0 + i)) >> 5);
        } else {
            int s = 0;
            if (mode == 2)      { for (int i = 0; i < 16; i++) s += top[i] + tile[i * IT_STRIDE + 3]; s = (s + 16) >> 5; }
            else if (mode == 4) { for (int i = 0; i < 16; i++) s += tile[i * IT_STRIDE + 3]; s = (s + 8) >> 4; }
            else if (mode == 5) { for (int i = 0; i < 16; i++) s += top[i]; s = (s + 8) >> 4; }
            else s = 128;
            for (int i = 0; i < 4; i++) pv[i] = s;
        }
        // ---- luma DC: unscan, idct4x4dc (core/dct.c:104-136), rounded dequant (core/quant.c:161-191) ----
        if (lane < 16) L.dc[c_zigzag[lane]] = (int16_t)ldc;
        wave_lds_fence();
        int dcv = 0;
        if (lane < 16) {
            int i = lane >> 2, j = lane & 3, tcol[4];
            for (int c = 0; c < 4; c++) {                                       // column pass for tmp[i][c]
                int d0 = L.dc[c], d1 = L.dc[4 + c], d2 = L.dc[8 + c], d3 = L.dc[12 + c];
                int s01 = d0 + d1, d01 = d0 - d1, s23 = d2 + d3, d23 = d2 - d3;
                int v = pick_addsub(s01, d01, s23, d23, i >= 2, i == 1 || i == 2);   // {s01+s23, s01-s23, d01-d23, d01+d23}[i]
                tcol[c] = (int)(int16_t)v;
            }
            int s01 = tcol[0] + tcol[1], d01 = tcol[0] - tcol[1], s23 = tcol[2] + tcol[3], d23 = tcol[2] - tcol[3];
            int v = pick_addsub(s01, d01, s23, d23, j >= 2, j == 1 || j == 2);
            v = (int)(int16_t)v;
            int qbits = m.qp / 6 - 6, mf = dq_scale(0, m.qp % 6);
            v = qbits >= 0 ? v * (int)((unsigned)mf << qbits) : (v * mf + (1 << (-qbits - 1))) >> (-qbits);
            dcv = (int)(int16_t)v;
        }
        wave_lds_fence();
        if (lane < 16) L.dc[lane] = (int16_t)dcv;                                // raster (y*4+x) of the 4x4 block grid
        wave_lds_fence();
        // ---- AC: scan position k of the block holds level k-1 (the slots hold 15 AC levels), DC goes to position 0
        // (:787-794).  This lane fetched levels 4q..4q+3 (q = lane&3); level 4q-1 comes from the lane before. ----
        {
            const int e3 = (int)(int16_t)(lv.y >> 16);
            const int prev = __shfl_up(e3, 1);
            const int lev[4] = { prev, (int)(int16_t)(lv.x & 0xffff), (int)(int16_t)(lv.x >> 16), (int)(int16_t)(lv.y & 0xffff) };
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                int k = (lane & 3) * 4 + kk, pos = c_zigzag[k];
                int v = k == 0 ? L.dc[blk_y(lb) * 4 + blk_x(lb)] : dequant_coef(lev[kk], pos, m.qp);
                L.coef[lb * 16 + pos] = (int16_t)v;
            }
        }
        wave_lds_fence();
        {
            int b = blk_at(lane & 3, row >> 2), yy = row & 3;
            idct4x4_rowpass(L.coef + b * 16, yy);                                // horizontal pass of this lane's row, in place
            wave_lds_fence();
            for (int i = 0; i < 4; i++)
                L.y[(row + 1) * IT_STRIDE + 4 + c0 + i] = (uint8_t)clip255(pv[i] + idct4x4_col_sample(L.coef + b * 16, i, yy));
        }
        wave_lds_fence();
    } else {
        // ---- I4x4: sixteen dependent blocks, 16 lanes each (decoder/macroblock.c:799-831); all coded blocks are
        // unscanned + dequantised into LDS first ----
        {
            const int lev[4] = { (int)(int16_t)(lv.x & 0xffff), (int)(int16_t)(lv.x >> 16), (int)(int16_t)(lv.y & 0xffff), (int)(int16_t)(lv.y >> 16) };
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                int k = (lane & 3) * 4 + kk, pos = c_zigzag[k];
                L.coef[lb * 16 + pos] = (int16_t)dequant_coef(lev[kk], pos, m.qp);
            }
        }
        wave_lds_fence();
        idct4x4_rowpass(L.coef + lb * 16, lane & 3);                              // horizontal pass of all 16 blocks: lane = (block, row)
        wave_lds_fence();
        const int x = lane & 3, y = (lane >> 2) & 3;
        // the lane's slot of the edge array: offset from the block origin inside the tile and the neighbour it belongs to
        // (0 left, 1 top-left, 2 top, 3 top-right)
        const int es = min(lane, 14);
        const int eoff = es <= 4 ? (es == 0 ? 3 : 4 - es) * IT_STRIDE - 1 : es == 5 ? -IT_STRIDE - 1 : -IT_STRIDE + min(es - 6, 7);
        const int eflag = es <= 4 ? 0 : es == 5 ? 1 : es <= 9 ? 2 : 3;
        // which of the 16 blocks (decode order) have their left / top / top-left / top-right neighbour: one bit per block,
        // from the macroblock's availability (inside the MB: fixed pattern, top-right per core/macroblock.c:1210-1231)
        const unsigned left_m = 0xFAFAu | (aL ? 0x0505u : 0u), top_m = 0xFFCCu | (aT ? 0x0033u : 0u);
        const unsigned tl_m = 0xFAC8u | (aT ? 0x0032u : 0u) | (aL ? 0x0504u : 0u) | (aTL ? 0x0001u : 0u);
        const unsigned tr_m = 0x5744u | (aT ? 0x0013u : 0u) | (aTR ? 0x0020u : 0u);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int bx = blk_x(i), by = blk_y(i);
            const bool left = (left_m >> i) & 1, top = (top_m >> i) & 1, topleft = (tl_m >> i) & 1, topright = (tr_m >> i) & 1;
            int mode = __builtin_amdgcn_readlane(modebyte, i);
            const uint8_t *o = L.y + (by * 4 + 1) * IT_STRIDE + 4 + bx * 4;           // block origin inside the tile
            // ---- edge array, missing neighbours substituted (decoder/macroblock.c:697-713): 128, or t3 for the top-right ----
            {
                const unsigned avail = (unsigned)left | (unsigned)topleft << 1 | (unsigned)top << 2 | (unsigned)topright << 3;
                const int own = o[eoff], t3 = o[-IT_STRIDE + 3];
                const int alt = (eflag == 3 && top) ? t3 : 128;
                if (lane < 15) L.edge[lane] = (uint8_t)(((avail >> eflag) & 1) ? own : alt);
            }
            wave_lds_fence();
            int v;
            if (mode == 2) {                                                           // DC and its fall-backs, :677-695
                const uint8_t *S = L.edge;
                const int sl = S[1] + S[2] + S[3] + S[4], st = S[6] + S[7] + S[8] + S[9];
                v = (left && top) ? (sl + st + 4) >> 3 : left ? (sl + 2) >> 2 : top ? (st + 2) >> 2 : 128;
            } else {
                int c, kind;
                pred4x4_where(mode, x, y, c, kind);
                const int a = L.edge[c - 1], b = L.edge[c], d = L.edge[c + 1];
                v = kind == P4_COPY ? b : kind == P4_F2 ? (b + d + 1) >> 1 : (a + 2 * b + d + 2) >> 2;
            }
            if ((mask >> i) & 1) v = clip255(v + idct4x4_col_sample(L.coef + i * 16, x, y));
            if (lane < 16) L.y[(by * 4 + y + 1) * IT_STRIDE + 4 + bx * 4 + x] = (uint8_t)v;
            wave_lds_fence();
        }
    }

    // ---- chroma prediction (core/predict.c:199-361), one sample per lane and plane ----
    if (CHROMA) {
        int mode = (m.intra_modes >> 4) & 3;
        if (mode == 0) mode = aTL ? 0 : aL ? 4 : aT ? 5 : 6;                     // :721-753
        const int px = lane & 7, py = lane >> 3;
        const int qpc = chroma_qp(clip3i(m.qp + pd->chroma_qp_offset, 0, 51));
        const int j = lane >> 4, k = lane & 15, pos = c_zigzag[k];
        for (int p = 0; p < 2; p++) {
            const uint8_t *top = L.c[p] + 4, *tile = L.c[p] + CT_STRIDE;
            int v;
            if (mode == 1) v = tile[py * CT_STRIDE + 3];
            else if (mode == 2) v = top[px];
            else if (mode == 3) {
                int H = 0, Vv = 0;
                for (int i = 0; i < 4; i++) {
                    H  += (i + 1) * (top[4 + i] - top[2 - i]);
                    Vv += (i + 1) * (tile[(4 + i) * CT_STRIDE + 3] - L.c[p][(3 - i) * CT_STRIDE + 3]);
                }
                int a = 16 * (tile[7 * CT_STRIDE + 3] + top[7]), b = (17 * H + 16) >> 5, c = (17 * Vv + 16) >> 5;
                v = clip255((a - 3 * b - 3 * c + 16 + c * py + b * px) >> 5);
            } else {
                int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
                for (int i = 0; i < 4; i++) { s0 += top[i]; s1 += top[4 + i]; s2 += tile[i * CT_STRIDE + 3]; s3 += tile[(4 + i) * CT_STRIDE + 3]; }
                int qd = ((py >> 2) << 1) | (px >> 2);
                if (mode == 0)      v = qd == 0 ? (s0 + s2 + 4) >> 3 : qd == 1 ? (s1 + 2) >> 2 : qd == 2 ? (s3 + 2) >> 2 : (s1 + s3 + 4) >> 3;
                else if (mode == 4) v = (qd < 2 ? s2 + 2 : s3 + 2) >> 2;
                else if (mode == 5) v = ((qd & 1) ? s1 + 2 : s0 + 2) >> 2;
                else v = 128;
            }
            if (has_chroma) {                                                     // residual, same arithmetic as kernel_mc.h
                // the four DC levels of plane p sit in lanes 4p .. 4p+3 (all shuffles before any use)
                const int d0 = __shfl(cdcv, p * 4), d1 = __shfl(cdcv, p * 4 + 1), d2 = __shfl(cdcv, p * 4 + 2), d3 = __shfl(cdcv, p * 4 + 3);
                int cv;
                if (k == 0) {
                    int t0 = d0 + d1, t1 = d0 - d1, t2 = d2 + d3, t3 = d2 - d3;
                    int f = pick_addsub(t0, t1, t2, t3, j & 1, j & 2);            // {t0+t2, t1+t3, t0-t2, t1-t3}[j]
                    f = (int)(int16_t)f;
                    int qbits = qpc / 6 - 5, mf = dq_scale(0, qpc % 6);
                    cv = qbits >= 0 ? f * (int)((unsigned)mf << qbits) : (f * mf) >> (-qbits);
                    cv = (int)(int16_t)cv;
                } else cv = dequant_coef(cac[p], pos, qpc);
                L.coef[j * 16 + pos] = (int16_t)cv;
                wave_lds_fence();
                if (lane < 16) idct4x4_rowpass(L.coef + (lane >> 2) * 16, lane & 3);     // 4 blocks x 4 rows
                wave_lds_fence();
                int jj = ((py >> 2) << 1) | (px >> 2);
                v = clip255(v + idct4x4_col_sample(L.coef + jj * 16, px & 3, py & 3));
            }
            wave_lds_fence();
            L.c[p][(py + 1) * CT_STRIDE + 4 + px] = (uint8_t)v;
            wave_lds_fence();
        }
    }

    // ---- write the macroblock out ----
    {
        int row = lane >> 2, d = lane & 3;
        // lane (row, d) owns dword `lane` of the macroblock's 256 contiguous luma bytes, lane (p, r, dd) a dword of its
        // 128 chroma bytes (rows of 8 bytes U + 8 bytes V)
        if (LUMA) gstore1(F + mb_luma_off(g, mbx, mby) + lane * 4, *(const uint32_t *)(L.y + (row + 1) * IT_STRIDE + 4 + d * 4));
        if (CHROMA && lane < 32) {
            int p = lane >> 4, r = (lane >> 1) & 7, dd = lane & 1;
            gstore1(F + mb_chroma_off(g, mbx, mby) + r * 16 + p * 8 + dd * 4, *(const uint32_t *)(L.c[p] + (r + 1) * CT_STRIDE + 4 + dd * 4));
        }
    }
}

#ifndef INTRA_ROW_WAVES
#define INTRA_ROW_WAVES 16          // most wavefronts per picture workgroup
#endif
#ifndef INTRA_WAVES_PER_EU
#define INTRA_WAVES_PER_EU 4
#endif
__global__ __launch_bounds__(INTRA_ROW_WAVES * 64, INTRA_WAVES_PER_EU)
void k_intra(const PicDev *__restrict__ pics, Geom g, int *status)
{
    __shared__ RowSync sync;
    // one tile set per wavefront, sized by the launch (dynamic shared memory = wavefronts x sizeof(IntraLds)): with the space of
    // sixteen wavefronts reserved for every workgroup only seven workgroups fitted a CU whatever their size - a batch of
    // 1024 pictures (2048 workgroups of 4 wavefronts) then ran as two rounds of workgroups, i.e. took twice a workgroup's time
    extern __shared__ __attribute__((aligned(16))) uint8_t intra_dyn_lds[];
    IntraLds *lds = (IntraLds *)intra_dyn_lds;
    const PicDev *pd = pics + blockIdx.x;
    const bool chroma_role = blockIdx.y != 0;               // grid.y = 2: luma and chroma of a picture in separate workgroups
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    rows_init(sync, g.mb_h);
    bool ok = true;
    const int n_waves = blockDim.x >> 6;                       // 16 per picture, or 8 when two pictures share a CU (host's choice)
    for (int row = wave; row < g.mb_h; row += n_waves) {
        for (int base = 0; base < g.mb_w; base += 64) {
            // which of the next 64 macroblocks of this row are intra, and which of them touch an intra macroblock of
            // the row above?  Only those can still be in flight there (inter MBs were finished by the motion-compensation kernels), so the
            // wavefront dependency only bites where intra macroblocks touch.  One batch of loads per 64 macroblocks.
            const int x = base + lane;
            const AS1 p264hip_mb_t *recs = glob(pd->mb) + row * g.mb_w;
            bool intra = false, dep = false;
            uint4 rec = make_uint4(0, 0, 0, 0);                                   // this lane's macroblock record
            if (x < g.mb_w) {
                rec = gload4(pd->mb + row * g.mb_w + x);
                intra = P264_MB_IS_INTRA(rec.x & 255);
                if (row > 0) {
                    const AS1 p264hip_mb_t *up = recs - g.mb_w;
                    dep = (int)P264_MB_IS_INTRA(up[x].mb_type) | (int)P264_MB_IS_INTRA(up[max(x - 1, 0)].mb_type) |
                          (int)P264_MB_IS_INTRA(up[min(x + 1, g.mb_w - 1)].mb_type);
                }
            }
            unsigned long long todo = __ballot(intra);
            const unsigned long long deps = __ballot(dep);
            while (todo) {
                int bit = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                int mbx = base + bit, mbi = row * g.mb_w + mbx;
                if (((deps >> bit) & 1) && ok) ok = row_wait(sync, row - 1, min(mbx + 2, g.mb_w), status);
                const uint4 mr = make_uint4((uint32_t)__builtin_amdgcn_readlane((int)rec.x, bit), (uint32_t)__builtin_amdgcn_readlane((int)rec.y, bit),
                                            (uint32_t)__builtin_amdgcn_readlane((int)rec.z, bit), (uint32_t)__builtin_amdgcn_readlane((int)rec.w, bit));
                if (chroma_role)       intra_mb<false, true>(pd, g, lds[wave], mbi, __builtin_bit_cast(p264hip_mb_t, mr), lane, sync, mbx);
                else                   intra_mb<true, false>(pd, g, lds[wave], mbi, __builtin_bit_cast(p264hip_mb_t, mr), lane, sync, mbx);
            }
        }
        row_publish(sync, row, g.mb_w);
    }
}
