// kernel_deblock.h - K4: in-loop deblocking filter.
//
// Replaces p264_frame_deblocking_filter + deblock_edge (core/frame.c:472-643) and the four
// sample filters deblock_luma_c / deblock_chroma_c / deblock_luma_intra_c /
// deblock_chroma_intra_c (core/frame.c:302-470).
//
// The filter is defined in macroblock raster order (left edge, inner vertical edges, top edge,
// inner horizontal edges of one MB before the next MB), and the result depends on that order:
// MB (x,y) must see (x-1,y) and (x+1,y-1) completely filtered.  "All vertical edges, then all
// horizontal edges" is NOT bit-exact.  So this is the same row wavefront as k_intra: one
// workgroup per picture, one wavefront per macroblock row, dependencies through LDS counters.
//
// Per macroblock a wavefront: (1) derives the 32 boundary strengths in parallel, (2) pulls the
// 20x20 luma / 2 x 10x12 chroma neighbourhood into LDS with aligned dword loads, (3) runs the
// eight edges in order, one sample line per lane (16 luma + 8 Cb + 8 Cr lanes), (4) writes the
// neighbourhood back with dword stores.  Macroblocks whose 32 strengths are all zero are
// skipped without touching pixels.
#pragma once
#include "device_common.h"
#include "wavefront_sync.h"

#define DY_STRIDE 20               // luma tile: rows -4..15, cols -4..15
#define DC_STRIDE 12               // chroma tile: rows -2..7, cols -4..7

struct DeblockLds {                // per wavefront
    uint8_t y[20 * DY_STRIDE];
    uint8_t c[2][10 * DC_STRIDE];
    uint8_t bs[32];                // [dir][edge][segment]
};

// One line across one edge.  q points at q0 inside an LDS tile, xs = distance between samples
// across the edge.  bS < 4: core/frame.c:302-341 (luma) / 351-377 (chroma); bS == 4: :387-462.
__device__ __forceinline__ void filter_line_luma(uint8_t *q, int xs, int bS, int alpha, int beta, int tc0)
{
    int p2 = q[-3*xs], p1 = q[-2*xs], p0 = q[-xs], q0 = q[0], q1 = q[xs], q2 = q[2*xs];
    if (!(abs(p0 - q0) < alpha && abs(p1 - p0) < beta && abs(q1 - q0) < beta)) return;
    if (bS < 4) {
        int tc = tc0;
        if (abs(p2 - p0) < beta) { q[-2*xs] = (uint8_t)(p1 + clip3i(((p2 + ((p0 + q0 + 1) >> 1)) >> 1) - p1, -tc0, tc0)); tc++; }
        if (abs(q2 - q0) < beta) { q[xs]    = (uint8_t)(q1 + clip3i(((q2 + ((p0 + q0 + 1) >> 1)) >> 1) - q1, -tc0, tc0)); tc++; }
        int delta = clip3i((((q0 - p0) * 4) + (p1 - q1) + 4) >> 3, -tc, tc);
        q[-xs] = (uint8_t)clip255(p0 + delta);
        q[0]   = (uint8_t)clip255(q0 - delta);
    } else {
        if (abs(p0 - q0) < ((alpha >> 2) + 2)) {
            if (abs(p2 - p0) < beta) {
                int p3 = q[-4*xs];
                q[-xs]   = (uint8_t)((p2 + 2*p1 + 2*p0 + 2*q0 + q1 + 4) >> 3);
                q[-2*xs] = (uint8_t)((p2 + p1 + p0 + q0 + 2) >> 2);
                q[-3*xs] = (uint8_t)((2*p3 + 3*p2 + p1 + p0 + q0 + 4) >> 3);
            } else q[-xs] = (uint8_t)((2*p1 + p0 + q1 + 2) >> 2);
            if (abs(q2 - q0) < beta) {
                int q3 = q[3*xs];
                q[0]    = (uint8_t)((p1 + 2*p0 + 2*q0 + 2*q1 + q2 + 4) >> 3);
                q[xs]   = (uint8_t)((p0 + q0 + q1 + q2 + 2) >> 2);
                q[2*xs] = (uint8_t)((2*q3 + 3*q2 + q1 + q0 + p0 + 4) >> 3);
            } else q[0] = (uint8_t)((2*q1 + q0 + p1 + 2) >> 2);
        } else {
            q[-xs] = (uint8_t)((2*p1 + p0 + q1 + 2) >> 2);
            q[0]   = (uint8_t)((2*q1 + q0 + p1 + 2) >> 2);
        }
    }
}

__device__ __forceinline__ void filter_line_chroma(uint8_t *q, int xs, int bS, int alpha, int beta, int tc)
{
    int p1 = q[-2*xs], p0 = q[-xs], q0 = q[0], q1 = q[xs];
    if (!(abs(p0 - q0) < alpha && abs(p1 - p0) < beta && abs(q1 - q0) < beta)) return;
    if (bS < 4) {
        int delta = clip3i((((q0 - p0) * 4) + (p1 - q1) + 4) >> 3, -tc, tc);
        q[-xs] = (uint8_t)clip255(p0 + delta);
        q[0]   = (uint8_t)clip255(q0 - delta);
    } else {
        q[-xs] = (uint8_t)((2*p1 + p0 + q1 + 2) >> 2);
        q[0]   = (uint8_t)((2*q1 + q0 + p1 + 2) >> 2);
    }
}

__device__ void deblock_mb(const PicDev *pd, const Geom &g, DeblockLds &L, int mbi, const p264hip_mb_t m, int lane)
{
    const int mbx = mbi % g.mb_w, mby = mbi / g.mb_w, X0 = mbx * 16, Y0 = mby * 16;
    const bool fL = m.edges & P264_EDGE_LEFT, fT = m.edges & P264_EDGE_TOP;
    const p264hip_mb_t mL = pd->mb[fL ? mbi - 1 : mbi], mT = pd->mb[fT ? mbi - g.mb_w : mbi];

    // ---- (1) boundary strengths, core/frame.c:535-581; lane = dir*16 + edge*4 + segment ----
    int bS = 0;
    if (lane < 32) {
        const int dir = lane >> 4, e = (lane >> 2) & 3, i = lane & 3;
        const bool outer = e == 0;
        const bool enabled = outer ? (dir == 0 ? fL : fT) : true;
        const p264hip_mb_t &n = outer ? (dir == 0 ? mL : mT) : m;
        const int nbi = outer ? (dir == 0 ? mbi - 1 : mbi - g.mb_w) : mbi;
        if (enabled) {
            if (P264_MB_IS_INTRA(m.mb_type) || P264_MB_IS_INTRA(n.mb_type)) bS = outer ? 4 : 3;
            else {
                int x = dir == 0 ? e : i, y = dir == 0 ? i : e;
                int xn = dir == 0 ? (x - 1) & 3 : x, yn = dir == 0 ? y : (y - 1) & 3;
                if (((m.coef_mask >> blk_at(x, y)) & 1) || ((n.coef_mask >> blk_at(xn, yn)) & 1)) bS = 2;
                else {
                    int rp = pd->ref_idx[mbi * 4 + (y >> 1) * 2 + (x >> 1)], rq = pd->ref_idx[nbi * 4 + (yn >> 1) * 2 + (xn >> 1)];
                    int vp = pd->mv[mbi * 16 + y * 4 + x], vq = pd->mv[nbi * 16 + yn * 4 + xn];
                    bS = (rp != rq || abs((int)(int16_t)vp - (int)(int16_t)vq) >= 4 || abs((vp >> 16) - (vq >> 16)) >= 4) ? 1 : 0;
                }
            }
        }
        L.bs[lane] = (uint8_t)bS;
    }
    if (__ballot(bS != 0) == 0) return;                     // nothing to filter in this macroblock
    wave_lds_fence();

    // ---- (2) neighbourhood into LDS (aligned dwords; the top-left 4x4 corner is never touched) ----
    uint8_t *Y = pd->dst, *U = pd->dst + g.off_u, *V = pd->dst + g.off_v;
    auto luma_word = [&](int idx, int &row, int &col) {     // 96 dwords: 16 rows x 5, then 4 top rows x 4
        if (idx < 80) { row = idx / 5; col = (idx % 5) * 4 - 4; return col >= 0 || fL; }
        idx -= 80; row = idx / 4 - 4; col = (idx % 4) * 4; return (bool)fT;
    };
    auto chroma_word = [&](int idx, int &p, int &row, int &col) {   // 2 planes x (8 rows x 3 + 2 top rows x 2) = 56 dwords
        p = idx / 28; idx %= 28;
        if (idx < 24) { row = idx / 3; col = (idx % 3) * 4 - 4; return col >= 0 || fL; }
        idx -= 24; row = idx / 2 - 2; col = (idx % 2) * 4; return (bool)fT;
    };
    for (int idx = lane; idx < 96; idx += 64) {
        int row, col;
        if (luma_word(idx, row, col))
            *(uint32_t *)(L.y + (row + 4) * DY_STRIDE + col + 4) = *(const uint32_t *)(Y + (size_t)(Y0 + row) * g.w + X0 + col);
    }
    if (lane < 56) {
        int p, row, col;
        if (chroma_word(lane, p, row, col))
            *(uint32_t *)(L.c[p] + (row + 2) * DC_STRIDE + col + 4) = *(const uint32_t *)((p ? V : U) + (size_t)(Y0 / 2 + row) * g.cw + X0 / 2 + col);
    }
    wave_lds_fence();

    // ---- (3) the eight edges in reference order; lanes 0-15 luma lines, 16-23 Cb, 24-31 Cr ----
    const int qp = m.qp;
    const int qpc_cur = c_chroma_qp[clip3i(qp + pd->chroma_qp_offset, 0, 51)];
    for (int dir = 0; dir < 2; dir++) {
        for (int e = 0; e < 4; e++) {
            if (lane < 32) {
                const bool outer = e == 0;
                const int qpn = outer ? (dir == 0 ? mL.qp : mT.qp) : qp;
                const bool chroma = lane >= 16;
                if (!chroma || !(e & 1)) {
                    const int line = chroma ? lane & 7 : lane;                 // position along the edge
                    const int seg = chroma ? line >> 1 : line >> 2;
                    const int b = L.bs[dir * 16 + e * 4 + seg];
                    int q_edge;                                                  // core/frame.c:593-601
                    if (!chroma) q_edge = (qp + qpn + 1) >> 1;
                    else q_edge = (qpc_cur + c_chroma_qp[clip3i(qpn + pd->chroma_qp_offset, 0, 51)] + 1) >> 1;
                    const int ia = clip3i(q_edge + pd->alpha_off, 0, 51);       // offsets unshifted: A-Q3
                    const int alpha = c_alpha[ia], beta = c_beta[clip3i(q_edge + pd->beta_off, 0, 51)];
                    const int first = L.bs[dir * 16 + e * 4];                    // deblock_edge keys the filter type on bS[0] (:480)
                    if (!chroma) {
                        uint8_t *q = dir == 0 ? L.y + (line + 4) * DY_STRIDE + 4 + 4 * e
                                              : L.y + (4 + 4 * e) * DY_STRIDE + 4 + line;
                        if (first < 4) { if (b) filter_line_luma(q, dir == 0 ? 1 : DY_STRIDE, b, alpha, beta, c_tc0[ia][b - 1]); }
                        else filter_line_luma(q, dir == 0 ? 1 : DY_STRIDE, 4, alpha, beta, 0);
                    } else {
                        uint8_t *t = L.c[(lane >> 3) & 1];
                        uint8_t *q = dir == 0 ? t + (line + 2) * DC_STRIDE + 4 + 2 * e
                                              : t + (2 + 2 * e) * DC_STRIDE + 4 + line;
                        if (first < 4) { if (b) filter_line_chroma(q, dir == 0 ? 1 : DC_STRIDE, b, alpha, beta, c_tc0[ia][b - 1] + 1); }
                        else filter_line_chroma(q, dir == 0 ? 1 : DC_STRIDE, 4, alpha, beta, 0);
                    }
                }
            }
            wave_lds_fence();
        }
    }

    // ---- (4) write back ----
    for (int idx = lane; idx < 96; idx += 64) {
        int row, col;
        if (luma_word(idx, row, col))
            *(uint32_t *)(Y + (size_t)(Y0 + row) * g.w + X0 + col) = *(const uint32_t *)(L.y + (row + 4) * DY_STRIDE + col + 4);
    }
    if (lane < 56) {
        int p, row, col;
        if (chroma_word(lane, p, row, col))
            *(uint32_t *)((p ? V : U) + (size_t)(Y0 / 2 + row) * g.cw + X0 / 2 + col) = *(const uint32_t *)(L.c[p] + (row + 2) * DC_STRIDE + col + 4);
    }
}

__global__ __launch_bounds__(ROW_WAVES * 64)
void k_deblock(const PicDev *__restrict__ pics, Geom g, int *status)
{
    __shared__ RowSync sync;
    __shared__ DeblockLds lds[ROW_WAVES];
    const PicDev *pd = pics + blockIdx.x;
    if (!pd->deblock) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    rows_init(sync, g.mb_h);
    bool ok = true;
    for (int row = wave; row < g.mb_h; row += ROW_WAVES) {
        for (int mbx = 0; mbx < g.mb_w; mbx++) {
            const int mbi = row * g.mb_w + mbx;
            const p264hip_mb_t m = pd->mb[mbi];
            if (m.edges) {
                if (row > 0 && ok) ok = row_wait(sync, row - 1, min(mbx + 2, g.mb_w), status);
                deblock_mb(pd, g, lds[wave], mbi, m, lane);
            }
            row_publish(sync, row, mbx + 1);
        }
    }
}
