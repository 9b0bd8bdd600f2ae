// kernel_deblock.h - K4: in-loop deblocking filter, in two kernels.
//
// Replaces p264_frame_deblocking_filter + deblock_edge (core/frame.c:472-643) and the four
// sample filters deblock_luma_c / deblock_chroma_c / deblock_luma_intra_c /
// deblock_chroma_intra_c (core/frame.c:302-470).
//
// K4a k_deblock_bs   - everything about an edge that does not depend on pixels: the 32 boundary
//                      strengths of a macroblock (core/frame.c:535-581) and, per edge class
//                      {left, top, inner} x {luma, chroma}, alpha / beta / tc0[bS] from the averaged
//                      QPs (core/frame.c:472-488,593-601).  Fully parallel, one launch per batch,
//                      48 bytes of "edge info" per macroblock.
// K4b k_deblock      - the sample filters.  The filter is defined in macroblock raster order
//                      (left edge, inner vertical edges, top edge, inner horizontal edges of one MB
//                      before the next MB) and the result depends on that order: MB (x,y) must see
//                      (x-1,y) and (x+1,y-1) completely filtered ("all vertical, then all
//                      horizontal edges" is NOT bit-exact).  So this is a row wavefront
//                      (wavefront_sync.h): one workgroup per picture, one wavefront per MB row.
//
// v2 of K4b is software-pipelined along the row so that no global round trip sits on the
// macroblock-to-macroblock critical path:
//   * edge info of 64 macroblocks is fetched with one coalesced load per lane;
//   * while MB x is filtered in LDS tile A, the pixels of MB x+1 are already in flight
//     (own 16x16 + 2 x 8x8 and the 4 / 2 rows above it) and land in tile B;
//   * the 4 (2) rightmost columns of MB x stay in LDS and become the left neighbourhood of
//     MB x+1, and are written back with it - every sample is read once and written once;
//   * progress is published one macroblock late, at the point where the wavefront waits for
//     its prefetch anyway, so the stores of MB x-1 have long completed.
#pragma once
#include "device_common.h"
#include "wavefront_sync.h"

#define DY_DW 5                    // luma tile row: 5 dwords = cols -4..15
#define DC_DW 3                    // chroma tile row: 3 dwords = cols -4..7
#define DY_STRIDE (DY_DW * 4)
#define DC_STRIDE (DC_DW * 4)

enum { EC_LEFT = 0, EC_TOP = 1, EC_INNER = 2 };     // edge classes

struct EdgeInfo {                  // 48 bytes per macroblock, written by k_deblock_bs
    uint32_t bs[4];                // 8 edges x 4 segments x 4 bits: word = dir*2 + (edge>>1), nibble = (edge&1)*4 + seg
    uint8_t  ab[6][2];             // [class + 3*chroma] = { alpha, beta }
    uint8_t  tc[6][3];             // tc0 for bS 1..3 (chroma: already +1)
    uint8_t  any;                  // some bS != 0
    uint8_t  pad;
};

struct DeblockLds {                // per wavefront: two tiles (ping-pong)
    uint32_t y[2][20 * DY_DW];     // rows -4..15
    uint32_t c[2][2][10 * DC_DW];  // [tile][plane], rows -2..7
};

// ------------------------------------------------------------------------------------------
// K4a
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void k_deblock_bs(const PicDev *__restrict__ pics, Geom g, EdgeInfo *__restrict__ info, int n_pics)
{
    // 32 lanes per macroblock (one per edge segment), 8 macroblocks per workgroup
    const int t = blockIdx.x * 8 + (threadIdx.x >> 5);
    const int pic = t / g.n_mb, mbi = t - pic * g.n_mb;
    if (pic >= n_pics) return;
    const PicDev *pd = pics + pic;
    if (!pd->deblock) return;
    const int lane = threadIdx.x & 31;
    const p264hip_mb_t m = pd->mb[mbi];
    const bool fL = m.edges & P264_EDGE_LEFT, fT = m.edges & P264_EDGE_TOP;
    const p264hip_mb_t mL = pd->mb[fL ? mbi - 1 : mbi], mT = pd->mb[fT ? mbi - g.mb_w : mbi];
    EdgeInfo *out = info + (size_t)pic * g.n_mb + mbi;

    // ---- boundary strengths, core/frame.c:535-581; lane = dir*16 + edge*4 + segment ----
    int bS = 0;
    {
        const int dir = lane >> 4, e = (lane >> 2) & 3, i = lane & 3;
        const bool outer = e == 0;
        const bool enabled = m.edges && (outer ? (dir == 0 ? fL : fT) : true);
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
    }
    // pack 8 nibbles per word: lanes 8w .. 8w+7 -> word w
    uint32_t word = (uint32_t)bS << (4 * (lane & 7));
    word |= __shfl_xor(word, 1); word |= __shfl_xor(word, 2); word |= __shfl_xor(word, 4);
    const unsigned long long nz = __ballot(bS != 0);
    const bool any = ((nz >> (threadIdx.x & 32)) & 0xffffffffull) != 0;
    if ((lane & 7) == 0) out->bs[lane >> 3] = word;

    // ---- per edge class: alpha, beta, tc0 (deblock_edge, core/frame.c:472-488; offsets unshifted: A-Q3) ----
    if (lane < 6) {
        const int cls = lane % 3, chroma = lane / 3;
        const int qp = m.qp, qpn = cls == EC_LEFT ? mL.qp : cls == EC_TOP ? mT.qp : m.qp;
        int q;
        if (!chroma) q = (qp + qpn + 1) >> 1;                                     // :593-595
        else q = (c_chroma_qp[clip3i(qp + pd->chroma_qp_offset, 0, 51)] + c_chroma_qp[clip3i(qpn + pd->chroma_qp_offset, 0, 51)] + 1) >> 1;   // :600-601
        const int ia = clip3i(q + pd->alpha_off, 0, 51);
        out->ab[lane][0] = c_alpha[ia];
        out->ab[lane][1] = c_beta[clip3i(q + pd->beta_off, 0, 51)];
        for (int b = 0; b < 3; b++) out->tc[lane][b] = (uint8_t)(c_tc0[ia][b] + chroma);
    }
    if (lane == 0) { out->any = any ? 1 : 0; out->pad = 0; }
}

// ------------------------------------------------------------------------------------------
// sample filters on an LDS tile
// ------------------------------------------------------------------------------------------
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

// the eight edges of one macroblock in reference order; lanes 0-15 luma lines, 16-23 Cb, 24-31 Cr.
// e0..e11 = the macroblock's EdgeInfo as 12 wave-uniform dwords.
__device__ __forceinline__ void filter_mb(uint32_t *ty, uint32_t *tcb, uint32_t *tcr, const uint32_t (&e)[12], int lane)
{
    const uint8_t *eb = nullptr; (void)eb;
    auto ab = [&](int k, int j) { int idx = 16 + k * 2 + j; return (int)((e[idx >> 2] >> (8 * (idx & 3))) & 255); };
    auto tcv = [&](int k, int b) { int idx = 28 + k * 3 + b; return (int)((e[idx >> 2] >> (8 * (idx & 3))) & 255); };
    const bool chroma = lane >= 16;
    const int line = chroma ? lane & 7 : lane;
    const int seg = chroma ? line >> 1 : line >> 2;
    uint8_t *yb = (uint8_t *)ty, *cb = (uint8_t *)((lane >> 3) & 1 ? tcr : tcb);
#pragma unroll
    for (int dir = 0; dir < 2; dir++) {
#pragma unroll
        for (int ed = 0; ed < 4; ed++) {
            const uint32_t nib = (e[dir * 2 + (ed >> 1)] >> ((ed & 1) * 16)) & 0xffffu;   // 4 segments of this edge (wave-uniform)
            if (nib != 0 && lane < 32 && !(chroma && (ed & 1))) {
                const int b = (nib >> (4 * seg)) & 15;
                const int first = nib & 15;                                            // deblock_edge keys the filter type on bS[0] (:480)
                const int k = (ed == 0 ? (dir == 0 ? EC_LEFT : EC_TOP) : EC_INNER) + (chroma ? 3 : 0);
                const int alpha = ab(k, 0), beta = ab(k, 1);
                if (!chroma) {
                    uint8_t *q = dir == 0 ? yb + (line + 4) * DY_STRIDE + 4 + 4 * ed : yb + (4 + 4 * ed) * DY_STRIDE + 4 + line;
                    if (first < 4) { if (b) filter_line_luma(q, dir == 0 ? 1 : DY_STRIDE, b, alpha, beta, tcv(k, b - 1)); }
                    else filter_line_luma(q, dir == 0 ? 1 : DY_STRIDE, 4, alpha, beta, 0);
                } else {
                    uint8_t *q = dir == 0 ? cb + (line + 2) * DC_STRIDE + 4 + 2 * ed : cb + (2 + 2 * ed) * DC_STRIDE + 4 + line;
                    if (first < 4) { if (b) filter_line_chroma(q, dir == 0 ? 1 : DC_STRIDE, b, alpha, beta, tcv(k, b - 1)); }
                    else filter_line_chroma(q, dir == 0 ? 1 : DC_STRIDE, 4, alpha, beta, 0);
                }
            }
            wave_lds_fence();
        }
    }
}

// ------------------------------------------------------------------------------------------
// K4b
// ------------------------------------------------------------------------------------------
// Lane -> dword maps of the per-macroblock pixel traffic.  "own": the MB's 16x16 luma (64 dwords);
// "aux": 16 dwords of the 4 luma rows above, 32 dwords of the two 8x8 chroma blocks, 8 dwords of the
// 2 chroma rows above each.
struct AuxMap { int plane, row, dwcol; bool luma, valid; };      // row / dwcol relative to the MB (dwcol in dwords, 0 = col 0)
__device__ __forceinline__ AuxMap aux_map(int lane)
{
    AuxMap a; a.valid = lane < 56; a.luma = lane < 16; a.plane = 0;
    if (lane < 16) { a.row = (lane >> 2) - 4; a.dwcol = lane & 3; }
    else if (lane < 48) { int l = lane - 16; a.plane = l >> 4; a.row = (l >> 1) & 7; a.dwcol = l & 1; }
    else { int l = lane - 48; a.plane = l >> 2; a.row = ((l >> 1) & 1) - 2; a.dwcol = l & 1; }
    return a;
}

__global__ __launch_bounds__(ROW_WAVES * 64)
void k_deblock(const PicDev *__restrict__ pics, Geom g, const EdgeInfo *__restrict__ info, int *status)
{
    __shared__ RowSync sync;
    __shared__ DeblockLds lds[ROW_WAVES];
    const PicDev *pd = pics + blockIdx.x;
    if (!pd->deblock) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    rows_init(sync, g.mb_h);
    DeblockLds &L = lds[wave];
    uint8_t *Y = pd->dst, *U = pd->dst + g.off_u, *V = pd->dst + g.off_v;
    const EdgeInfo *pinfo = info + (size_t)blockIdx.x * g.n_mb;
    const AuxMap am = aux_map(lane);
    bool ok = true;

    for (int row = wave; row < g.mb_h; row += ROW_WAVES) {
        const bool fT = row > 0;                    // rows above exist (their filtering is what we wait for)
        const int Y0 = row * 16;
        uint32_t pre_own = 0, pre_aux = 0;          // prefetched pixels of the next macroblock
        uint32_t ei[12];                            // edge info of macroblock (chunk base + lane)

        // prefetch of macroblock x (pixels only; row-above dependency first)
        auto prefetch = [&](int x) {
            if (fT && ok) ok = row_wait(sync, row - 1, min(x + 2, g.mb_w), status);
            const int X0 = x * 16;
            pre_own = *(const uint32_t *)(Y + (size_t)(Y0 + (lane >> 2)) * g.w + X0 + (lane & 3) * 4);
            pre_aux = 0;
            if (am.valid && (am.row >= 0 || fT)) {
                if (am.luma) pre_aux = *(const uint32_t *)(Y + (size_t)(Y0 + am.row) * g.w + X0 + am.dwcol * 4);
                else pre_aux = *(const uint32_t *)((am.plane ? V : U) + (size_t)(Y0 / 2 + am.row) * g.cw + X0 / 2 + am.dwcol * 4);
            }
        };
        // land the prefetched registers in tile t
        auto land = [&](int t) {
            L.y[t][((lane >> 2) + 4) * DY_DW + 1 + (lane & 3)] = pre_own;
            if (am.valid) {
                if (am.luma) L.y[t][(am.row + 4) * DY_DW + 1 + am.dwcol] = pre_aux;
                else L.c[t][am.plane][(am.row + 2) * DC_DW + 1 + am.dwcol] = pre_aux;
            }
        };

        prefetch(0);
        int cur = 0;
        for (int x = 0; x < g.mb_w; x++) {
            if ((x & 63) == 0) {                    // edge info of the next 64 macroblocks: 3 x 16 bytes per lane
                int xi = min(x + lane, g.mb_w - 1);
                const uint4 *src = (const uint4 *)(pinfo + row * g.mb_w + xi);
                uint4 a = src[0], b = src[1], c = src[2];
                ei[0] = a.x; ei[1] = a.y; ei[2] = a.z; ei[3] = a.w; ei[4] = b.x; ei[5] = b.y; ei[6] = b.z; ei[7] = b.w;
                ei[8] = c.x; ei[9] = c.y; ei[10] = c.z; ei[11] = c.w;
            }
            if (x == 0) { land(cur); wave_lds_fence(); }
            // (1) pixels of the next macroblock start their trip now
            if (x + 1 < g.mb_w) prefetch(x + 1);
            // (2) filter macroblock x in tile `cur`
            uint32_t e[12];
#pragma unroll
            for (int k = 0; k < 12; k++) e[k] = (uint32_t)__builtin_amdgcn_readlane((int)ei[k], x & 63);
            if ((e[11] >> 16) & 255) filter_mb(L.y[cur], L.c[cur][0], L.c[cur][1], e, lane);
            // (3) everything issued before this point has completed: the stores of macroblock x-1 and the
            //     prefetch of x+1.  Publish x-1 (release = s_waitcnt vmcnt(0) + the LDS store).
            row_publish(sync, row, x);
            // (4) hand the right-hand columns over to the next tile, land the prefetch next to them
            const int nxt = cur ^ 1;
            if (x + 1 < g.mb_w) {
                land(nxt);
                if (lane < 16) L.y[nxt][(lane + 4) * DY_DW] = L.y[cur][(lane + 4) * DY_DW + 4];
                else if (lane < 32) { int p = (lane >> 3) & 1, r = lane & 7; L.c[nxt][p][(r + 2) * DC_DW] = L.c[cur][p][(r + 2) * DC_DW + 2]; }
            }
            // (5) write macroblock x back: columns -4..11 (and 12..15 for the last MB of the row), rows above included
            {
                const int X0 = x * 16;
                const bool last = x + 1 == g.mb_w;
                const int r = lane >> 2, d = lane & 3;                          // luma rows 0..15, tile dwords 0..3
                if (d > 0 || x > 0) *(uint32_t *)(Y + (size_t)(Y0 + r) * g.w + X0 - 4 + d * 4) = L.y[cur][(r + 4) * DY_DW + d];
                if (am.valid && (am.row >= 0 || fT)) {
                    if (am.luma) *(uint32_t *)(Y + (size_t)(Y0 + am.row) * g.w + X0 + am.dwcol * 4) = L.y[cur][(am.row + 4) * DY_DW + 1 + am.dwcol];
                    else if (am.row < 0) *(uint32_t *)((am.plane ? V : U) + (size_t)(Y0 / 2 + am.row) * g.cw + X0 / 2 + am.dwcol * 4) = L.c[cur][am.plane][(am.row + 2) * DC_DW + 1 + am.dwcol];
                    else if (am.dwcol > 0 || x > 0)                             // chroma rows 0..7: tile dwords 0..1 = cols -4..3
                        *(uint32_t *)((am.plane ? V : U) + (size_t)(Y0 / 2 + am.row) * g.cw + X0 / 2 - 4 + am.dwcol * 4) = L.c[cur][am.plane][(am.row + 2) * DC_DW + am.dwcol];
                }
                if (last) {
                    if (lane < 16) *(uint32_t *)(Y + (size_t)(Y0 + lane) * g.w + X0 + 12) = L.y[cur][(lane + 4) * DY_DW + 4];
                    else if (lane < 32) { int p = (lane >> 3) & 1, rr = lane & 7; *(uint32_t *)((p ? V : U) + (size_t)(Y0 / 2 + rr) * g.cw + X0 / 2 + 4) = L.c[cur][p][(rr + 2) * DC_DW + 2]; }
                }
            }
            wave_lds_fence();
            cur = nxt;
        }
        row_publish(sync, row, g.mb_w);             // waits for the last stores of the row
    }
}
