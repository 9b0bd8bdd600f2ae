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
#ifndef EXPD_NOFILTER
#define EXPD_NOFILTER 0
#endif
#ifndef EXPD_NOSTORE
#define EXPD_NOSTORE 0
#endif
#ifndef EXPD_NOLOAD
#define EXPD_NOLOAD 0
#endif
#ifndef EXPD_NOWAIT
#define EXPD_NOWAIT 0
#endif
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
// sample filters on register values
// ------------------------------------------------------------------------------------------
// One line across one edge: p[0..3] = p0..p3, q[0..3] = q0..q3 (ints), updated in place.
// bS < 4: core/frame.c:302-341 (luma) / 351-377 (chroma); bS == 4: :387-462.
__device__ __forceinline__ void filter_luma(int (&p)[4], int (&q)[4], int bS, int alpha, int beta, int tc0)
{
    const int p0 = p[0], p1 = p[1], p2 = p[2], q0 = q[0], q1 = q[1], q2 = q[2];
    if (!(abs(p0 - q0) < alpha && abs(p1 - p0) < beta && abs(q1 - q0) < beta)) return;
    if (bS < 4) {
        int tc = tc0;
        if (abs(p2 - p0) < beta) { p[1] = p1 + clip3i(((p2 + ((p0 + q0 + 1) >> 1)) >> 1) - p1, -tc0, tc0); tc++; }
        if (abs(q2 - q0) < beta) { q[1] = q1 + clip3i(((q2 + ((p0 + q0 + 1) >> 1)) >> 1) - q1, -tc0, tc0); tc++; }
        int delta = clip3i((((q0 - p0) * 4) + (p1 - q1) + 4) >> 3, -tc, tc);
        p[0] = clip255(p0 + delta);
        q[0] = clip255(q0 - delta);
    } else {
        if (abs(p0 - q0) < ((alpha >> 2) + 2)) {
            if (abs(p2 - p0) < beta) {
                p[0] = (p2 + 2*p1 + 2*p0 + 2*q0 + q1 + 4) >> 3;
                p[1] = (p2 + p1 + p0 + q0 + 2) >> 2;
                p[2] = (2*p[3] + 3*p2 + p1 + p0 + q0 + 4) >> 3;
            } else p[0] = (2*p1 + p0 + q1 + 2) >> 2;
            if (abs(q2 - q0) < beta) {
                q[0] = (p1 + 2*p0 + 2*q0 + 2*q1 + q2 + 4) >> 3;
                q[1] = (p0 + q0 + q1 + q2 + 2) >> 2;
                q[2] = (2*q[3] + 3*q2 + q1 + q0 + p0 + 4) >> 3;
            } else q[0] = (2*q1 + q0 + p1 + 2) >> 2;
        } else {
            p[0] = (2*p1 + p0 + q1 + 2) >> 2;
            q[0] = (2*q1 + q0 + p1 + 2) >> 2;
        }
    }
}
__device__ __forceinline__ void filter_chroma(int &p0, int p1, int &q0, int q1, int bS, int alpha, int beta, int tc)
{
    if (!(abs(p0 - q0) < alpha && abs(p1 - p0) < beta && abs(q1 - q0) < beta)) return;
    if (bS < 4) {
        int delta = clip3i((((q0 - p0) * 4) + (p1 - q1) + 4) >> 3, -tc, tc);
        int np = clip255(p0 + delta), nq = clip255(q0 - delta);
        p0 = np; q0 = nq;
    } else {
        int np = (2*p1 + p0 + q1 + 2) >> 2, nq = (2*q1 + q0 + p1 + 2) >> 2;
        p0 = np; q0 = nq;
    }
}

// wave-uniform accessors into the 12 dwords of an EdgeInfo
struct EdgeRegs {
    uint32_t e[12];
    __device__ __forceinline__ uint32_t nib(int dir, int ed) const { return (e[dir * 2 + (ed >> 1)] >> ((ed & 1) * 16)) & 0xffffu; }
    __device__ __forceinline__ int ab(int k, int j) const { int i = 16 + k * 2 + j; return (int)((e[i >> 2] >> (8 * (i & 3))) & 255); }
    // k is a compile-time constant at every call site; only b (0..2) varies per lane, so the three bytes are
    // gathered with static indices and b selects by shift (a dynamic e[] index would push the array to memory)
    __device__ __forceinline__ int byte_at(int i) const { return (int)((e[i >> 2] >> (8 * (i & 3))) & 255); }
    __device__ __forceinline__ int tc(int k, int b) const
    {
        uint32_t three = (uint32_t)byte_at(28 + k * 3) | ((uint32_t)byte_at(29 + k * 3) << 8) | ((uint32_t)byte_at(30 + k * 3) << 16);
        return (int)((three >> (8 * b)) & 255);
    }
    __device__ __forceinline__ bool any() const { return (e[11] >> 16) & 255; }
};
__device__ __forceinline__ int edge_class(int dir, int ed) { return ed == 0 ? (dir == 0 ? EC_LEFT : EC_TOP) : EC_INNER; }

// ------------------------------------------------------------------------------------------
// K4b
// ------------------------------------------------------------------------------------------
// Lane roles inside a wavefront (one macroblock at a time):
//   0..15   luma row `lane`:        v[0] = cols -4..-1 (carried over from the previous MB), v[1..4] = cols 0..15
//   16..31  chroma plane (lane>>3)&1, row lane&7:  v[0] = cols -4..-1, v[1..2] = cols 0..7
//   32..35  luma row lane-36 (-4..-1) of the MB above: v[1..4]
//   36..39  chroma plane (lane>>1)&1, row (lane&1)-2 of the MB above: v[1..2]
// Vertical edges are filtered in these registers (an edge sits on a dword boundary); horizontal
// edges need columns, so the tile takes one trip through LDS: rows in, columns out, columns in,
// rows out.  The right-hand dword of every row stays in its lane for the next macroblock.
#ifndef DEBLOCK_WAVES_PER_EU
#define DEBLOCK_WAVES_PER_EU 4
#endif
__global__ __launch_bounds__(ROW_WAVES * 64, DEBLOCK_WAVES_PER_EU)
void k_deblock(const PicDev *__restrict__ pics, Geom g, const EdgeInfo *__restrict__ info, int *status)
{
    __shared__ RowSync sync;
    __shared__ uint32_t tiles[ROW_WAVES][20 * DY_DW + 2 * 10 * DC_DW];
    const PicDev *pd = pics + blockIdx.x;
    if (!pd->deblock) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    rows_init(sync, g.mb_h);
    uint32_t *ty = tiles[wave];
    uint8_t *Y = pd->dst, *U = pd->dst + g.off_u, *V = pd->dst + g.off_v;
    const EdgeInfo *pinfo = info + (size_t)blockIdx.x * g.n_mb;
    const bool isY = lane < 16, isC = lane >= 16 && lane < 32, isTY = lane >= 32 && lane < 36, isTC = lane >= 36 && lane < 40;
    const int cpl = isC ? (lane >> 3) & 1 : (lane >> 1) & 1;          // chroma plane of this lane (C and TC roles)
    uint32_t *tc_ = ty + 20 * DY_DW + cpl * 10 * DC_DW;                 // this lane's chroma tile
    bool ok = true;

    for (int row = wave; row < g.mb_h; row += ROW_WAVES) {
        const bool fT = row > 0;
        const int Y0 = row * 16;
        // address of this lane's row segment for macroblock 0 (advances by 16 / 8 bytes per macroblock)
        const uint8_t *src = Y;
        if (isY) src = Y + (size_t)(Y0 + lane) * g.w;
        else if (isC) src = (cpl ? V : U) + (size_t)(Y0 / 2 + (lane & 7)) * g.cw;
        else if (isTY) src = Y + (size_t)(Y0 + lane - 36) * g.w;
        else if (isTC) src = (cpl ? V : U) + (size_t)(Y0 / 2 + (lane & 1) - 2) * g.cw;
        const bool wide = isY || (isTY && fT), narrow = isC || (isTC && fT);
        uint4 pre4 = make_uint4(0, 0, 0, 0); uint2 pre2 = make_uint2(0, 0);
        uint32_t v[5] = { 0, 0, 0, 0, 0 };
        uint32_t ei[12];

        auto prefetch = [&](int x) {
            if (fT && ok && !EXPD_NOWAIT) ok = row_wait(sync, row - 1, min(x + 2, g.mb_w), status);
            if (EXPD_NOLOAD) return;
            if (wide) pre4 = *(const uint4 *)(src + x * 16);
            if (narrow) pre2 = *(const uint2 *)(src + x * 8);
        };

        prefetch(0);
        for (int x = 0; x < g.mb_w; x++) {
            if ((x & 63) == 0) {                    // edge info of the next 64 macroblocks: 3 x 16 bytes per lane
                int xi = min(x + lane, g.mb_w - 1);
                const uint4 *s4 = (const uint4 *)(pinfo + row * g.mb_w + xi);
                uint4 a = s4[0], b = s4[1], c = s4[2];
                ei[0] = a.x; ei[1] = a.y; ei[2] = a.z; ei[3] = a.w; ei[4] = b.x; ei[5] = b.y; ei[6] = b.z; ei[7] = b.w;
                ei[8] = c.x; ei[9] = c.y; ei[10] = c.z; ei[11] = c.w;
            }
            // the prefetched pixels of macroblock x move into place (v[0] holds the carried-over columns)
            if (isY || isTY) { v[1] = pre4.x; v[2] = pre4.y; v[3] = pre4.z; v[4] = pre4.w; }
            else { v[1] = pre2.x; v[2] = pre2.y; }
            if (x + 1 < g.mb_w) prefetch(x + 1);    // pixels of the next macroblock start their trip now
            EdgeRegs E;
#pragma unroll
            for (int k = 0; k < 12; k++) E.e[k] = (uint32_t)__builtin_amdgcn_readlane((int)ei[k], x & 63);

            if (E.any() && !EXPD_NOFILTER) {
                // ---------- vertical edges, in registers ----------
                if (E.e[0] | E.e[1]) {
                    if (isY) {
                        const int seg = lane >> 2;
#pragma unroll
                        for (int ed = 0; ed < 4; ed++) {
                            const uint32_t nib = E.nib(0, ed);
                            if (nib == 0) continue;
                            const int b = (nib >> (4 * seg)) & 15, first = nib & 15, k = edge_class(0, ed);
                            if (first >= 4 || b) {
                                int p[4] = { (int)(v[ed] >> 24), (int)((v[ed] >> 16) & 255), (int)((v[ed] >> 8) & 255), (int)(v[ed] & 255) };
                                int q[4] = { (int)(v[ed+1] & 255), (int)((v[ed+1] >> 8) & 255), (int)((v[ed+1] >> 16) & 255), (int)(v[ed+1] >> 24) };
                                filter_luma(p, q, first >= 4 ? 4 : b, E.ab(k, 0), E.ab(k, 1), first >= 4 ? 0 : E.tc(k, b - 1));
                                v[ed] = (uint32_t)p[3] | ((uint32_t)p[2] << 8) | ((uint32_t)p[1] << 16) | ((uint32_t)p[0] << 24);
                                v[ed+1] = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
                            }
                        }
                    } else if (isC) {
                        const int seg = (lane & 7) >> 1;
#pragma unroll
                        for (int ed = 0; ed < 4; ed += 2) {
                            const uint32_t nib = E.nib(0, ed);
                            if (nib == 0) continue;
                            const int b = (nib >> (4 * seg)) & 15, first = nib & 15, k = edge_class(0, ed) + 3;
                            if (first >= 4 || b) {
                                const int j = ed >> 1;
                                int p1 = (int)((v[j] >> 16) & 255), p0 = (int)(v[j] >> 24), q0 = (int)(v[j+1] & 255), q1 = (int)((v[j+1] >> 8) & 255);
                                filter_chroma(p0, p1, q0, q1, first >= 4 ? 4 : b, E.ab(k, 0), E.ab(k, 1), first >= 4 ? 0 : E.tc(k, b - 1));
                                v[j] = (v[j] & 0x00ffffffu) | ((uint32_t)p0 << 24);
                                v[j+1] = (v[j+1] & 0xffffff00u) | (uint32_t)q0;
                            }
                        }
                    }
                }
                // ---------- horizontal edges: rows -> LDS -> columns ----------
                if (E.e[2] | E.e[3]) {
                    if (isY) { uint32_t *r = ty + (lane + 4) * DY_DW; r[0] = v[0]; r[1] = v[1]; r[2] = v[2]; r[3] = v[3]; r[4] = v[4]; }
                    else if (isC) { uint32_t *r = tc_ + ((lane & 7) + 2) * DC_DW; r[0] = v[0]; r[1] = v[1]; r[2] = v[2]; }
                    else if (isTY) { uint32_t *r = ty + (lane - 32) * DY_DW; r[1] = v[1]; r[2] = v[2]; r[3] = v[3]; r[4] = v[4]; }
                    else if (isTC) { uint32_t *r = tc_ + (lane & 1) * DC_DW; r[1] = v[1]; r[2] = v[2]; }
                    wave_lds_fence();
                    if (isY) {
                        uint8_t *colp = (uint8_t *)ty + 4 + lane;       // column `lane`, row r at colp[(r+4)*DY_STRIDE]
                        int c[20];
#pragma unroll
                        for (int r = 0; r < 20; r++) c[r] = colp[r * DY_STRIDE];
                        const int seg = lane >> 2;
#pragma unroll
                        for (int ed = 0; ed < 4; ed++) {
                            const uint32_t nib = E.nib(1, ed);
                            if (nib == 0) continue;
                            const int b = (nib >> (4 * seg)) & 15, first = nib & 15, k = edge_class(1, ed);
                            if (first >= 4 || b) {
                                int p[4] = { c[4*ed+3], c[4*ed+2], c[4*ed+1], c[4*ed] }, q[4] = { c[4*ed+4], c[4*ed+5], c[4*ed+6], c[4*ed+7] };
                                filter_luma(p, q, first >= 4 ? 4 : b, E.ab(k, 0), E.ab(k, 1), first >= 4 ? 0 : E.tc(k, b - 1));
                                c[4*ed+3] = p[0]; c[4*ed+2] = p[1]; c[4*ed+1] = p[2]; c[4*ed+4] = q[0]; c[4*ed+5] = q[1]; c[4*ed+6] = q[2];
                            }
                        }
#pragma unroll
                        for (int r = 1; r < 19; r++) colp[r * DY_STRIDE] = (uint8_t)c[r];
                    } else if (isC) {
                        uint8_t *colp = (uint8_t *)tc_ + 4 + (lane & 7);
                        int c[10];
#pragma unroll
                        for (int r = 0; r < 10; r++) c[r] = colp[r * DC_STRIDE];
                        const int seg = (lane & 7) >> 1;
#pragma unroll
                        for (int ed = 0; ed < 4; ed += 2) {
                            const uint32_t nib = E.nib(1, ed);
                            if (nib == 0) continue;
                            const int b = (nib >> (4 * seg)) & 15, first = nib & 15, k = edge_class(1, ed) + 3;
                            if (first >= 4 || b)
                                filter_chroma(c[2*ed+1], c[2*ed], c[2*ed+2], c[2*ed+3], first >= 4 ? 4 : b, E.ab(k, 0), E.ab(k, 1), first >= 4 ? 0 : E.tc(k, b - 1));
                        }
#pragma unroll
                        for (int r = 1; r < 9; r++) colp[r * DC_STRIDE] = (uint8_t)c[r];
                    }
                    wave_lds_fence();
                    if (isY) { const uint32_t *r = ty + (lane + 4) * DY_DW; v[1] = r[1]; v[2] = r[2]; v[3] = r[3]; v[4] = r[4]; }
                    else if (isC) { const uint32_t *r = tc_ + ((lane & 7) + 2) * DC_DW; v[1] = r[1]; v[2] = r[2]; }
                    else if (isTY) { const uint32_t *r = ty + (lane - 32) * DY_DW; v[1] = r[1]; v[2] = r[2]; v[3] = r[3]; v[4] = r[4]; }
                    else if (isTC) { const uint32_t *r = tc_ + (lane & 1) * DC_DW; v[1] = r[1]; v[2] = r[2]; }
                    wave_lds_fence();
                }
            }
            // everything issued before this point has completed at the release below: the stores of macroblock x-1
            // and the prefetch of x+1.  Publish x-1.
            row_publish(sync, row, x);
            // write macroblock x back: columns -4..11 now, 12..15 (chroma 4..7) with the next macroblock
            if (!EXPD_NOSTORE) {
                const bool last = x + 1 == g.mb_w;
                uint8_t *dst = (uint8_t *)src;
                if (isY) {                                   // naturally aligned pieces only
                    if (x > 0) *(uint32_t *)(dst + x * 16 - 4) = v[0];
                    *(uint2 *)(dst + x * 16) = make_uint2(v[1], v[2]);
                    *(uint32_t *)(dst + x * 16 + 8) = v[3];
                    if (last) *(uint32_t *)(dst + x * 16 + 12) = v[4];
                } else if (isC) {
                    if (x > 0) *(uint32_t *)(dst + x * 8 - 4) = v[0];
                    *(uint32_t *)(dst + x * 8) = v[1];
                    if (last) *(uint32_t *)(dst + x * 8 + 4) = v[2];
                } else if (isTY && fT) *(uint4 *)(dst + x * 16) = make_uint4(v[1], v[2], v[3], v[4]);
                else if (isTC && fT) *(uint2 *)(dst + x * 8) = make_uint2(v[1], v[2]);
            }
            // the right-hand columns become the left-hand columns of the next macroblock
            if (isY) v[0] = v[4]; else if (isC) v[0] = v[2];
        }
        row_publish(sync, row, g.mb_w);             // waits for the last stores of the row
    }
}
