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

struct EdgeClass { uint8_t alpha, beta, tc[3], any, pad[2]; };   // tc = tc0 for bS 1..3 (chroma: already +1)
struct EdgeInfo {                  // 64 bytes per macroblock, written by k_deblock_bs
    uint32_t  bs[4];               // 8 edges x 4 segments x 4 bits: word = dir*2 + (edge>>1), nibble = (edge&1)*4 + seg
    EdgeClass cls[6];              // [class + 3*chroma]; cls[0].any = some bS != 0
};
#define EDGE_DW 16

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
    // lanes 0,1,2 fetch the records of this MB, its left and its top neighbour; everybody gets the few
    // fields it needs by shuffle (width 32 = one macroblock)
    uint4 rec = make_uint4(0, 0, 0, 0);
    {
        const uint4 *recs = (const uint4 *)pd->mb;
        int src = mbi;
        if (lane == 1) src = mbi % g.mb_w > 0 ? mbi - 1 : mbi;
        if (lane == 2) src = mbi >= g.mb_w ? mbi - g.mb_w : mbi;
        if (lane < 3) rec = recs[src];
    }
    const unsigned m0 = __shfl((int)rec.x, 0, 32), mmask = __shfl((int)rec.y, 0, 32), mflags = __shfl((int)rec.w, 0, 32);
    const unsigned l0 = __shfl((int)rec.x, 1, 32), lmask = __shfl((int)rec.y, 1, 32);
    const unsigned t0 = __shfl((int)rec.x, 2, 32), tmask = __shfl((int)rec.y, 2, 32);
    const int m_type = m0 & 255, m_qp = (m0 >> 8) & 255, m_edges = (mflags >> 8) & 255;
    const bool fL = m_edges & P264_EDGE_LEFT, fT = m_edges & P264_EDGE_TOP;
    EdgeInfo *out = info + (size_t)pic * g.n_mb + mbi;

    // ---- boundary strengths, core/frame.c:535-581; lane = dir*16 + edge*4 + segment ----
    int bS = 0;
    {
        const int dir = lane >> 4, e = (lane >> 2) & 3, i = lane & 3;
        const bool outer = e == 0;
        const bool enabled = m_edges && (outer ? (dir == 0 ? fL : fT) : true);
        const int n_type = outer ? ((dir == 0 ? l0 : t0) & 255) : m_type;
        const unsigned n_mask = outer ? (dir == 0 ? lmask : tmask) : mmask;
        const int nbi = outer ? (dir == 0 ? mbi - 1 : mbi - g.mb_w) : mbi;
        if (enabled) {
            if (P264_MB_IS_INTRA(m_type) || P264_MB_IS_INTRA(n_type)) bS = outer ? 4 : 3;
            else {
                int x = dir == 0 ? e : i, y = dir == 0 ? i : e;
                int xn = dir == 0 ? (x - 1) & 3 : x, yn = dir == 0 ? y : (y - 1) & 3;
                if (((mmask >> blk_at(x, y)) & 1) || ((n_mask >> blk_at(xn, yn)) & 1)) bS = 2;
                else {
                    int rp = pd->ref_idx[mbi * 4 + (y >> 1) * 2 + (x >> 1)], rq = pd->ref_idx[nbi * 4 + (yn >> 1) * 2 + (xn >> 1)];
                    int vp = pd->mv[mbi * 16 + y * 4 + x], vq = pd->mv[nbi * 16 + yn * 4 + xn];
                    bS = (rp != rq || abs((int)(int16_t)vp - (int)(int16_t)vq) >= 4 || abs((vp >> 16) - (vq >> 16)) >= 4) ? 1 : 0;
                }
            }
        }
    }
    // pack 8 nibbles per word: lanes 8w .. 8w+7 -> word w; lane 0 writes all four
    uint32_t word = (uint32_t)bS << (4 * (lane & 7));
    word |= __shfl_xor(word, 1); word |= __shfl_xor(word, 2); word |= __shfl_xor(word, 4);
    const uint32_t w1 = __shfl((int)word, 8, 32), w2 = __shfl((int)word, 16, 32), w3 = __shfl((int)word, 24, 32);
    if (lane == 0) *(uint4 *)out->bs = make_uint4(word, w1, w2, w3);
    const bool any = (word | w1 | w2 | w3) != 0;

    // ---- per edge class: alpha, beta, tc0 (deblock_edge, core/frame.c:472-488; offsets unshifted: A-Q3) ----
    if (lane < 6) {
        const int cls = lane % 3, chroma = lane / 3;
        const int qp = m_qp, qpn = cls == EC_LEFT ? (int)((l0 >> 8) & 255) : cls == EC_TOP ? (int)((t0 >> 8) & 255) : m_qp;
        int q;
        if (!chroma) q = (qp + qpn + 1) >> 1;                                     // :593-595
        else q = (c_chroma_qp[clip3i(qp + pd->chroma_qp_offset, 0, 51)] + c_chroma_qp[clip3i(qpn + pd->chroma_qp_offset, 0, 51)] + 1) >> 1;   // :600-601
        const int ia = clip3i(q + pd->alpha_off, 0, 51);
        uint32_t lo = (uint32_t)c_alpha[ia] | ((uint32_t)c_beta[clip3i(q + pd->beta_off, 0, 51)] << 8) |
                      ((uint32_t)(c_tc0[ia][0] + chroma) << 16) | ((uint32_t)(c_tc0[ia][1] + chroma) << 24);
        uint32_t hi = (uint32_t)(c_tc0[ia][2] + chroma) | ((uint32_t)(any ? 1 : 0) << 8);
        *(uint2 *)&out->cls[lane] = make_uint2(lo, hi);
    }
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

// the 16 dwords of a macroblock's EdgeInfo, held per lane (all 16 lanes of a row group hold the same values)
struct EdgeRegs {
    uint32_t e[EDGE_DW];
    __device__ __forceinline__ uint32_t nib(int dir, int ed) const { return (e[dir * 2 + (ed >> 1)] >> ((ed & 1) * 16)) & 0xffffu; }
    // class k occupies dwords 4+2k (alpha, beta, tc[0], tc[1]) and 5+2k (tc[2], any); k is a compile-time constant
    __device__ __forceinline__ int ab(int k, int j) const { return (int)((e[4 + 2 * k] >> (8 * j)) & 255); }
    __device__ __forceinline__ int tc(int k, int b) const
    {
        uint32_t three = (e[4 + 2 * k] >> 16) | (e[5 + 2 * k] << 16);          // tc[0], tc[1], tc[2]; b (0..2) varies per lane
        return (int)((three >> (8 * b)) & 255);
    }
};
__device__ __forceinline__ int edge_class(int dir, int ed) { return ed == 0 ? (dir == 0 ? EC_LEFT : EC_TOP) : EC_INNER; }

// ------------------------------------------------------------------------------------------
// K4b
// ------------------------------------------------------------------------------------------
// A wavefront owns a BAND of four macroblock rows and walks them as a diagonal: in iteration t row
// group g (lanes 16g..16g+15) works on macroblock x = t - 2g, which is exactly the 2-MB lag the
// raster order demands between neighbouring rows.  All four groups execute the same instructions, so
// every filter instruction does useful work in all 64 lanes, and rows inside a band hand their bottom
// samples to the row below through LDS.  Only bands synchronise through the progress counters
// (wavefront_sync.h), with pixels of the band above travelling through global memory.
//
// Per lane, luma view: lane (g,i) holds luma row i of its macroblock as five dwords yl[0..4] = columns
// -4..15 (yl[0] is carried over from the previous macroblock); chroma view: lane (g, p = i>>3, j = i&7)
// holds chroma row j of plane p as cl[0..2] = columns -4..7.  Vertical edges sit on dword boundaries and
// are filtered in these registers.  For horizontal edges the rows go to an LDS tile, every lane picks up
// one column (20 luma / 10 chroma samples), filters it and puts it back.  The four (two) rows above a
// macroblock live in a small LDS ring written by the row group above (or loaded from the band above).
#define RING_SLOTS   4
#define RING_DW      24            // per slot: 4 luma rows x 4 dwords, then 2 planes x 2 rows x 2 dwords
#define TILE_DW      (16 * 4 + 2 * 8 * 2)   // per group: 16 luma rows x 4 dwords, 2 planes x 8 rows x 2 dwords
struct BandLds {                   // per wavefront
    uint32_t tile[4][TILE_DW];
    uint32_t ring[4][RING_SLOTS][RING_DW];
    uint32_t edge[4][EDGE_DW];
};
#define BAND_ROWS 4
#define MAX_BANDS ((MAX_MB_ROWS + BAND_ROWS - 1) / BAND_ROWS)

__global__ __launch_bounds__(ROW_WAVES * 64)
void k_deblock(const PicDev *__restrict__ pics, Geom g_, const EdgeInfo *__restrict__ info, int *status)
{
    __shared__ RowSync sync;                  // progress[band]
    __shared__ BandLds lds[ROW_WAVES];
    const Geom g = g_;
    const PicDev *pd = pics + blockIdx.x;
    if (!pd->deblock) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n_bands = (g.mb_h + BAND_ROWS - 1) / BAND_ROWS;
    rows_init(sync, n_bands);
    BandLds &L = lds[wave];
    uint8_t *Y = pd->dst, *U = pd->dst + g.off_u, *V = pd->dst + g.off_v;
    const EdgeInfo *pinfo = info + (size_t)blockIdx.x * g.n_mb;
    const int grp = lane >> 4, i = lane & 15;                 // row group, line inside the group
    const int cp = i >> 3, cj = i & 7;                        // chroma plane / row of this lane
    uint32_t *tY = L.tile[grp], *tC = L.tile[grp] + 64 + cp * 16;
    bool ok = true;

    for (int band = wave; band < n_bands; band += ROW_WAVES) {
        const int R0 = band * BAND_ROWS;
        const int nrows = min(BAND_ROWS, g.mb_h - R0), last = nrows - 1;
        const int row = R0 + grp;
        const bool have_row = grp < nrows;
        const bool below_in_band = grp < last;                 // the row below is handled by the next group of this wave
        const bool top_exists = row > 0;
        // this lane's luma / chroma row of macroblock 0 of its MB row (clamped for idle groups)
        const int rowc = min(row, g.mb_h - 1);
        uint8_t *srcY = Y + (size_t)(rowc * 16 + i) * g.w;
        uint8_t *srcC = (cp ? V : U) + (size_t)(rowc * 8 + cj) * g.cw;
        // group 0's rows above come from the band above: lanes 0..3 luma rows -4..-1, lanes 4..7 chroma (plane, row)
        uint8_t *srcT = Y;
        if (lane < 4) srcT = Y + (size_t)(R0 * 16 - 4 + lane) * g.w;
        else if (lane < 8) srcT = ((lane >> 1) & 1 ? V : U) + (size_t)(R0 * 8 - 2 + (lane & 1)) * g.cw;
        const bool band_above = R0 > 0;

        uint32_t yl[5] = { 0, 0, 0, 0, 0 }, cl[3] = { 0, 0, 0 };
        uint4 pY = make_uint4(0, 0, 0, 0), pT = make_uint4(0, 0, 0, 0), pE = make_uint4(0, 0, 0, 0);
        uint2 pC = make_uint2(0, 0);
        const int n_iter = g.mb_w + 2 * last;

        // prefetch for iteration t: own rows of x = t - 2*grp, top rows of group 0's macroblock, edge info of all four
        auto prefetch = [&](int t) {
            const int x = t - 2 * grp;
            const bool act = have_row && x >= 0 && x < g.mb_w;
            const int x0 = t;                                   // group 0's macroblock
            if (band_above && x0 < g.mb_w && ok && !EXPD_NOWAIT) ok = row_wait(sync, band - 1, min(x0 + 2, g.mb_w), status);
            if (act) { pY = *(const uint4 *)(srcY + x * 16); pC = *(const uint2 *)(srcC + x * 8); }
            if (band_above && x0 < g.mb_w) {
                if (lane < 4) pT = *(const uint4 *)(srcT + x0 * 16);
                else if (lane < 8) { uint2 v2 = *(const uint2 *)(srcT + x0 * 8); pT.x = v2.x; pT.y = v2.y; }
            }
            if (lane < 16) {                                    // 4 groups x 4 x 16 bytes
                const int eg = lane >> 2, ex = t - 2 * eg, er = R0 + eg;
                pE = make_uint4(0, 0, 0, 0);
                if (eg < nrows && ex >= 0 && ex < g.mb_w) pE = ((const uint4 *)(pinfo + er * g.mb_w + ex))[lane & 3];
            }
        };

        prefetch(0);
        for (int t = 0; t < n_iter; t++) {
            const int x = t - 2 * grp;
            const bool act = have_row && x >= 0 && x < g.mb_w;
            const int slot = x & 3;
            // ---- land what was prefetched for this iteration ----
            if (act) { yl[1] = pY.x; yl[2] = pY.y; yl[3] = pY.z; yl[4] = pY.w; cl[1] = pC.x; cl[2] = pC.y; }
            if (band_above && t < g.mb_w) {
                uint32_t *rg = L.ring[0][t & 3];
                if (lane < 4) { rg[lane * 4] = pT.x; rg[lane * 4 + 1] = pT.y; rg[lane * 4 + 2] = pT.z; rg[lane * 4 + 3] = pT.w; }
                else if (lane < 8) { rg[16 + (lane - 4) * 2] = pT.x; rg[16 + (lane - 4) * 2 + 1] = pT.y; }
            }
            if (lane < 16) { uint32_t *ed = &L.edge[lane >> 2][(lane & 3) * 4]; ed[0] = pE.x; ed[1] = pE.y; ed[2] = pE.z; ed[3] = pE.w; }
            wave_lds_fence();
            if (t + 1 < n_iter) prefetch(t + 1);                // the next iteration's loads start their trip now
            EdgeRegs E;
#pragma unroll
            for (int k = 0; k < EDGE_DW; k++) E.e[k] = act ? L.edge[grp][k] : 0u;
            uint32_t *ringY = L.ring[grp][slot], *ringC = L.ring[grp][slot] + 16 + cp * 4;

            if (!EXPD_NOFILTER) {
                // ---------- vertical edges, in registers (luma: lane = row i; chroma: lane = (plane, row)) ----------
#pragma unroll
                for (int ed = 0; ed < 4; ed++) {
                    const uint32_t nib = E.nib(0, ed);
                    const int b = (nib >> (4 * (i >> 2))) & 15, first = nib & 15, k = edge_class(0, ed);
                    const bool go = first >= 4 || b;
                    if (__ballot(go) == 0) continue;
                    if (go) {
                        int p[4] = { (int)(yl[ed] >> 24), (int)((yl[ed] >> 16) & 255), (int)((yl[ed] >> 8) & 255), (int)(yl[ed] & 255) };
                        int q[4] = { (int)(yl[ed+1] & 255), (int)((yl[ed+1] >> 8) & 255), (int)((yl[ed+1] >> 16) & 255), (int)(yl[ed+1] >> 24) };
                        filter_luma(p, q, first >= 4 ? 4 : b, E.ab(k, 0), E.ab(k, 1), first >= 4 ? 0 : E.tc(k, b - 1));
                        yl[ed] = (uint32_t)p[3] | ((uint32_t)p[2] << 8) | ((uint32_t)p[1] << 16) | ((uint32_t)p[0] << 24);
                        yl[ed+1] = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
                    }
                }
#pragma unroll
                for (int ed = 0; ed < 4; ed += 2) {
                    const uint32_t nib = E.nib(0, ed);
                    const int b = (nib >> (4 * (cj >> 1))) & 15, first = nib & 15, k = edge_class(0, ed) + 3;
                    const bool go = first >= 4 || b;
                    if (__ballot(go) == 0) continue;
                    if (go) {
                        const int j = ed >> 1;
                        int p1 = (int)((cl[j] >> 16) & 255), p0 = (int)(cl[j] >> 24), q0 = (int)(cl[j+1] & 255), q1 = (int)((cl[j+1] >> 8) & 255);
                        filter_chroma(p0, p1, q0, q1, first >= 4 ? 4 : b, E.ab(k, 0), E.ab(k, 1), first >= 4 ? 0 : E.tc(k, b - 1));
                        cl[j] = (cl[j] & 0x00ffffffu) | ((uint32_t)p0 << 24);
                        cl[j+1] = (cl[j+1] & 0xffffff00u) | (uint32_t)q0;
                    }
                }
            }
            // the previous macroblock's right-hand columns are final now: its bottom rows go to the group below
            if (act && below_in_band && x > 0) {
                uint32_t *nr = L.ring[grp + 1][(x - 1) & 3];
                if (i >= 12) nr[(i - 12) * 4 + 3] = yl[0];
                if (cj >= 6) nr[16 + cp * 4 + (cj - 6) * 2 + 1] = cl[0];
            }
            // ---------- horizontal edges: rows -> LDS -> columns -> LDS -> rows ----------
            if (act) {
                tY[i * 4] = yl[1]; tY[i * 4 + 1] = yl[2]; tY[i * 4 + 2] = yl[3]; tY[i * 4 + 3] = yl[4];
                tC[cj * 2] = cl[1]; tC[cj * 2 + 1] = cl[2];
            }
            wave_lds_fence();
            if (!EXPD_NOFILTER) {
                const uint32_t hb = E.e[2] | E.e[3];
                if (__ballot(hb != 0)) {
                    if (hb) {
                        // luma: this lane takes column i
                        const uint8_t *top = (const uint8_t *)ringY + i, *col = (const uint8_t *)tY + i;
                        int c[20];
#pragma unroll
                        for (int r = 0; r < 4; r++) c[r] = top[r * 16];
#pragma unroll
                        for (int r = 0; r < 16; r++) c[4 + r] = col[r * 16];
#pragma unroll
                        for (int ed = 0; ed < 4; ed++) {
                            const uint32_t nib = E.nib(1, ed);
                            const int b = (nib >> (4 * (i >> 2))) & 15, first = nib & 15, k = edge_class(1, ed);
                            if (first >= 4 || b) {
                                int p[4] = { c[4*ed+3], c[4*ed+2], c[4*ed+1], c[4*ed] }, q[4] = { c[4*ed+4], c[4*ed+5], c[4*ed+6], c[4*ed+7] };
                                filter_luma(p, q, first >= 4 ? 4 : b, E.ab(k, 0), E.ab(k, 1), first >= 4 ? 0 : E.tc(k, b - 1));
                                c[4*ed+3] = p[0]; c[4*ed+2] = p[1]; c[4*ed+1] = p[2]; c[4*ed+4] = q[0]; c[4*ed+5] = q[1]; c[4*ed+6] = q[2];
                            }
                        }
                        uint8_t *topw = (uint8_t *)ringY + i, *colw = (uint8_t *)tY + i;
#pragma unroll
                        for (int r = 1; r < 4; r++) topw[r * 16] = (uint8_t)c[r];
#pragma unroll
                        for (int r = 0; r < 15; r++) colw[r * 16] = (uint8_t)c[4 + r];
                        // chroma: this lane takes column cj of plane cp
                        const uint8_t *ctop = (const uint8_t *)ringC + cj, *ccol = (const uint8_t *)tC + cj;
                        int d[10];
                        d[0] = ctop[0]; d[1] = ctop[8];
#pragma unroll
                        for (int r = 0; r < 8; r++) d[2 + r] = ccol[r * 8];
#pragma unroll
                        for (int ed = 0; ed < 4; ed += 2) {
                            const uint32_t nib = E.nib(1, ed);
                            const int b = (nib >> (4 * (cj >> 1))) & 15, first = nib & 15, k = edge_class(1, ed) + 3;
                            if (first >= 4 || b)
                                filter_chroma(d[2*ed+1], d[2*ed], d[2*ed+2], d[2*ed+3], first >= 4 ? 4 : b, E.ab(k, 0), E.ab(k, 1), first >= 4 ? 0 : E.tc(k, b - 1));
                        }
                        uint8_t *ctopw = (uint8_t *)ringC + cj, *ccolw = (uint8_t *)tC + cj;
                        ctopw[8] = (uint8_t)d[1];
#pragma unroll
                        for (int r = 0; r < 7; r++) ccolw[r * 8] = (uint8_t)d[2 + r];
                    }
                    wave_lds_fence();
                    if (act) {
                        yl[1] = tY[i * 4]; yl[2] = tY[i * 4 + 1]; yl[3] = tY[i * 4 + 2]; yl[4] = tY[i * 4 + 3];
                        cl[1] = tC[cj * 2]; cl[2] = tC[cj * 2 + 1];
                    }
                }
            }
            // bottom rows of this macroblock (columns 0..11, plus 12..15 if it is the last of its row) go to the group below
            if (act && below_in_band) {
                uint32_t *nr = L.ring[grp + 1][slot];
                const bool lastmb = x + 1 == g.mb_w;
                if (i >= 12) { uint32_t *r = nr + (i - 12) * 4; r[0] = yl[1]; r[1] = yl[2]; r[2] = yl[3]; if (lastmb) r[3] = yl[4]; }
                if (cj >= 6) { uint32_t *r = nr + 16 + cp * 4 + (cj - 6) * 2; r[0] = cl[1]; if (lastmb) r[1] = cl[2]; }
            }
            // everything issued before this point has completed at the release below: the stores of the previous
            // iteration and the prefetch for the next one.  Publish the last row's progress, one macroblock late.
            {
                const int xl = t - 2 * last;
                if (xl >= 0) row_publish(sync, band, min(xl, g.mb_w));
                else wave_lds_fence();
            }
            // ---- write back ----
            if (act && !EXPD_NOSTORE) {
                const bool lastmb = x + 1 == g.mb_w;
                if (!below_in_band || i < 12) {              // rows 12..15 are written by the group below, as its "rows above"
                    uint8_t *dY = srcY + x * 16;
                    if (x > 0) *(uint32_t *)(dY - 4) = yl[0];
                    *(uint2 *)dY = make_uint2(yl[1], yl[2]);
                    *(uint32_t *)(dY + 8) = yl[3];
                    if (lastmb) *(uint32_t *)(dY + 12) = yl[4];
                }
                if (!below_in_band || cj < 6) {
                    uint8_t *dC = srcC + x * 8;
                    if (x > 0) *(uint32_t *)(dC - 4) = cl[0];
                    *(uint32_t *)dC = cl[1];
                    if (lastmb) *(uint32_t *)(dC + 4) = cl[2];
                }
                if (top_exists) {                             // the rows above this macroblock are final: lines 0..3 luma, 4..7 chroma
                    if (i < 4) {
                        const uint32_t *r = ringY + i * 4;
                        *(uint4 *)(Y + (size_t)(row * 16 - 4 + i) * g.w + x * 16) = make_uint4(r[0], r[1], r[2], r[3]);
                    } else if (i < 8) {
                        const int tp = (i >> 1) & 1, tr = i & 1;
                        const uint32_t *r = L.ring[grp][slot] + 16 + tp * 4 + tr * 2;
                        *(uint2 *)((tp ? V : U) + (size_t)(row * 8 - 2 + tr) * g.cw + x * 8) = make_uint2(r[0], r[1]);
                    }
                }
            }
            yl[0] = yl[4]; cl[0] = cl[2];                     // right-hand columns become the left-hand columns of the next macroblock
            wave_lds_fence();
        }
        row_publish(sync, band, g.mb_w);             // waits for the last stores of the band
    }
}
