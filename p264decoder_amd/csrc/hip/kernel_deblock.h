// kernel_deblock.h - K4: in-loop deblocking filter, in two kernels.
//
// Replaces p264_frame_deblocking_filter + deblock_edge (core/frame.c:472-643) and the four
// sample filters deblock_luma_c / deblock_chroma_c / deblock_luma_intra_c /
// deblock_chroma_intra_c (core/frame.c:302-470).
//
// K4a edge info      - everything about an edge that does not depend on pixels: the 32 boundary
//                      strengths of a macroblock (core/frame.c:535-581) and the averaged QPs of the
//                      edge classes {left, top, inner} x {luma, chroma} (core/frame.c:593-601), 16
//                      bytes per macroblock (EdgeInfo below; alpha / beta / tc0 are expanded by the
//                      consumer, edge_expand).  Fully parallel, one lane per macroblock: as extra
//                      workgroups of the k_intra_sparse launch in batches of P pictures
//                      (kernel_intra.h), as its own launch k_deblock_bs otherwise.
// K4b k_deblock      - the sample filters.  The filter is defined in macroblock raster order
//                      (left edge, inner vertical edges, top edge, inner horizontal edges of one MB
//                      before the next MB) and the result depends on that order: MB (x,y) must see
//                      (x-1,y) and (x+1,y-1) completely filtered ("all vertical, then all
//                      horizontal edges" is NOT bit-exact).  So this is a row wavefront
//                      (wavefront_sync.h).
//
// K4b is issue-bound, not bandwidth-bound (profiles/): the design below is about instructions per
// macroblock.  Eight lanes own one macroblock, every lane filters TWO lines at a time in packed
// 16-bit arithmetic without branches, and a wavefront walks eight macroblock rows as a diagonal.
#pragma once
#include "device_common.h"
#include "wavefront_sync.h"
#ifndef DEBLOCK_WAIT_SLEEP
#define DEBLOCK_WAIT_SLEEP 16
#endif

// Timing experiments (scratch/variant.sh, r4_dbexp.sh): results are wrong unless all defaults hold, so the switches only exist
// in a build that says what it is (-DP264AMD_TIMING_BUILD, see kernel_mc.h and p264hip_build_info()).
#if !defined(P264AMD_TIMING_BUILD) && (defined(EXPD_LUMA_EDGES) || defined(EXPD_CHROMA_EDGES) || defined(EXPD_STRONG) || defined(EXPD_HPASS) || defined(EXPD_BANDSYNC) || defined(EXPD_VMCNT) || defined(EXPD_STAMPS) || defined(EXPD_NO_SAMPLE_LOADS) || defined(EXPD_EDGE_INFO_EDGES))
#error "EXPD_* switches produce wrong pictures: they need -DP264AMD_TIMING_BUILD"
#endif
#ifndef EXPD_LUMA_EDGES
#define EXPD_LUMA_EDGES 4
#endif
#ifndef EXPD_CHROMA_EDGES
#define EXPD_CHROMA_EDGES 4
#endif
#ifndef EXPD_STRONG
#define EXPD_STRONG 1
#endif
#ifndef EXPD_HPASS
#define EXPD_HPASS 1
#endif
#ifndef EXPD_BANDSYNC
#define EXPD_BANDSYNC 1
#endif
#ifndef EXPD_VMCNT
#define EXPD_VMCNT 1
#endif
#ifndef EXPD_EDGE_INFO_EDGES
#define EXPD_EDGE_INFO_EDGES 4     // 1: the edge-info pass looks at the macroblock edges only (round 5: what a cheap road for macroblocks with one vector could save at most)
#endif
#ifndef EXPD_NO_SAMPLE_LOADS
#define EXPD_NO_SAMPLE_LOADS 0     // 1: the macroblock's own samples are not loaded (round 5: what a fused prediction + filter pass could save at most)
#endif
#if defined(EXPD_STAMPS) && !defined(P264HIP_K_DEBLOCK_DECL_ONLY)
// in-kernel clock stamps of one wavefront (diagnostic build only: scratch/r4_stamps.sh)
__device__ unsigned long long g_db_stamps[256 * 8];
#define DB_STAMP(k) do { if (stamp_me && t < 256) g_db_stamps[t * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DB_STAMP(k) do { } while (0)
#endif
#define DY_DW 5                    // luma tile row: 5 dwords = cols -4..15
#define DC_DW 3                    // chroma tile row: 3 dwords = cols -4..7
#define DY_STRIDE (DY_DW * 4)
#define DC_STRIDE (DC_DW * 4)

enum { EC_LEFT = 0, EC_TOP = 1, EC_INNER = 2 };     // edge classes

// What k_deblock_bs hands to k_deblock: 16 bytes per macroblock (64 in rounds 1-3: the class parameters travelled expanded).
//   bs[dir]  two bits per (edge, segment), bit 2 * (4 * edge + segment): the boundary strength, except that on a macroblock
//            edge (edge 0) code 3 stands for strength 4 - strength 3 only exists on inner edges, strength 4 only on
//            macroblock edges (core/frame.c:535-538)
//   qp       own luma QP (= the averaged QP of the inner edges)
//   avg      the averaged QPs of the other five edge classes, six bits each: luma left, luma top, chroma left, chroma
//            top, chroma inner (core/frame.c:593-601)
// alpha / beta / tc0 come out of tables in the consumer (edge_expand below).
struct EdgeInfo { uint32_t bs[2], qp, avg; };
#define EDGE_DW 56                 // per macroblock in LDS: the four words above, then per class [class + 3 * chroma] eight words:
                                   // alpha, beta, -alpha, -beta as 16-bit pairs (the same value in both halves: what the packed filter
                                   // arithmetic takes - round 6: expanded once per QP change instead of two v_perm and two negations
                                   // per edge pass), tc0 of code 1..3 in bytes 1..3 (chroma: already + 1), three words of padding
                                   // (the octets' copies 252 words apart: their banks differ)

// ------------------------------------------------------------------------------------------
// K4a
// ------------------------------------------------------------------------------------------
// One LANE per macroblock.  (The first versions spent 32 lanes per macroblock, one per edge segment: the arithmetic per
// segment is a handful of compares, so shuffling records around and packing nibbles across lanes dominated, ~57 vector
// instructions per macroblock.  With the whole macroblock in one lane's registers - its 16 vectors, the left column and
// the bottom row of the neighbours - everything unrolls at compile time to ~8 per macroblock, and the loads and the
// 16-byte store of neighbouring lanes still cover whole cache lines.)
__device__ __forceinline__ int bs_motion(int vp, int vq, int rp, int rq)
{   // core/frame.c:565-577: different reference or a vector component differing by >= 4 quarter-pels
    const int dx = (int)(int16_t)vp - (int)(int16_t)vq, dy = (vp >> 16) - (vq >> 16);
    return (int)(rp != rq) | (int)(abs(dx) >= 4) | (int)(abs(dy) >= 4);
}

__device__ __forceinline__ bool mv_far(int a, int b)
{
    return abs((int)(int16_t)a - (int)(int16_t)b) >= 4 || abs((a >> 16) - (b >> 16)) >= 4;
}
// B pictures, H.264 8.7.2.1: strength 1 when the two blocks predict from different reference PICTURES or a different number
// of vectors, or when vectors belonging to the same picture differ by >= 4 quarter-pels.  p0 / p1, q0 / q1: the pictures the
// two blocks read through list 0 / list 1 (told apart by their place in the frame store; ~0 = list unused, whose vector is
// zero).  Either the lists correspond as they stand or crossed - with the same picture in both lists both are tried.  (The
// reference's encoder-side loop, core/frame.c:565-577, compares list by list on the indices: the same thing only while no
// picture sits in both lists.  Its decoder never gets here, decoder/macroblock.c:168-171.)
__device__ __forceinline__ int bs_motion_b(uint32_t p0, uint32_t p1, uint32_t q0, uint32_t q1, int vp0, int vp1, int vq0, int vq1)
{
    const bool straight = p0 == q0 && p1 == q1 && !mv_far(vp0, vq0) && !mv_far(vp1, vq1);
    const bool crossed  = p0 == q1 && p1 == q0 && !mv_far(vp0, vq1) && !mv_far(vp1, vq0);
    return !(straight || crossed);
}

// wave-uniform base + 32-bit byte offset per lane: the form global_load / global_store take with the base in scalar registers
// (no 64-bit address per lane)
template <class T> __device__ __forceinline__ const uint8_t *ubase(const T *base, uint32_t byte_off) { return (const uint8_t *)base + byte_off; }
// the value lane - 1 of the wavefront holds (DPP wave_shr:1; lane 0 keeps its own)
__device__ __forceinline__ uint32_t lane_below(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xf, 0xf, false); }

// The edge info of one macroblock from its record, vectors and reference indices plus what it loads of the neighbours.  TWO_LISTS: B pictures in the batch (their motion test:
// bs_motion_b; pic_of = reference index -> picture, both lists, in LDS).
struct PicOf { uint32_t of[2][P264HIP_MAX_REFS]; };
__device__ __forceinline__ void pic_of_init(PicOf &T, const PicDev *pd)
{
    if (threadIdx.x < 2 * P264HIP_MAX_REFS) {
        const int l = threadIdx.x / P264HIP_MAX_REFS, k = threadIdx.x % P264HIP_MAX_REFS;
        T.of[l][k] = l == 0 ? pd->ref_off[k < pd->n_ref ? k : 0] : pd->ref_off_l1[k < pd->n_ref_l1 ? k : 0];
    }
}
// what edge_info_of needs of the macroblock above: the type / QP word and the coded-block mask of its record, the bottom row of
// its vectors, its reference indices.  A caller that walks a column downwards has them in registers from the row before
// (the edge-info role of k_intra_sparse, round 6); everybody else lets edge_info_of load them (top = nullptr).
struct EdgeTop { uint32_t rec_x, rec_y, refs; uint4 mv; };
template <bool TWO_LISTS>
__device__ __forceinline__ uint4 edge_info_of(const PicDev *pd, const Geom &g, int mbi, int mbx, int mby, const uint4 rec,
                                              const uint4 m0, const uint4 m1, const uint4 m2, const uint4 m3, const uint32_t refs, const PicOf *pic_tab,
                                              const EdgeTop *top = nullptr)
{
    const bool b_pic = TWO_LISTS && pd->slice_type == P264_SLICE_B;
    // neighbours (self where there is none: unused).  (ti written so that no select needs the picture width in a vector register)
    const int above = mbi - g.mb_w;
    const int li = mbx > 0 ? mbi - 1 : mbi, ti = above >= 0 ? above : mbi;
    const uint4 *recs = (const uint4 *)pd->mb;
    const int *mvs = pd->mv;
    uint4 recT, mT; uint32_t refsT;
    if (top) { recT = make_uint4(top->rec_x, top->rec_y, 0u, 0u); mT = top->mv; refsT = top->refs; }       // (compile time: the pointer is a constant of the call site)
    else {
        recT = gload4(ubase(recs, (uint32_t)ti * 16u));
        mT = gload4(ubase(mvs, (uint32_t)ti * 64u + 48u));                              // bottom row of the macroblock above
        refsT = gload1(ubase(pd->ref_idx, (uint32_t)ti * 4u));
    }
    // The left macroblock is the lane below's own macroblock (callers: consecutive lanes = consecutive macroblocks, and a lane's
    // left neighbour is active whenever the lane is): its record, the right column of its vectors and its reference indices
    // come out of that lane's registers (DPP wave_shr:1) - as loads they were six requests per lane for bytes that another
    // lane of the same wavefront holds, four of them single dwords of four different 16-byte rows.  Lane 0 of a wavefront has
    // no lane below: it loads.
    const bool first_lane = (threadIdx.x & 63) == 0;
    uint32_t recLx = lane_below(rec.x), recLy = lane_below(rec.y), refsL = lane_below(refs);
    int mL[4] = { (int)lane_below(m0.w), (int)lane_below(m1.w), (int)lane_below(m2.w), (int)lane_below(m3.w) };   // right column of the left one
    if (first_lane) {
        const uint4 r = gload4(ubase(recs, (uint32_t)li * 16u));
        recLx = r.x; recLy = r.y; refsL = gload1(ubase(pd->ref_idx, (uint32_t)li * 4u));
        mL[0] = (int)gload1(ubase(mvs, (uint32_t)li * 64u + 12u)); mL[1] = (int)gload1(ubase(mvs, (uint32_t)li * 64u + 28u));
        mL[2] = (int)gload1(ubase(mvs, (uint32_t)li * 64u + 44u)); mL[3] = (int)gload1(ubase(mvs, (uint32_t)li * 64u + 60u));
    }
    const int mv[16] = { (int)m0.x, (int)m0.y, (int)m0.z, (int)m0.w, (int)m1.x, (int)m1.y, (int)m1.z, (int)m1.w,
                         (int)m2.x, (int)m2.y, (int)m2.z, (int)m2.w, (int)m3.x, (int)m3.y, (int)m3.z, (int)m3.w };
    const int mvTop[4] = { (int)mT.x, (int)mT.y, (int)mT.z, (int)mT.w };
    // list 1 of a B picture (everything zero otherwise: the extra test below is then always false)
    int mv1[16] = { 0 }, mv1Top[4] = { 0 }, mv1L[4] = { 0 };
    uint32_t refs1 = 0, refs1L = 0, refs1T = 0;
    // pictures per 8x8 quadrant: own (4), right column of the left macroblock (quadrants 1, 3), bottom row of the one above (2, 3)
    uint32_t own0[4] = { 0 }, own1[4] = { 0 }, lft0[2] = { 0 }, lft1[2] = { 0 }, top0[2] = { 0 }, top1[2] = { 0 };
    if (b_pic) {
        const int *m1 = pd->mv_l1;
        const uint4 a0 = gload4(ubase(m1, (uint32_t)mbi * 64u)), a1 = gload4(ubase(m1, (uint32_t)mbi * 64u + 16u)), a2 = gload4(ubase(m1, (uint32_t)mbi * 64u + 32u)), a3 = gload4(ubase(m1, (uint32_t)mbi * 64u + 48u));
        const uint4 aT = gload4(ubase(m1, (uint32_t)ti * 64u + 48u));
        refs1 = gload1(ubase(pd->ref_idx_l1, (uint32_t)mbi * 4u)); refs1T = gload1(ubase(pd->ref_idx_l1, (uint32_t)ti * 4u));
        // (the left macroblock's list-1 data: as above - every lane of a B picture's wavefront is here, b_pic is wave-uniform)
        mv1L[0] = (int)lane_below(a0.w); mv1L[1] = (int)lane_below(a1.w); mv1L[2] = (int)lane_below(a2.w); mv1L[3] = (int)lane_below(a3.w);
        refs1L = lane_below(refs1);
        if (first_lane) {
            mv1L[0] = (int)gload1(ubase(m1, (uint32_t)li * 64u + 12u)); mv1L[1] = (int)gload1(ubase(m1, (uint32_t)li * 64u + 28u));
            mv1L[2] = (int)gload1(ubase(m1, (uint32_t)li * 64u + 44u)); mv1L[3] = (int)gload1(ubase(m1, (uint32_t)li * 64u + 60u));
            refs1L = gload1(ubase(pd->ref_idx_l1, (uint32_t)li * 4u));
        }
        const int t[16] = { (int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w,
                            (int)a2.x, (int)a2.y, (int)a2.z, (int)a2.w, (int)a3.x, (int)a3.y, (int)a3.z, (int)a3.w };
#pragma unroll
        for (int i = 0; i < 16; i++) mv1[i] = t[i];
        mv1Top[0] = (int)aT.x; mv1Top[1] = (int)aT.y; mv1Top[2] = (int)aT.z; mv1Top[3] = (int)aT.w;
        auto pics_of = [&](uint32_t r0, uint32_t r1, int q, uint32_t &a, uint32_t &b) {
            const int i0 = (int)(int8_t)(r0 >> (8 * q)), i1 = (int)(int8_t)(r1 >> (8 * q));
            a = i0 < 0 ? ~0u : pic_tab->of[0][i0 & (P264HIP_MAX_REFS - 1)];
            b = i1 < 0 ? ~0u : pic_tab->of[1][i1 & (P264HIP_MAX_REFS - 1)];
            if (i0 < 0 && i1 < 0) a = pic_tab->of[0][0];          // (no list at all: list 0, entry 0, as the motion compensation reads it)
        };
#pragma unroll
        for (int q = 0; q < 4; q++) pics_of(refs, refs1, q, own0[q], own1[q]);
        pics_of(refsL, refs1L, 1, lft0[0], lft1[0]); pics_of(refsL, refs1L, 3, lft0[1], lft1[1]);
        pics_of(refsT, refs1T, 2, top0[0], top1[0]); pics_of(refsT, refs1T, 3, top0[1], top1[1]);
    }

    const int m_type = rec.x & 255, m_qp = (rec.x >> 8) & 255, m_edges = (rec.w >> 8) & 255;
    const unsigned mmask = rec.y, lmask = recLy, tmask = recT.y;
    const bool m_intra = P264_MB_IS_INTRA(m_type), l_intra = P264_MB_IS_INTRA(recLx & 255), t_intra = P264_MB_IS_INTRA(recT.x & 255);
    const bool fL = m_edges & P264_EDGE_LEFT, fT = m_edges & P264_EDGE_TOP;
    auto ref_of = [](uint32_t r4, int x, int y) { return (int)((r4 >> (8 * ((y >> 1) * 2 + (x >> 1)))) & 255); };

    // ---- boundary strengths, core/frame.c:535-581 ----
    uint32_t word[2] = { 0, 0 };
    // (the top edge first: what it needs of the macroblock above - four vectors, indices, mask - is dead after four segments instead of
    // alive through all thirty-two; the edge-info role of the 64-register k_intra_sparse build has no register to spare)
#pragma unroll
    for (int pass = 0; pass < 2; pass++)
#pragma unroll
    for (int dir = 0; dir < 2; dir++)
#pragma unroll
        for (int e = 0; e < EXPD_EDGE_INFO_EDGES; e++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if ((pass == 0) != (dir == 1 && e == 0)) continue;
                const int x = dir == 0 ? e : i, y = dir == 0 ? i : e;
                const int xn = dir == 0 ? (x + 3) & 3 : x, yn = dir == 0 ? y : (y + 3) & 3;
                const bool outer = e == 0;
                const bool enabled = m_edges && (outer ? (dir == 0 ? fL : fT) : true);
                const bool n_intra = outer ? (dir == 0 ? l_intra : t_intra) : m_intra;
                const unsigned n_mask = outer ? (dir == 0 ? lmask : tmask) : mmask;
                const int vq = mv[y * 4 + x], vp = !outer ? mv[yn * 4 + xn] : dir == 0 ? mL[y] : mvTop[x];
                const int rq = ref_of(refs, x, y), rp = ref_of(!outer ? refs : dir == 0 ? refsL : refsT, xn, yn);
                int bS = bs_motion(vp, vq, rp, rq);
                if (TWO_LISTS && b_pic) {
                    const int wq = mv1[y * 4 + x], wp = !outer ? mv1[yn * 4 + xn] : dir == 0 ? mv1L[y] : mv1Top[x];
                    const int qq = (y >> 1) * 2 + (x >> 1), qn = (yn >> 1) * 2 + (xn >> 1);
                    const uint32_t p0 = !outer ? own0[qn] : dir == 0 ? lft0[y >> 1] : top0[x >> 1], p1 = !outer ? own1[qn] : dir == 0 ? lft1[y >> 1] : top1[x >> 1];
                    bS = bs_motion_b(p0, p1, own0[qq], own1[qq], vp, wp, vq, wq);
                }
                if (((mmask >> blk_at(x, y)) | (n_mask >> blk_at(xn, yn))) & 1) bS = 2;
                if (m_intra | n_intra) bS = 3;                    // on a macroblock edge: the code of strength 4
                if (!enabled) bS = 0;
                word[dir] |= (uint32_t)bS << (2 * (4 * e + i));
            }

    // ---- averaged QPs per edge class (deblock_edge, core/frame.c:472-488,593-601) ----
    const int cqo = pd->chroma_qp_offset;
    const int qpL = (int)((recLx >> 8) & 255), qpT = (int)((recT.x >> 8) & 255);
    const int cq_own = chroma_qp(clip3i(m_qp + cqo, 0, 51));
    const uint32_t avg = (uint32_t)((m_qp + qpL + 1) >> 1) | (uint32_t)((m_qp + qpT + 1) >> 1) << 6
                       | (uint32_t)((cq_own + chroma_qp(clip3i(qpL + cqo, 0, 51)) + 1) >> 1) << 12
                       | (uint32_t)((cq_own + chroma_qp(clip3i(qpT + cqo, 0, 51)) + 1) >> 1) << 18 | (uint32_t)cq_own << 24;
    return make_uint4(word[0], word[1], (uint32_t)m_qp, avg);
}

// One lane per macroblock.  (Round 4 also had the work-list sort of the MC stage write the edge info while it holds the same
// arrays in registers - scratch/r4_fuse/: the sort went from 0.32 to 0.49 ms (112 bytes per lane of spills), this kernel's
// 0.19 ms went away: -0.03 ms per step, and 0.16 ms of loop-filter work booked on the MC stage.  Not adopted.)
template <bool TWO_LISTS>
__global__ __launch_bounds__(256)
void k_deblock_bs(const PicDev *__restrict__ pics, Geom g, EdgeInfo *__restrict__ info, uint32_t inv_mbw)
{
    const PicDev *pd = pics + blockIdx.y;
    if (!pd->deblock) return;
    __shared__ PicOf pic_tab;
    if (TWO_LISTS) {
        if (pd->slice_type == P264_SLICE_B) pic_of_init(pic_tab, pd);
        __syncthreads();
    }
    const int mbi = blockIdx.x * 256 + threadIdx.x;
    if (mbi >= g.n_mb) return;
    int mby = (int)__umulhi((unsigned)mbi, inv_mbw);
    if (mbi - mby * g.mb_w >= g.mb_w) mby++;
    const int mbx = mbi - mby * g.mb_w;
    const uint4 rec = gload4(ubase(pd->mb, (uint32_t)mbi * 16u));
    const int *mvs = pd->mv;
    const uint4 m0 = gload4(ubase(mvs, (uint32_t)mbi * 64u)), m1 = gload4(ubase(mvs, (uint32_t)mbi * 64u + 16u)), m2 = gload4(ubase(mvs, (uint32_t)mbi * 64u + 32u)), m3 = gload4(ubase(mvs, (uint32_t)mbi * 64u + 48u));
    const uint32_t refs = gload1(ubase(pd->ref_idx, (uint32_t)mbi * 4u));
    gstore4((uint8_t *)(info + (size_t)blockIdx.y * g.n_mb) + (uint32_t)mbi * 16u, edge_info_of<TWO_LISTS>(pd, g, mbi, mbx, mby, rec, m0, m1, m2, m3, refs, &pic_tab));
}

// alpha | tc0 of strengths 1..3 (one dword per index A) and beta (per index B): core/frame.c:262-291
struct EdgeTables { uint32_t alpha_tc[52]; uint8_t beta[52]; };
__device__ __forceinline__ void edge_tables_init(EdgeTables &T)
{
    if (threadIdx.x < 52) {
        const int i = threadIdx.x;
        T.alpha_tc[i] = (uint32_t)c_alpha[i] | ((uint32_t)c_tc0[i][0] << 8) | ((uint32_t)c_tc0[i][1] << 16) | ((uint32_t)c_tc0[i][2] << 24);
        T.beta[i] = c_beta[i];
    }
}
// Class k = class + 3 * chroma of a macroblock: alpha, beta, tc0 (deblock_edge, core/frame.c:472-488; offsets unshifted:
// A-Q3).  k may vary per lane.  On the macroblock-edge classes code 3 means strength 4, which has no tc0: byte 3 = 0.
struct EdgeClass { uint4 ab; uint32_t tc; };         // {alpha, beta, -alpha, -beta} as pairs; tc0 bytes
__device__ __forceinline__ EdgeClass edge_expand(const EdgeTables &T, uint32_t qp, uint32_t avg, int k, int alpha_off, int beta_off)
{
    const int sh = k < 2 ? 6 * k : 6 * (k - 1);                        // k: 0 1 [2] 3 4 5 -> field 0 1 [qp] 2 3 4 of avg
    const int q = k == 2 ? (int)(qp & 63u) : (int)((avg >> sh) & 63u);
    const uint32_t at = T.alpha_tc[clip3i(q + alpha_off, 0, 51)];
    const uint32_t be = T.beta[clip3i(q + beta_off, 0, 51)];
    uint32_t v = at + (k >= 3 ? 0x01010100u : 0u);                     // chroma: tc0 + 1 (no carries: tc0 <= 25)
    if (k != 2 && k != 5) v &= 0x00ffffffu;
    const uint32_t a2 = (v & 0xffu) * 0x00010001u, b2 = be * 0x00010001u;
    EdgeClass c;
    c.ab = make_uint4(a2, b2, 0u - a2 + ((a2 & 0xffffu) ? 0x00010000u : 0u), 0u - b2 + ((b2 & 0xffffu) ? 0x00010000u : 0u));   // (per-half negation: the low half's borrow given back)
    c.tc = v & 0xffffff00u;
    return c;
}

// A macroblock's EdgeInfo as its eight lanes see it: the two boundary-strength words in registers, the class
// parameters fetched from the octet's LDS copy where an edge needs them.
struct EdgeRegs {
    uint32_t e[2];
    const uint32_t *lds;            // the octet's copy: 4 raw words, then 6 classes x 8 words
    // code of (edge ed, the lane's segment): seg2 = 2 * segment
    __device__ __forceinline__ int code(int dir, int ed, int seg2) const { return (int)__builtin_amdgcn_ubfe(e[dir], (unsigned)(8 * ed + seg2), 2u); }
};
// class k occupies dwords 4+8k .. 8+8k: alpha, beta, -alpha, -beta as 16-bit pairs with the same value in both halves (the filter
// arithmetic is packed), then (0, tc0 of code 1..3); k is a compile-time constant.
struct EdgeParams {
    uint4 ab; uint32_t hi;
    __device__ __forceinline__ EdgeParams(const EdgeRegs &E, int k) { ab = *(const uint4 *)(E.lds + 4 + 8 * k); hi = E.lds[8 + 8 * k]; }
    __device__ __forceinline__ uint32_t alpha2() const { return ab.x; }
    __device__ __forceinline__ uint32_t beta2() const { return ab.y; }
    __device__ __forceinline__ uint32_t nalpha2() const { return ab.z; }
    __device__ __forceinline__ uint32_t nbeta2() const { return ab.w; }
    // tc0 of code c (per lane; 0 for code 0 and for strength 4, which does not use it): byte c of {0, hi}
    __device__ __forceinline__ uint32_t tc2(int c) const { return perm(0u, hi, 0x0c000c00u + (uint32_t)c * 0x00010001u); }
};
// all ones where the boundary strength is 1..3 / is 4, from the code (one sign-extending bit-field extract instead of compare +
// select): macroblock edges (ed == 0) hold strengths {0, 1, 2, 4 as code 3}, inner edges {0, 1, 2, 3}
__device__ __forceinline__ uint32_t mask_bs123(int c, int ed) { return (uint32_t)__builtin_amdgcn_sbfe(ed == 0 ? 0x06 : 0x0e, (unsigned)c, 1u); }
__device__ __forceinline__ uint32_t mask_bs4(int c) { return (uint32_t)__builtin_amdgcn_sbfe(0x08, (unsigned)c, 1u); }     // ed == 0 only
__device__ __forceinline__ int edge_class(int dir, int ed) { return ed == 0 ? (dir == 0 ? EC_LEFT : EC_TOP) : EC_INNER; }

// ------------------------------------------------------------------------------------------
// sample filters on two lines at once: every value is a pair of 16-bit lanes (v_pk_*_i16)
// ------------------------------------------------------------------------------------------
typedef short pk16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk16 as_pk(uint32_t v) { return __builtin_bit_cast(pk16, v); }
__device__ __forceinline__ uint32_t as_u(pk16 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ pk16 pk_splat(int v) { return as_pk((uint32_t)v | ((uint32_t)v << 16)); }    // 0 <= v < 65536
__device__ __forceinline__ pk16 pk_min(pk16 a, pk16 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ pk16 pk_max(pk16 a, pk16 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ pk16 pk_clamp(pk16 v, pk16 lo, pk16 hi) { return pk_min(pk_max(v, lo), hi); }
__device__ __forceinline__ pk16 pk_absd(pk16 a, pk16 b) { pk16 d = a - b; return pk_max(d, -d); }
__device__ __forceinline__ pk16 pk_lt(pk16 a, pk16 b) { return (a - b) >> (pk16)15; }                  // a < b ? 0xffff : 0
__device__ __forceinline__ pk16 pk_sel(pk16 m, pk16 a, pk16 b) { return (a & m) | (b & ~m); }            // v_bfi_b32
// a + b where both halves of both operands are non-negative and the sums stay below 65536 (samples and sums of a few samples):
// one 32-bit add - nothing carries from the low half into the high one.  v_add_u32 issues in 2.4 cycles per wavefront on
// gfx950, v_pk_add_u16 in 4.3 (scratch/r4_rates/).
__device__ __forceinline__ pk16 pk_addu(pk16 a, pk16 b) { return as_pk(as_u(a) + as_u(b)); }

// Sample conditions as SIGN BITS (round 5: the kernel is bound by vector-instruction issue, so the filter arithmetic counts):
// |d| < T  <=>  d < T and -d < T  <=>  (d - T) & (-T - d) is negative - two subtractions on the difference the filter needs anyway,
// no absolute value, and the three conditions of the sample flag share ONE shift (the AND of their sign words).
__device__ __forceinline__ pk16 pk_within(pk16 d, pk16 T, pk16 nT) { return (d - T) & (nT - d); }     // negative where |d| < T (nT = -T)
__device__ __forceinline__ pk16 pk_sign(pk16 v) { return v >> (pk16)15; }                             // all ones where negative
// the three "filterSamplesFlag" conditions, core/frame.c:311,357,398,444; e = q0 - p0 (what the filters start from)
__device__ __forceinline__ pk16 pk_edge_flag(pk16 p1, pk16 p0, pk16 q0, pk16 q1, pk16 A, pk16 nA, pk16 B, pk16 nB, pk16 &e)
{
    e = q0 - p0;
    return pk_sign(pk_within(e, A, nA) & pk_within(p1 - p0, B, nB) & pk_within(q1 - q0, B, nB));
}

// bS 1..3 on a luma edge (core/frame.c:302-341).  f = sample flag in the lanes whose bS is 1..3; e = q0 - p0.  p0 and q0 come out
// NOT clipped (p0 + delta, q0 - delta): the caller clips them where it turns them into bytes (v_sat_pk_u8_i16: one instruction
// instead of max + min).  The masks go into the clamp bounds instead of onto the clamped value (a 32-bit AND issues in 2.4 cycles,
// the packed forms in 4.3).
__device__ __forceinline__ void pk_luma_normal(pk16 p2, pk16 &p1, pk16 &p0, pk16 &q0, pk16 &q1, pk16 q2,
                                               pk16 e, pk16 f, pk16 ap, pk16 aq, pk16 T0)
{
    const pk16 avg = pk_addu(pk_addu(p0, q0), (pk16)1) >> (pk16)1;
    const pk16 dp = pk_clamp((pk_addu(p2, avg) >> (pk16)1) - p1, -T0, T0) & (f & ap);
    const pk16 dq = pk_clamp((pk_addu(q2, avg) >> (pk16)1) - q1, -T0, T0) & (f & aq);
    const pk16 tc = (T0 - ap - aq) & f;                             // the masks are -1 where true; 0 where the flag is off: delta = 0
    const pk16 delta = pk_clamp(((e << (pk16)2) + (p1 - q1) + (pk16)4) >> (pk16)3, -tc, tc);
    p1 += dp; q1 += dq;
    p0 = p0 + delta;
    q0 = q0 - delta;
}
// bS 4 on a luma edge (core/frame.c:387-432); s = f & str in the lanes that take it
__device__ __forceinline__ void pk_luma_strong(pk16 p3, pk16 &p2, pk16 &p1, pk16 &p0, pk16 &q0, pk16 &q1, pk16 &q2, pk16 q3,
                                               pk16 s, pk16 ap, pk16 aq, pk16 A)
{
    const pk16 small = pk_lt(pk_absd(p0, q0), (A >> (pk16)2) + (pk16)2);
    const pk16 sp = small & ap, sq = small & aq;
    // (sums of samples: 32-bit adds, pk_addu; the largest, eight samples + 4, stays below 2048)
    const pk16 pq = pk_addu(p0, q0), p11 = pk_addu(p1, p1), q11 = pk_addu(q1, q1), pq2 = pk_addu(pk_addu(pq, pq), (pk16)4);
    const pk16 c2 = (pk16)2;
    const pk16 p0w = pk_addu(pk_addu(p11, p0), pk_addu(q1, c2)) >> (pk16)2, q0w = pk_addu(pk_addu(q11, q0), pk_addu(p1, c2)) >> (pk16)2;
    const pk16 p0s = pk_addu(pk_addu(p2, p11), pk_addu(pq2, q1)) >> (pk16)3, q0s = pk_addu(pk_addu(p1, pq2), pk_addu(q11, q2)) >> (pk16)3;
    const pk16 p1s = pk_addu(pk_addu(p2, p1), pk_addu(pq, c2)) >> (pk16)2, q1s = pk_addu(pk_addu(pq, q1), pk_addu(q2, c2)) >> (pk16)2;
    const pk16 p22 = pk_addu(p2, p2), q22 = pk_addu(q2, q2);
    const pk16 p2s = pk_addu(pk_addu(pk_addu(p3, p3), pk_addu(p22, p2)), pk_addu(pk_addu(p1, pq), (pk16)4)) >> (pk16)3;
    const pk16 q2s = pk_addu(pk_addu(pk_addu(q3, q3), pk_addu(q22, q2)), pk_addu(pk_addu(q1, pq), (pk16)4)) >> (pk16)3;
    p0 = pk_sel(s, pk_sel(sp, p0s, p0w), p0); q0 = pk_sel(s, pk_sel(sq, q0s, q0w), q0);
    p1 = pk_sel(s & sp, p1s, p1); q1 = pk_sel(s & sq, q1s, q1);
    p2 = pk_sel(s & sp, p2s, p2); q2 = pk_sel(s & sq, q2s, q2);
}
// chroma edge, any bS (core/frame.c:351-377, 438-462); T = tc0+1 (from K4a), en/str = masks of the bS 1..3 / bS 4 lanes
// any_strong (wave-uniform): some lane of the wavefront has bS 4 on this edge.  e = q0 - p0; p0 / q0 come out not clipped (as above).
__device__ __forceinline__ void pk_chroma(pk16 p1, pk16 &p0, pk16 &q0, pk16 q1, pk16 e, pk16 f, pk16 en, pk16 str, bool any_strong, pk16 T)
{
    const pk16 Tm = T & (f & en);
    const pk16 delta = pk_clamp(((e << (pk16)2) + (p1 - q1) + (pk16)4) >> (pk16)3, -Tm, Tm);
    const pk16 op0 = p0, oq0 = q0;
    p0 = p0 + delta;
    q0 = q0 - delta;
    if (any_strong) {
        const pk16 p0w = pk_addu(pk_addu(pk_addu(p1, p1), op0), pk_addu(q1, (pk16)2)) >> (pk16)2, q0w = pk_addu(pk_addu(pk_addu(q1, q1), oq0), pk_addu(p1, (pk16)2)) >> (pk16)2;
        const pk16 s = f & str;
        p0 = pk_sel(s, p0w, p0);
        q0 = pk_sel(s, q0w, q0);
    }
}
// a pair clipped to 0..255 and squeezed into bytes 0 (low half) and 1 (high half): what the byte-wise consumers take
__device__ __forceinline__ uint32_t pk_clip_bytes(pk16 v) { return sat_pk_u8_i16(as_u(v)); }
template <int OFF> __device__ __forceinline__ void lds_put_clipped(uint8_t *base, pk16 v) { *(uint16_t *)(base + OFF) = (uint16_t)pk_clip_bytes(v); }

// A 16-bit pair of samples (values 0..255 in both halves) stored as two adjacent bytes of LDS: ds_write_b8 takes bits 0..7,
// ds_write_b8_d16_hi bits 16..23 - no v_perm to squeeze the pair into 16 bits first (the kernel is bound by vector-instruction
// issue, the LDS pipe is not).  A wavefront's DS operations execute in order, and the extra entries in lgkmcnt only make the
// compiler's own waits stricter.
template <int OFF> __device__ __forceinline__ void lds_put_pair(const uint8_t *base, pk16 v)
{
    const uint32_t a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)base;
    asm volatile("ds_write_b8 %0, %1 offset:%2\n\tds_write_b8_d16_hi %0, %1 offset:%3" :: "v"(a), "v"(as_u(v)), "n"(OFF), "n"(OFF + 1) : "memory");
}
// byte K of two dwords as a 16-bit pair (a -> low half, b -> high half)
template <int K> __device__ __forceinline__ pk16 pair_byte(uint32_t a, uint32_t b)
{
    return as_pk(__builtin_amdgcn_perm(b, a, 0x0c040c00u + (uint32_t)K * 0x00010001u));
}

// ------------------------------------------------------------------------------------------
// K4b
// ------------------------------------------------------------------------------------------
// Lanes: an OCTET of eight lanes owns one macroblock; lane j of the octet holds
//   * luma rows 2j, 2j+1 as ya[0..4] / yb[0..4] = columns -4..15 (column group 0 is carried over from
//     the previous macroblock), and
//   * rows 2(j&3), 2(j&3)+1 of chroma plane j>>2 as ca[0..2] / cb[0..2] = columns -4..7.
// Vertical edges sit on dword boundaries: the lane unpacks the bytes either side into 16-bit pairs
// (row 2j in the low half, 2j+1 in the high half - both in the same 4-line bS segment), filters both
// rows with one stream of v_pk instructions and no branches, and packs them back.  For horizontal edges
// the rows go to an LDS tile and come back as column pairs (lane j: luma columns 2j, 2j+1 over the 4
// rows above + 16 rows; chroma columns 2(j&3), +1 of plane j>>2), again two lines per instruction.
//
// A wavefront's eight octets are `8 >> rb_log2` pictures x `1 << rb_log2` consecutive macroblock rows
// (a BAND).  Octet g of a band works on macroblock x = t - DB_LAG * g in iteration t (DB_LAG = 1: the lag
// the raster order demands once an iteration runs its vertical edges before its horizontal ones, see DB_LAG), and hands the bottom rows of a finished macroblock to the octet below
// through an LDS ring.  Bands synchronise through progress counters in LDS (wavefront_sync.h); the
// pixels of the band above travel through global memory (same CU, same L1).
//
// Every sample is loaded once and stored once, both as whole 16-byte (8-byte chroma) rows: the store
// of macroblock x-1 waits one iteration for its last four columns, which the left edge of x changes.
// Columns a row of a band lags behind the row above it.  The raster order needs MB (x, y) after (x - 1, y) and after the LEFT EDGE
// of (x + 1, y - 1) - which changes the last columns of (x, y - 1), rows 12 - 15 included - but not after the rest of (x + 1, y - 1):
// an iteration filters all vertical edges first (registers), hands the finished rows 12 - 15 of macroblock x - 1 to the octet
// below, and only then runs the horizontal edges, whose top edge is the first thing to read those rows.  So the octet below can
// work on macroblock x - 1 of its row in the SAME iteration: a lag of one column per row, not two (rounds 1 - 5 ran with two:
// 2 x 67 + 120 = 254 dependent iterations down a 1080p picture instead of 67 + 120 = 187 - what a picture costs where it has a
// CU to itself: 0.98 -> 0.7x ms at 256 pictures per launch and for the single picture of the drop-in API).
#ifndef DB_LAG
#define DB_LAG 1
#endif
#define RING_SLOTS   4
#define RING_DW      24            // per slot: 4 luma rows x 4 dwords, then 2 planes x 2 rows x 2 dwords
#define TILE_DW      100           // 16 luma rows x 4 dwords, 2 planes x 8 rows x 2 dwords, +4 so that octets land on different banks
struct OctLds {                    // per octet: 252 dwords (= 28 modulo 32: the eight octets of a wavefront start on eight different banks)
    uint32_t tile[TILE_DW];
    uint32_t ring[RING_SLOTS][RING_DW];
    uint32_t edge[EDGE_DW];
};
#define MAX_PICS_PER_WG 16             // pictures one workgroup serves (in groups of as many as a wavefront holds)
#define MAX_BANDS (MAX_MB_ROWS / 2)

// The kernel is DEFINED in a translation unit of its own (k_deblock.hip, round 6) and only declared where it is launched
// (p264hip.hip defines P264HIP_K_DEBLOCK_DECL_ONLY): that unit is compiled with the backend's max-ILP scheduling strategy
// (-mllvm -amdgpu-sched-strategy=max-ilp, build.py), which suits this kernel - one long dependent iteration per wavefront, no
// spills at 115 registers: 0.833 -> 0.803 ms at 256 pictures per launch - and none of the others (k_intra_sparse spills under it,
// 0.79 -> 0.91 ms; k_mc is unchanged).  The option is per compilation, there is no per-function form a HIP source can spell.
#ifdef P264HIP_K_DEBLOCK_DECL_ONLY
__global__ void k_deblock(const PicDev *__restrict__ pics, Geom g_, const EdgeInfo *__restrict__ info, int *status, int n_pics, int rb_log2_, int pics_per_wg, int odd_single);
#else
__global__ __launch_bounds__(ROW_WAVES * 64)
void k_deblock(const PicDev *__restrict__ pics, Geom g_, const EdgeInfo *__restrict__ info, int *status, int n_pics, int rb_log2_, int pics_per_wg, int odd_single)
{
    __shared__ int progress[MAX_PICS_PER_WG][MAX_BANDS];     // fully stored macroblocks of a band's last row
    __shared__ OctLds lds[ROW_WAVES][8];
    __shared__ EdgeTables tables;
    edge_tables_init(tables);
    const Geom g = g_;
    // (the wavefront number is a scalar: without readfirstlane the compiler takes the unit loop for divergent and keeps all of its
    // book-keeping - unit, band, picture pointers - in vector registers)
    const int wave = rfl((int)(threadIdx.x >> 6)), n_waves = blockDim.x >> 6, lane = threadIdx.x & 63;
    // A work unit = (band, group of pictures): `1 << rb_log2` rows of `8 >> rb_log2` pictures.  odd_single (with bands of 4 rows and
    // an odd number of pictures, round 5): the pictures go in pairs and the LAST one on its own in bands of 8 rows - a pair's
    // group with one picture missing would issue every instruction of its 17 units for half the lanes.
    const int n_bands_all = (g.mb_h + (1 << rb_log2_) - 1) >> rb_log2_;
    const int n_groups_all = (pics_per_wg + (8 >> rb_log2_) - 1) / (8 >> rb_log2_);
    const int n_pairs = pics_per_wg >> 1, n_bands8 = (g.mb_h + 7) >> 3, per_step = 2 * n_pairs + 1;     // odd_single: units per two pair bands + one single band
    const int n_units = !odd_single ? n_bands_all * n_groups_all : (n_bands8 - 1) * per_step + (n_bands_all - 2 * (n_bands8 - 1)) * n_pairs + 1;
    for (int k = threadIdx.x; k < MAX_PICS_PER_WG * MAX_BANDS; k += blockDim.x) (&progress[0][0])[k] = 0;
    __syncthreads();
    bool ok = true;

    // units in band-major order: a wavefront takes unit u only after u - n_waves, and band b of a group only waits for band
    // b - 1 of the same group, which is an earlier unit - nobody waits for a unit that has not been started
    for (int unit = wave; unit < n_units; unit += n_waves) {
        // the unit's shape, band and first picture (scalars).  odd_single: steps of {band 2s of every pair, band s of the single
        // picture, band 2s + 1 of every pair} - a band still only waits for the band above it of the same pictures, an earlier unit
        int rb_log2 = rb_log2_, band, pic0;
        if (!odd_single) { band = unit / n_groups_all; pic0 = (unit - band * n_groups_all) * (8 >> rb_log2_); }
        else {
            const int st = unit / per_step, r = unit - st * per_step;
            if (r < n_pairs) { band = 2 * st; pic0 = 2 * r; }
            else if (r == n_pairs) { rb_log2 = 3; band = st; pic0 = pics_per_wg - 1; }
            else { band = 2 * st + 1; pic0 = 2 * (r - n_pairs - 1); }
        }
        const int RB = 1 << rb_log2, PW = 8 >> rb_log2;         // rows of the band, pictures of the wavefront
        // Everything that follows from the lane number is derived again per unit (a few dozen instructions against ~100 000 of the
        // unit): hoisted out of this loop, the lane constants that only the unit's set-up needs (64-bit offsets of the lane's rows,
        // products with the strip sizes) stayed alive through the iteration loop - the kernel has no register to spare for them
        // and spilled 17.
        int lane_u = lane;
        asm volatile("" : "+v"(lane_u));
        const int o = lane_u >> 3, j = lane_u & 7;
        const int gr = o & (RB - 1), pi = o >> rb_log2;           // row inside the band, picture inside the wavefront
        OctLds &L = lds[wave][o];
        const int seg2 = (j >> 1) * 2, cseg2 = (j & 3) * 2;       // 2 x the bS segment of this lane's luma / chroma lines
        const int cp = j >> 2, cr = (j & 3) * 2;                  // chroma plane, first chroma line of this lane
        uint8_t *tile8 = (uint8_t *)L.tile;
        const int piw = pic0 + pi;                                // picture inside the workgroup
        const int pic = blockIdx.x * pics_per_wg + piw;
        const PicDev *pd = pics + min(pic, n_pics - 1);
        const bool pic_ok = pi < PW && piw < pics_per_wg && pic < n_pics && pd->deblock;
        uint8_t *F = pd->dst;                                     // strip frame layout (device_common.h)
        const EdgeInfo *pinfo = info + (size_t)min(pic, n_pics - 1) * g.n_mb;
        const int alpha_off = pd->alpha_off, beta_off = pd->beta_off;
        uint32_t last_qp = 0xffffffffu, last_avg = 0xffffffffu;   // QPs the octet's class parameters in LDS were expanded for
        const int R0 = band << rb_log2;
        const int nrows = min(RB, g.mb_h - R0), last = nrows - 1;
        const int row = R0 + gr;
        const bool have_row = pic_ok && gr < nrows;
        const bool below_in_band = gr < last;                  // the row below belongs to the next octet
        const bool top_exists = row > 0;
        const bool from_above = EXPD_BANDSYNC && have_row && gr == 0 && band > 0;   // the rows above come from the band above, through memory
        const int rowc = min(row, g.mb_h - 1);
        // Strip layout: macroblock x of this row owns the 256 luma bytes at x*ystrip + row*256 and the 128 chroma bytes at
        // coff + x*cstrip + row*128 (rows of 8 bytes U, 8 bytes V).  This lane's two luma rows are the 32 bytes at +32j, its
        // two chroma rows 8 bytes each: an octet reads and writes whole cache lines.
        const uint32_t sY = g.ystrip, sC = g.cstrip;
        uint8_t *ownY = F + (size_t)rowc * MB_LUMA_BYTES + j * 32, *ownC = F + g.coff + (size_t)rowc * MB_CHROMA_BYTES + cr * 16 + cp * 8;
        // the rows above (valid if top_exists): lanes 0..3 luma rows 12..15 of the macroblock above, lanes 4..7 its chroma
        // rows 6,7 of plane (j>>1)&1
        const uint32_t sT = j < 4 ? sY : sC;
        uint8_t *topP = j < 4 ? F + (ptrdiff_t)(rowc - 1) * MB_LUMA_BYTES + 192 + j * 16
                              : F + g.coff + (ptrdiff_t)(rowc - 1) * MB_CHROMA_BYTES + (6 + (j & 1)) * 16 + ((j >> 1) & 1) * 8;
        // Addresses inside the iteration loop: macroblock x = t - 2 gr of the row.  The lane-constant part (- 2 gr strips) is folded
        // into the bases here, in 64-bit arithmetic, so that the loop adds only t x strip - a scalar - to one pointer per array
        // (as (uint32) x * strip the compiler cannot fold it and keeps one 64-bit constant per array alive: registers it does not have).
        const ptrdiff_t lag = -(ptrdiff_t)(DB_LAG * gr);
        uint8_t *ownY0 = ownY + lag * (ptrdiff_t)sY, *ownC0 = ownC + lag * (ptrdiff_t)sC, *topP0 = topP + lag * (ptrdiff_t)sT;
        const EdgeInfo *pinfo0 = pinfo + (ptrdiff_t)row * g.mb_w + lag;
        int *my_progress = &progress[piw & (MAX_PICS_PER_WG - 1)][band];
        const bool publisher = have_row && gr == last && j == 0;
        OctLds &Lnext = lds[wave][min(o + 1, 7)];

        uint32_t ya0 = 0, yb0 = 0, ca0 = 0, cb0 = 0;                  // columns -4..-1: the previous macroblock's last four
        uint4 fYa = make_uint4(0, 0, 0, 0), fYb = fYa, fC = fYa, fT = fYa, fE = fYa;   // in flight for the next iteration
        const int n_iter = g.mb_w + 1 + DB_LAG * last;
        const int *wait_on = from_above ? my_progress - 1 : my_progress;
        // tile addresses of this lane's rows
        uint32_t *tYa = L.tile + (2 * j) * 4, *tYb = tYa + 4, *tCa = L.tile + 64 + cp * 16 + cr * 2, *tCb = tCa + 2;

        // loads for iteration t
        auto prefetch = [&](int t) {
            const int x = t - DB_LAG * gr;
            const bool actn = have_row && x >= 0 && x < g.mb_w;
            if (ok) {
                // macroblock x needs the band above to have stored x completely
                const bool need = from_above && actn;
                const int want = x + 1;
                int spins = 0;
                while (__ballot(need && __hip_atomic_load(wait_on, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want)) {
                    __builtin_amdgcn_s_sleep(DEBLOCK_WAIT_SLEEP);
                    if (++spins > SPIN_LIMIT) { if (lane_u == 0) atomicOr(status, 1); ok = false; break; }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
            if (actn) {
                if (j == 0) fE = gload4(pinfo0 + t);
                const uint8_t *tp = topP0 + (ptrdiff_t)t * (ptrdiff_t)sT;
                if (from_above) {
                    if (j < 4) fT = gload4(tp);
                    else { uint2 v2 = gload2(tp); fT.x = v2.x; fT.y = v2.y; }
                }
                const uint8_t *yp = ownY0 + (ptrdiff_t)t * (ptrdiff_t)sY, *cp2 = ownC0 + (ptrdiff_t)t * (ptrdiff_t)sC;
                if (EXPD_NO_SAMPLE_LOADS) {
                    const uint32_t a = (uint32_t)(uintptr_t)yp, b2 = (uint32_t)(uintptr_t)cp2;
                    fYa = make_uint4(a, a + 1, a + 2, a + 3); fYb = fYa; fC = make_uint4(b2, b2 + 1, b2 + 2, b2 + 3);
                    asm volatile("" : "+v"(fYa.x), "+v"(fYb.y), "+v"(fC.z));
                } else {
                fYa = gload4(yp); fYb = gload4(yp + 16);
                { const uint2 ca2 = gload2(cp2), cb2 = gload2(cp2 + 16); fC = make_uint4(ca2.x, ca2.y, cb2.x, cb2.y); }
                }
            }
        };

#ifdef EXPD_STAMPS
        const bool stamp_me = blockIdx.x == 100 && wave == EXPD_STAMPS && unit == wave && lane == 0;
#endif
        prefetch(0);
        for (int t = 0; t < n_iter; t++) {
            DB_STAMP(0);
            const int x = t - DB_LAG * gr;
            const bool act = have_row && x >= 0 && x < g.mb_w;         // filter macroblock x
            const bool flush = have_row && x >= 1 && x <= g.mb_w;       // store macroblock x-1
            uint32_t *ring = L.ring[x & 3];
            // ---- land what was prefetched for this iteration: pixels stay in registers for the vertical pass ----
            uint32_t ya[5] = { ya0, fYa.x, fYa.y, fYa.z, fYa.w }, yb[5] = { yb0, fYb.x, fYb.y, fYb.z, fYb.w };
            uint32_t ca[3] = { ca0, fC.x, fC.y }, cb[3] = { cb0, fC.z, fC.w };
            if (act) {
                if (from_above) {
                    if (j < 4) *(uint4 *)(ring + j * 4) = fT;
                    else *(uint2 *)(ring + 16 + (j - 4) * 2) = make_uint2(fT.x, fT.y);
                }
                if (j == 0) *(uint4 *)L.edge = fE;
            }
            // Everything this wave has issued is complete here (the prefetch was issued before the previous
            // iteration's horizontal pass, its stores before that): macroblocks 0 .. x-2 of this row are in memory.
            // Publish with RELEASE semantics at workgroup scope: the band below reads these macroblocks' pixels through global
            // memory on the same CU.  (The explicit wait drains the WHOLE wave's stores - the publisher lane speaks for all
            // eight octets of its wave, and a release fence only orders the publishing lane's own view.)
            if (EXPD_VMCNT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            DB_STAMP(1);
            if (publisher) __hip_atomic_store(my_progress, min(max(x - 1, 0), g.mb_w), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            wave_lds_fence();
            // the rows above macroblock x-1 were finished by its horizontal pass in the previous iteration
            if (flush && top_exists) {
                const uint32_t *pr = L.ring[(x - 1) & 3];
                uint8_t *tp = topP0 + (ptrdiff_t)(t - 1) * (ptrdiff_t)sT;
                if (j < 4) gstore4(tp, *(const uint4 *)(pr + j * 4));
                else gstore2(tp, *(const uint2 *)(pr + 16 + (j - 4) * 2));
            }
            EdgeRegs E;
            {
                const uint4 v = act ? *(const uint4 *)L.edge : make_uint4(0, 0, 0, 0);
                E.e[0] = v.x; E.e[1] = v.y; E.lds = L.edge;
                // the class parameters of the octet's LDS copy follow the macroblock's QPs: expanded again only when those change
                const bool changed = act && (v.z != last_qp || v.w != last_avg);
                if (__ballot(changed)) {
                    if (act) {
                        if (j < 6) { const EdgeClass ec = edge_expand(tables, v.z, v.w, j, alpha_off, beta_off); *(uint4 *)(L.edge + 4 + 8 * j) = ec.ab; L.edge[8 + 8 * j] = ec.tc; }
                        last_qp = v.z; last_avg = v.w;
                    }
                    wave_lds_fence();
                }
            }
            const bool any_edges = __ballot((E.e[0] | E.e[1]) != 0) != 0;

            if (any_edges) {
                // ---------- vertical edges, in registers ----------
#pragma unroll
                for (int ed = 0; ed < EXPD_LUMA_EDGES; ed++) {
                    const int b = E.code(0, ed, seg2), k = edge_class(0, ed);
                    if (__ballot(b != 0) == 0) continue;
                    pk16 p2 = pair_byte<1>(ya[ed], yb[ed]), p1 = pair_byte<2>(ya[ed], yb[ed]), p0 = pair_byte<3>(ya[ed], yb[ed]);
                    pk16 q0 = pair_byte<0>(ya[ed+1], yb[ed+1]), q1 = pair_byte<1>(ya[ed+1], yb[ed+1]), q2 = pair_byte<2>(ya[ed+1], yb[ed+1]);
                    const EdgeParams ep(E, k);
                    const pk16 A = as_pk(ep.alpha2()), B = as_pk(ep.beta2()), nA = as_pk(ep.nalpha2()), nB = as_pk(ep.nbeta2());
                    pk16 e;
                    const pk16 f = pk_edge_flag(p1, p0, q0, q1, A, nA, B, nB, e);
                    const pk16 ap = pk_sign(pk_within(p2 - p0, B, nB)), aq = pk_sign(pk_within(q2 - q0, B, nB));
                    const pk16 en = as_pk(mask_bs123(b, ed));
                    const pk16 op2 = p2, op1 = p1, op0 = p0, oq0 = q0, oq1 = q1, oq2 = q2;
                    pk_luma_normal(p2, p1, p0, q0, q1, q2, e, f & en, ap, aq, as_pk(ep.tc2(b)));
                    // bS 4 exists on macroblock edges only (k_deblock_bs), and the strong filter changes nothing where the
                    // sample flag is off
                    if (EXPD_STRONG && ed == 0 && __ballot((as_u(f) & mask_bs4(b)) != 0)) {
                        const pk16 str = as_pk(mask_bs4(b));
                        pk16 sp2 = op2, sp1 = op1, sp0 = op0, sq0 = oq0, sq1 = oq1, sq2 = oq2;
                        pk_luma_strong(pair_byte<0>(ya[ed], yb[ed]), sp2, sp1, sp0, sq0, sq1, sq2, pair_byte<3>(ya[ed+1], yb[ed+1]), f & str, ap, aq, A);
                        p1 = pk_sel(str, sp1, p1); p0 = pk_sel(str, sp0, p0); q0 = pk_sel(str, sq0, q0); q1 = pk_sel(str, sq1, q1);
                        ya[ed] = perm(as_u(sp2), ya[ed], 0x03020400u);   yb[ed] = perm(as_u(sp2), yb[ed], 0x03020600u);
                        ya[ed+1] = perm(as_u(sq2), ya[ed+1], 0x03040100u); yb[ed+1] = perm(as_u(sq2), yb[ed+1], 0x03060100u);
                    }
                    // p0 / q0 clipped into byte pairs (row 2j in byte 0, row 2j+1 in byte 1), merged with p1 / q1: bytes {p1 a, p0 a, p1 b, p0 b}
                    const uint32_t tp = perm(pk_clip_bytes(p0), as_u(p1), 0x05020400u), tq = perm(as_u(q1), pk_clip_bytes(q0), 0x06010400u);
                    ya[ed] = perm(tp, ya[ed], 0x05040100u);     yb[ed] = perm(tp, yb[ed], 0x07060100u);
                    ya[ed+1] = perm(tq, ya[ed+1], 0x03020504u); yb[ed+1] = perm(tq, yb[ed+1], 0x03020706u);
                }
#pragma unroll
                for (int ed = 0; ed < EXPD_CHROMA_EDGES; ed += 2) {
                    const int b = E.code(0, ed, cseg2), k = edge_class(0, ed) + 3, c = ed >> 1;
                    if (__ballot(b != 0) == 0) continue;
                    pk16 p1 = pair_byte<2>(ca[c], cb[c]), p0 = pair_byte<3>(ca[c], cb[c]);
                    pk16 q0 = pair_byte<0>(ca[c+1], cb[c+1]), q1 = pair_byte<1>(ca[c+1], cb[c+1]);
                    const EdgeParams ep(E, k);
                    const pk16 A = as_pk(ep.alpha2()), B = as_pk(ep.beta2());
                    pk16 e;
                    const pk16 f = pk_edge_flag(p1, p0, q0, q1, A, as_pk(ep.nalpha2()), B, as_pk(ep.nbeta2()), e);
                    pk_chroma(p1, p0, q0, q1, e, f, as_pk(mask_bs123(b, ed)), as_pk(ed == 0 ? mask_bs4(b) : 0u), ed == 0 && __ballot(b == 3) != 0, as_pk(ep.tc2(b)));
                    const uint32_t P0 = pk_clip_bytes(p0), Q0 = pk_clip_bytes(q0);      // (row a in byte 0, row b in byte 1)
                    ca[c] = perm(P0, ca[c], 0x04020100u);     cb[c] = perm(P0, cb[c], 0x05020100u);
                    ca[c+1] = perm(Q0, ca[c+1], 0x03020104u); cb[c+1] = perm(Q0, cb[c+1], 0x03020105u);
                }
            }
            DB_STAMP(2);
            // ---- macroblock x-1 is final now: columns 0..11 still sit in the tile, its last four columns are ya[0]/yb[0].
            // Store it as whole rows; rows 12..15 go to the octet below instead, which stores them as its "rows above".
            if (flush) {
                uint4 sa = *(const uint4 *)tYa, sb = *(const uint4 *)tYb;
                uint2 ta = *(const uint2 *)tCa, tb = *(const uint2 *)tCb;
                sa.w = ya[0]; sb.w = yb[0]; ta.y = ca[0]; tb.y = cb[0];
                uint32_t *nr = Lnext.ring[(x - 1) & 3];
                uint8_t *yp = ownY0 + (ptrdiff_t)(t - 1) * (ptrdiff_t)sY, *cp2 = ownC0 + (ptrdiff_t)(t - 1) * (ptrdiff_t)sC;
                if (below_in_band && j >= 6) { *(uint4 *)(nr + (2 * j - 12) * 4) = sa; *(uint4 *)(nr + (2 * j - 11) * 4) = sb; }
                else { gstore4(yp, sa); gstore4(yp + 16, sb); }
                if (below_in_band && (j & 3) == 3) { *(uint2 *)(nr + 16 + cp * 4) = ta; *(uint2 *)(nr + 16 + cp * 4 + 2) = tb; }
                else { gstore2(cp2, ta); gstore2(cp2 + 16, tb); }
            }
            wave_lds_fence();
            // ---------- rows of macroblock x -> tile ----------
            if (act) {
                *(uint4 *)tYa = make_uint4(ya[1], ya[2], ya[3], ya[4]); *(uint4 *)tYb = make_uint4(yb[1], yb[2], yb[3], yb[4]);
                *(uint2 *)tCa = make_uint2(ca[1], ca[2]);               *(uint2 *)tCb = make_uint2(cb[1], cb[2]);
            }
            wave_lds_fence();
            DB_STAMP(3);
            if (t + 1 < n_iter) prefetch(t + 1);                      // the next iteration's loads travel during the horizontal pass
            // ---------- horizontal edges: column pairs out of the tile, filtered, back into the tile ----------
            DB_STAMP(4);
            const bool h_edges = EXPD_HPASS && __ballot(E.e[1] != 0) != 0;
            if (h_edges) {
                if (act) {
                    // luma: columns 2j, 2j+1
                    const uint8_t *top = (const uint8_t *)ring + 2 * j, *col = tile8 + 2 * j;
                    pk16 c[20];
#pragma unroll
                    for (int r = 0; r < 4; r++) { uint32_t v = *(const uint16_t *)(top + r * 16); c[r] = as_pk(perm(v, v, 0x0c010c00u)); }
#pragma unroll
                    for (int r = 0; r < 16; r++) { uint32_t v = *(const uint16_t *)(col + r * 16); c[4 + r] = as_pk(perm(v, v, 0x0c010c00u)); }
#pragma unroll
                    for (int ed = 0; ed < EXPD_LUMA_EDGES; ed++) {
                        const int b = E.code(1, ed, seg2), k = edge_class(1, ed);
                        if (__ballot(b != 0) == 0) continue;
                        pk16 &p3 = c[4*ed], &p2 = c[4*ed+1], &p1 = c[4*ed+2], &p0 = c[4*ed+3], &q0 = c[4*ed+4], &q1 = c[4*ed+5], &q2 = c[4*ed+6], &q3 = c[4*ed+7];
                        const EdgeParams ep(E, k);
                        const pk16 A = as_pk(ep.alpha2()), B = as_pk(ep.beta2()), nA = as_pk(ep.nalpha2()), nB = as_pk(ep.nbeta2());
                        pk16 e;
                        const pk16 f = pk_edge_flag(p1, p0, q0, q1, A, nA, B, nB, e);
                        const pk16 ap = pk_sign(pk_within(p2 - p0, B, nB)), aq = pk_sign(pk_within(q2 - q0, B, nB));
                        const pk16 en = as_pk(mask_bs123(b, ed));
                        pk16 sp2 = p2, sp1 = p1, sp0 = p0, sq0 = q0, sq1 = q1, sq2 = q2;
                        pk_luma_normal(p2, p1, p0, q0, q1, q2, e, f & en, ap, aq, as_pk(ep.tc2(b)));
                        if (EXPD_STRONG && ed == 0 && __ballot((as_u(f) & mask_bs4(b)) != 0)) {
                            const pk16 str = as_pk(mask_bs4(b));
                            pk_luma_strong(p3, sp2, sp1, sp0, sq0, sq1, sq2, q3, f & str, ap, aq, A);
                            p2 = pk_sel(str, sp2, p2); p1 = pk_sel(str, sp1, p1); p0 = pk_sel(str, sp0, p0);
                            q0 = pk_sel(str, sq0, q0); q1 = pk_sel(str, sq1, q1); q2 = pk_sel(str, sq2, q2);
                        }
                    }
                    // rows -3 .. 14 back (row -4 cannot change).  The rows either side of an edge (p0 / q0: -1 | 0, 3 | 4, 7 | 8, 11 | 12) may
                    // hold p0 + delta / q0 - delta not yet clipped: clipped into a byte pair here, one 16-bit store
                    {
                        uint8_t *topw = (uint8_t *)ring + 2 * j, *colw = tile8 + 2 * j;
                        lds_put_pair<1 * 16>(topw, c[1]); lds_put_pair<2 * 16>(topw, c[2]); lds_put_clipped<3 * 16>(topw, c[3]);
                        lds_put_clipped<0 * 16>(colw, c[4]);  lds_put_pair<1 * 16>(colw, c[5]);   lds_put_pair<2 * 16>(colw, c[6]);   lds_put_clipped<3 * 16>(colw, c[7]);
                        lds_put_clipped<4 * 16>(colw, c[8]);  lds_put_pair<5 * 16>(colw, c[9]);   lds_put_pair<6 * 16>(colw, c[10]);  lds_put_clipped<7 * 16>(colw, c[11]);
                        lds_put_clipped<8 * 16>(colw, c[12]); lds_put_pair<9 * 16>(colw, c[13]);  lds_put_pair<10 * 16>(colw, c[14]); lds_put_clipped<11 * 16>(colw, c[15]);
                        lds_put_clipped<12 * 16>(colw, c[16]); lds_put_pair<13 * 16>(colw, c[17]); lds_put_pair<14 * 16>(colw, c[18]);
                    }
                    // chroma: columns cr, cr+1 of plane cp; only rows -1, 0, 3, 4 can change
                    const uint8_t *ctop = (const uint8_t *)ring + 64 + cp * 16 + cr;
                    uint8_t *ccol = tile8 + 256 + cp * 64 + cr;
                    pk16 d[10];
#pragma unroll
                    for (int r = 0; r < 2; r++) { uint32_t v = *(const uint16_t *)(ctop + r * 8); d[r] = as_pk(perm(v, v, 0x0c010c00u)); }
#pragma unroll
                    for (int r = 0; r < 8; r++) { uint32_t v = *(const uint16_t *)(ccol + r * 8); d[2 + r] = as_pk(perm(v, v, 0x0c010c00u)); }
#pragma unroll
                    for (int ed = 0; ed < EXPD_CHROMA_EDGES; ed += 2) {
                        const int b = E.code(1, ed, cseg2), k = edge_class(1, ed) + 3;
                        if (__ballot(b != 0) == 0) continue;
                        const EdgeParams ep(E, k);
                        const pk16 A = as_pk(ep.alpha2()), B = as_pk(ep.beta2());
                        pk16 e;
                        const pk16 f = pk_edge_flag(d[2*ed], d[2*ed+1], d[2*ed+2], d[2*ed+3], A, as_pk(ep.nalpha2()), B, as_pk(ep.nbeta2()), e);
                        pk_chroma(d[2*ed], d[2*ed+1], d[2*ed+2], d[2*ed+3], e, f, as_pk(mask_bs123(b, ed)), as_pk(ed == 0 ? mask_bs4(b) : 0u), ed == 0 && __ballot(b == 3) != 0, as_pk(ep.tc2(b)));
                    }
                    // (all four are p0 / q0 rows: clipped here)
                    lds_put_clipped<8>((uint8_t *)ring + 64 + cp * 16 + cr, d[1]);
                    lds_put_clipped<0>(ccol, d[2]);
                    lds_put_clipped<3 * 8>(ccol, d[5]);
                    lds_put_clipped<4 * 8>(ccol, d[6]);
                }
                wave_lds_fence();
            }
            DB_STAMP(5);
            // the last four columns of this macroblock are the next one's columns -4..-1
            if (act) { ya0 = tYa[3]; yb0 = tYb[3]; ca0 = tCa[1]; cb0 = tCb[1]; }
            wave_lds_fence();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (publisher) __hip_atomic_store(my_progress, g.mb_w, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        wave_lds_fence();
    }
}
#endif   // P264HIP_K_DEBLOCK_DECL_ONLY
