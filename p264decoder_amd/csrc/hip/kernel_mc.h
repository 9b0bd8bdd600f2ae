// kernel_mc.h - K1: inter prediction + residual for every non-intra macroblock of a batch.
//
// Replaces p264_mb_mc / p264_mb_mc_0xywh (core/macroblock.c:506-524,633-676), mc_luma / pixel_avg / mc_copy
// (core/mc.c:58-74,160-171,237-266), the half-pel plane generator p264_frame_filter (core/mc.c:172-235,409-451 -
// computed on the fly here, never stored), motion_compensation_chroma (core/mc.c:303-334), p264_macroblock_decode_skip
// (decoder/macroblock.c:895-934), the border expansion (core/frame.c:183-222 - clamped coordinates instead, SURVEY A-Q9)
// and the inter half of p264_macroblock_decode (decoder/macroblock.c:832-890: unscan, dequant_4x4, add4x4_idct, chroma DC).
//
// The first version of this stage (one wavefront per macroblock, everything about the macroblock in SGPRs) was bound by
// scalar and vector instruction issue.  This version is built on three ideas:
//
//  1. ONE LANE = ONE 4x4 BLOCK.  The lane interpolates its 16 samples from a 9x9 (or smaller) window held in registers,
//     runs the whole inverse transform of its block in registers and stores two rows of 8 samples (after swapping halves
//     with its neighbour lane).  Every partition shape down to 4x4 is the same code: a lane only ever looks at its own
//     vector.  All per-macroblock header work is vector work shared by 64 blocks.
//  2. WORK LISTS SORTED BY WHAT THE CODE HAS TO DO.  The quarter-pel phase decides the arithmetic (copy / horizontal /
//     vertical / both / centre ...), so k_mc_sort (one workgroup per picture, a counting sort in LDS) hands the work over
//     grouped by {phase class, window inside the picture or not, residual present or not} (and by band of macroblock
//     rows, for cache locality).  A wavefront takes one chunk of ONE key: the class is a scalar, the switch on it is free,
//     nobody executes code it does not need - a P_SKIP macroblock at an integer position costs a few loads and stores,
//     windows inside the picture need no clamping, wavefronts without coded blocks skip the transform.  Sorting on the
//     device keeps the host parser and the CPU->GPU seam unchanged.
//  3. WINDOWS SHARED THROUGH LDS, AT THE LARGEST GRANULARITY THAT HAS ONE VECTOR.  A lane fetching its own window from
//     memory costs one L1 tag look-up per lane and dword (measured: ~50 look-ups per load instruction, 27 loads per
//     wavefront - the L1's look-up rate was the bound).  So the lanes of a work item stage the item's window in LDS with
//     16-byte loads of consecutive rows of a strip (64 contiguous bytes per four lanes) and read their own 9x9 windows out
//     of that image: one address register, rows and columns as immediates.  Work items are whole macroblocks where the
//     macroblock has one vector (16 lanes, a 21-row window: P_SKIP, 16x16 - most of a typical stream) and 8x8 quadrants
//     otherwise (4 lanes, a 13-row window); quadrants whose 4x4 blocks differ (sub-8x8 partitions) fetch per lane with
//     clamped coordinates.
//
// Chroma (4x4 per quadrant and plane, bilinear) has no phase classes; it has its own roles and lists (macroblock items
// of 8 lanes; quadrant items of 2 lanes, the four of a macroblock consecutive), keys {inside / clamped, residual or not}.
//
// One launch (k_mc) runs the four kinds of work - luma / chroma x macroblock / quadrant items - as roles of the workgroups a
// picture gets (see k_mc below); B pictures take a second launch (k_mc_second) for the list-1 half of the blocks that
// predict from both lists (mc_classify).
//
// Arithmetic to preserve: core/mc.c:172-266 (half-pel planes, quarter-pel averages), :303-334 (chroma),
// core/quant.c:66-99,138-159, core/dct.c:55-68,205-247 with their int16 stores (A-Q8).
#pragma once
#include "device_common.h"
#include <type_traits>
// Timing experiments (scratch/r4_mcexp.sh, r4_mcparts.sh): pieces of the kernels compiled out to time the rest.  Results are
// wrong unless all defaults hold, so the switches only exist in a build that says what it is: -DP264AMD_TIMING_BUILD, in
// which p264hip_create refuses to run without P264AMD_TIMING_BUILD_OK=1 and p264hip_build_info() reports the flag.
#if !defined(P264AMD_TIMING_BUILD) && (defined(EXPM_LUMA_COPY) || defined(EXPM_NO_STORE) || defined(EXPM_NO_WINDOW) || defined(EXPM_ONLY) || defined(EXPM_RESID) || defined(EXPM_FORCE_KEY) || defined(EXPM_HALF_LEVELS))
#error "EXPM_* switches produce wrong pictures: they need -DP264AMD_TIMING_BUILD"
#endif
#ifndef EXPM_LUMA_COPY
#define EXPM_LUMA_COPY 0
#endif
#ifndef EXPM_NO_STORE
#define EXPM_NO_STORE 0            // 1: the sample stores are left out (their operands are still computed)
#endif
#ifndef EXPM_NO_WINDOW
#define EXPM_NO_WINDOW 0           // 1: the reference-window loads are left out (the staging stores and everything behind them stay)
#endif
#ifndef EXPM_ONLY
#define EXPM_ONLY 0                // 1: only the luma roles work, 2: only the chroma roles
#endif
#ifndef EXPM_RESID
#define EXPM_RESID 1
#endif
#ifndef EXPM_HALF_LEVELS
#define EXPM_HALF_LEVELS 0         // 1: a coded block's levels are read as ONE 16-byte piece at half the stride (round 6: what 8-bit levels could save at most)
#endif
// a coded block's sixteen levels (two 16-byte pieces)
#define MC_LOAD_LEVELS(cfp, la_, lb_) do { if (EXPM_HALF_LEVELS) { la_ = gload4((const int16_t *)((const uint8_t *)(cfp) - ((const uint8_t *)(cfp) - (const uint8_t *)pd->coefs) / 2)); lb_ = la_; } \
                                            else { la_ = gload4(cfp); lb_ = gload4((cfp) + 8); } } while (0)
// EXPM_FORCE_KEY=k: every chunk is taken for key k (scratch/mc_count.sh: static instruction counts per role and class)
#ifdef EXPM_FORCE_KEY
#define MC_CHUNK_KEY(v) ((void)(v), (int)(EXPM_FORCE_KEY))
#else
#define MC_CHUNK_KEY(v) (v)
#endif

// ------------------------------------------------------------------------------------------
// work lists
// ------------------------------------------------------------------------------------------
enum { PC_COPY = 0, PC_H = 1, PC_V = 2, PC_DIAG = 3, PC_C = 4, PC_CH = 5, PC_CV = 6, PC_GEN = 7 };
// PC_GEN | MCY_CLAMP: the 4x4 blocks of the quadrant have different vectors (sub-8x8 partitions), every lane fetches its own
// window.  PC_GEN without MCY_CLAMP: the macroblock belongs to a B picture and predicts from list 1 somewhere: every lane
// looks up which lists its quadrant uses, predicts from each of them like the class above and combines the two
// (p264_mb_mc_1xywh / _01xywh, core/macroblock.c:525-583).
#define MCY_CLAMP   8               // luma key bits: phase class | window not inside the picture | item has coded luma blocks
#define MCY_RESID   16
#define MCY_KEYS    32
#define MCC_CLAMP   1               // chroma key bits: window not inside the picture (quadrant items: or vectors differing inside) | residual
#define MCC_RESID   2
#define MCC_BI      4               // | macroblock of a B picture that predicts from list 1 somewhere (quadrant items only)
#define MCC_KEYS    8
enum { ML_YM = 0, ML_YQ = 1, ML_CM = 2, ML_CQ = 3, ML_LISTS = 4 };      // luma macroblocks, luma quadrants, chroma macroblocks, chroma quadrants
#define MC_MAX_BANDS 32
#define MC_SORT_THREADS 1024
__host__ __device__ static inline int mc_chunk_items(int l) { return l == ML_YM ? 4 : l == ML_YQ ? 16 : l == ML_CM ? 8 : 32; }   // items per wavefront
__host__ __device__ static inline int mc_list_keys(int l) { return l < ML_CM ? MCY_KEYS : MCC_KEYS; }

// Per-picture scratch written by k_mc_sort (32-bit words): [l] chunks in use of list l, then per list one class byte
// per chunk, then the lists.  A first-pass entry is 8 bytes: x = reference index << 28 | flags | macroblock row << 13 |
// column << 2 | quadrant (low 28 bits all ones = padding; no division in the consumers), y = the item's vector (packed) -
// all a wavefront needs to start fetching its windows; chunks with residual (a third of them in the bench's pictures) fetch
// the macroblock's record for QP, coded-block mask and place in the coefficient stream, one chunk ahead (round 4 carried the
// three in every entry: 16 bytes, written and read for all chunks, and the sort read the record a second time to fill them
// in).  Second-pass entries (B pictures) stay 16 bytes: z = coded-block mask | QP << 26, w = place in the coefficient
// stream | weight of the pair of references << 23.  Same layout for every picture of a batch.
#define MC_ITEM_MASK 0x0fffffffu
__host__ __device__ static inline uint32_t mc_entry_words(int pass) { return pass ? 4u : 2u; }
struct McLayout {
    uint32_t band_log2, n_bands;
    uint32_t max_chunks[2 * ML_LISTS], off_cls[2 * ML_LISTS], off_list[2 * ML_LISTS], words;     // [pass * ML_LISTS + list]
};
static inline McLayout mc_layout(int mb_w, int mb_h, int band_log2)
{
    McLayout L;
    L.band_log2 = (uint32_t)band_log2;
    L.n_bands = (uint32_t)((mb_h + (1 << band_log2) - 1) >> band_log2);
    const uint32_t n_mb = (uint32_t)(mb_w * mb_h);
    uint32_t at = 16;
    for (int l = 0; l < 2 * ML_LISTS; l++) {
        const int t = l % ML_LISTS;
        const uint32_t items = (t == ML_YM || t == ML_CM) ? n_mb : n_mb * 4u, per = (uint32_t)mc_chunk_items(t);
        L.max_chunks[l] = (items + L.n_bands * (uint32_t)mc_list_keys(t) * (per - 1)) / per + 1;
        L.off_cls[l] = at; at += (L.max_chunks[l] + 3) / 4;
    }
    at = (at + 15) & ~15u;
    for (int l = 0; l < 2 * ML_LISTS; l++) { L.off_list[l] = at; at += L.max_chunks[l] * (uint32_t)mc_chunk_items(l % ML_LISTS) * mc_entry_words(l / ML_LISTS); }
    L.words = (at + 63) & ~63u;
    return L;
}

__device__ __forceinline__ int mv_x(int packed) { return (int)(int16_t)(packed & 0xffff); }
__device__ __forceinline__ int mv_y(int packed) { return packed >> 16; }

// phase class of a quarter-pel vector (which of the reference's planes core/mc.c:244-257 combines)
__device__ __forceinline__ int phase_class(int fx, int fy)
{
    if ((fx | fy) == 0) return PC_COPY;
    if (fy == 0) return PC_H;
    if (fx == 0) return PC_V;
    if (fx & fy & 1) return PC_DIAG;
    if (fx == 2) return fy == 2 ? PC_C : PC_CH;
    return PC_CV;
}

__device__ __forceinline__ void split_mb(int mbi, const Geom &g, uint32_t inv_mbw, int &mbx, int &mby)
{
    mby = (int)__umulhi((unsigned)mbi, inv_mbw);
    if (mbi - mby * g.mb_w >= g.mb_w) mby++;
    mbx = mbi - mby * g.mb_w;
}

// What one inter macroblock contributes to one pass: either one macroblock item per plane kind (one vector, one reference)
// or its four quadrants.  Packed so that a thread can keep the classification of several macroblocks in registers between
// the counting and the scattering pass: key[q] = luma key | chroma key << 16, vec[q] = the entry's vector,
// info = inter | whole << 1 | reference indices (4 bits each) << 8 | list per quadrant << 24 | Z per quadrant << 28.
//
// B pictures take TWO passes (two list sets, two launches of the consumer): a block that predicts from both lists gets its
// list-0 prediction in the first pass (entry with an empty coded-block mask: Z) and the list-1 prediction in the second,
// which reads the first one back, combines the two (core/mc.c:76-155) and adds the residual.  Blocks with one list are
// finished in the first pass, whichever list it is.  In the second pass Z on a quadrant means "not mine": no luma entry,
// and a chroma entry that only carries the quadrant's samples through (the four chroma entries of a macroblock stay
// together: their lanes store whole rows).  Macroblocks of a B picture whose vectors differ inside a quadrant go to the
// generic two-list class, in the first pass, as a whole.
struct McMb { uint32_t info, key[4], vec[4]; };
#define MCMB_INTER 1u
#define MCMB_WHOLE 2u
#define MCE_LIST1  (1u << 23)       // entry flag: the reference index counts in list 1
#define MCE_KEEP   (1u << 24)       // entry flag (second pass, chroma quadrant entries): pass the samples through
#define MCE_Z      (1u << 25)       // entry flag (first pass): no residual here - the second pass adds it
#define MCE_W_SHIFT 23              // second pass: w = place in the coefficient stream | weight << 23
__device__ __forceinline__ uint32_t mcmb_entry(const McMb &k, int mbx, int mby, int q, int pass)
{
    const uint32_t l1 = (k.info >> (24 + q)) & 1u, z = (k.info >> (28 + q)) & 1u;
    return ((k.info >> (8 + 4 * q)) & 15u) << 28 | (l1 ? MCE_LIST1 : 0u) | (z ? (pass ? MCE_KEEP : MCE_Z) : 0u) | (uint32_t)mby << 13 | (uint32_t)mbx << 2 | (uint32_t)q;
}
__device__ __forceinline__ bool quad_uniform(const uint4 &m0, const uint4 &m1, const uint4 &m2, const uint4 &m3, int q, uint32_t &v)
{
    // vectors of the quadrant's four 4x4 blocks (raster inside the macroblock: b0, b0+1, b0+4, b0+5)
    const uint4 top = q < 2 ? m0 : m2, bot = q < 2 ? m1 : m3;
    const uint32_t va = (q & 1) ? top.z : top.x, vb = (q & 1) ? top.w : top.y, vc = (q & 1) ? bot.z : bot.x, vd = (q & 1) ? bot.w : bot.y;
    v = va;
    return va == vb && va == vc && va == vd;
}
// what the classification reads of a macroblock (once, whatever the number of passes)
struct McIn { uint4 rec, m0, m1, m2, m3, n0, n1, n2, n3; uint32_t refs, refs1; bool two_lists; };
template <bool BPIC>
__device__ __forceinline__ McIn mc_load(const PicDev *pd, int mbi)
{
    McIn in;
    in.rec = gload4(pd->mb + mbi);
    const int *mvp = pd->mv + mbi * 16;
    in.m0 = gload4(mvp); in.m1 = gload4(mvp + 4); in.m2 = gload4(mvp + 8); in.m3 = gload4(mvp + 12);
    in.refs = gload1(pd->ref_idx + mbi * 4);
    in.refs1 = 0xffffffffu; in.two_lists = false;
    in.n0 = in.n1 = in.n2 = in.n3 = make_uint4(0, 0, 0, 0);
    // B pictures: a macroblock that predicts from list 0 only is an ordinary P-type macroblock for this stage
    if (BPIC && pd->slice_type == P264_SLICE_B && !P264_MB_IS_INTRA(in.rec.x & 255)) {
        in.refs1 = gload1(pd->ref_idx_l1 + mbi * 4);
        in.two_lists = ((~in.refs1 | in.refs) & 0x80808080u) != 0;             // some list-1 index >= 0, or some list-0 index < 0
        if (in.two_lists) { const int *mvp1 = pd->mv_l1 + mbi * 16; in.n0 = gload4(mvp1); in.n1 = gload4(mvp1 + 4); in.n2 = gload4(mvp1 + 8); in.n3 = gload4(mvp1 + 12); }
    }
    return in;
}
template <bool BPIC>
__device__ __forceinline__ McMb mc_classify(const PicDev *pd, const Geom &g, const McIn &in, int mbi, uint32_t inv_mbw, int band_log2, int pass)
{
    McMb k;
    const uint4 rec = in.rec;
    uint4 m0 = in.m0, m1 = in.m1, m2 = in.m2, m3 = in.m3;
    const uint32_t refs = in.refs;
    k.info = P264_MB_IS_INTRA(rec.x & 255) ? 0u : MCMB_INTER;
#pragma unroll
    for (int q = 0; q < 4; q++) { k.key[q] = 0; k.vec[q] = 0; }
    int mby = (int)__umulhi((unsigned)mbi, inv_mbw);
    if (mbi - mby * g.mb_w >= g.mb_w) mby++;
    const int mbx = mbi - mby * g.mb_w, band = mby >> band_log2;
    const unsigned mask = rec.y, cc = mask & (0x00ff0000u | P264_COEF_CHROMA_DC);   // any chroma level present
    const int n_ref = pd->n_ref;
    int ri[4];
    bool zq[4] = { false, false, false, false };           // Z per quadrant
    const bool two_lists = in.two_lists;
    const uint32_t refs1 = in.refs1;
    if (!BPIC || !two_lists) {
        if (pass) { k.info = 0; return k; }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            ri[q] = (int)(int8_t)(refs >> (8 * q));
            if (ri[q] < 0 || ri[q] >= n_ref) ri[q] = 0;    // negative or past the list: entry 0, as the reference's flat lists
            k.info |= (uint32_t)ri[q] << (8 + 4 * q);
        }
    } else {
        const uint4 n0 = in.n0, n1 = in.n1, n2 = in.n2, n3 = in.n3;
        bool ok = true, any = false;
        uint32_t lq = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int r0 = (int)(int8_t)(refs >> (8 * q)), r1 = (int)(int8_t)(refs1 >> (8 * q));
            const bool u1 = r1 >= 0, u0 = r0 >= 0 || !u1;  // (no list at all: list 0, entry 0)
            uint32_t v0, v1;
            const bool un0 = quad_uniform(m0, m1, m2, m3, q, v0), un1 = quad_uniform(n0, n1, n2, n3, q, v1);
            ok &= (!u0 || un0) && (!u1 || un1);
            const bool bi = u0 && u1;
            bool l1; int r;
            if (!pass) { l1 = !u0; zq[q] = bi; r = l1 ? r1 : r0; any = true; }
            else       { l1 = bi; zq[q] = !bi; r = bi ? r1 : 0; any |= bi; }
            if (r < 0 || r >= (l1 ? pd->n_ref_l1 : n_ref)) r = 0;
            // (list and index: what a whole macroblock has to agree on; second pass: the list-0 index too - the weight is per pair)
            ri[q] = r | (l1 ? 16 : 0) | ((pass && bi) ? (r0 & 15) << 5 : 0);
            k.info |= (uint32_t)r << (8 + 4 * q);
            lq |= (l1 ? 1u : 0u) << q;
            // this pass's vector of the quadrant takes the place of the list-0 vectors below
            const uint32_t v = (pass && !bi) ? 0u : l1 ? v1 : v0;
            if (q == 0) { m0.x = m0.y = m1.x = m1.y = v; } else if (q == 1) { m0.z = m0.w = m1.z = m1.w = v; }
            else if (q == 2) { m2.x = m2.y = m3.x = m3.y = v; } else { m2.z = m2.w = m3.z = m3.w = v; }
        }
        if (!ok) {
            // vectors differing inside a quadrant: the generic two-list class, all four quadrants, first pass
            if (pass) { k.info = 0; return k; }
            k.info = MCMB_INTER;
            const int kc = band * MCC_KEYS + MCC_BI + (cc ? MCC_RESID : 0);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int ky = band * MCY_KEYS + (PC_GEN | (((mask >> (4 * q)) & 15) ? MCY_RESID : 0));
                k.key[q] = (uint32_t)ky | (uint32_t)kc << 16;
            }
            return k;
        }
        if (!any) { k.info = 0; return k; }
        k.info |= lq << 24 | ((uint32_t)zq[0] | (uint32_t)zq[1] << 1 | (uint32_t)zq[2] << 2 | (uint32_t)zq[3] << 3) << 28;
    }
    const uint32_t v0 = m0.x;
    const uint32_t diff = (m0.y ^ v0) | (m0.z ^ v0) | (m0.w ^ v0) | (m1.x ^ v0) | (m1.y ^ v0) | (m1.z ^ v0) | (m1.w ^ v0) | (m2.x ^ v0) | (m2.y ^ v0)
                        | (m2.z ^ v0) | (m2.w ^ v0) | (m3.x ^ v0) | (m3.y ^ v0) | (m3.z ^ v0) | (m3.w ^ v0);
    const bool whole = diff == 0 && ri[0] == ri[1] && ri[0] == ri[2] && ri[0] == ri[3] && zq[0] == zq[1] && zq[0] == zq[2] && zq[0] == zq[3];
    if (whole) {
        k.info |= MCMB_WHOLE;
        const int mvx = mv_x((int)v0), mvy = mv_y((int)v0);
        const int wx = mbx * 16 + (mvx >> 2) - 2, wy = mby * 16 + (mvy >> 2) - 2;      // 21 x 21 samples
        const bool in_y = wx >= 0 && wx + 21 <= g.w && wy >= 0 && wy + 21 <= g.h;
        const int cx = mbx * 8 + (mvx >> 3), cy = mby * 8 + (mvy >> 3);                 // 9 x 9 samples
        const bool in_c = cx >= 0 && cx + 9 <= g.cw && cy >= 0 && cy + 9 <= g.ch;
        const int ky = band * MCY_KEYS + (phase_class(mvx & 3, mvy & 3) | (in_y ? 0 : MCY_CLAMP) | (((mask & 0xffffu) && !zq[0]) ? MCY_RESID : 0));
        const int kc = band * MCC_KEYS + (in_c ? 0 : MCC_CLAMP) + ((cc && !zq[0]) ? MCC_RESID : 0);
        k.key[0] = (uint32_t)ky | (uint32_t)kc << 16;
        k.vec[0] = v0;
        return k;
    }
    bool c_inside = true;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint32_t va;
        const bool uniform = quad_uniform(m0, m1, m2, m3, q, va);
        const int X0 = mbx * 16 + (q & 1) * 8, Y0 = mby * 16 + (q >> 1) * 8;
        const int mvx = mv_x((int)va), mvy = mv_y((int)va);
        const int wx = X0 + (mvx >> 2) - 2, wy = Y0 + (mvy >> 2) - 2;                   // 13 x 13 samples
        const bool in_y = wx >= 0 && wx + 13 <= g.w && wy >= 0 && wy + 13 <= g.h;
        const int cx = X0 / 2 + (mvx >> 3), cy = Y0 / 2 + (mvy >> 3);                   // 5 x 5 samples
        const bool in_c = cx >= 0 && cx + 5 <= g.cw && cy >= 0 && cy + 5 <= g.ch;
        int pc = phase_class(mvx & 3, mvy & 3), fl = in_y ? 0 : MCY_CLAMP;
        if (!uniform) { pc = PC_GEN; fl = MCY_CLAMP; }
        if (((mask >> (4 * q)) & 15) && !zq[q]) fl |= MCY_RESID;
        const int ky = band * MCY_KEYS + (pc | fl);
        c_inside &= in_c && uniform;
        k.key[q] = (uint32_t)ky;
        k.vec[q] = va;
    }
    // the chroma entries of a split macroblock stay together (four consecutive list entries, quadrant 0..3): one key for all
    const int kc = band * MCC_KEYS + (c_inside ? 0 : MCC_CLAMP) + (cc ? MCC_RESID : 0);
#pragma unroll
    for (int q = 0; q < 4; q++) k.key[q] |= (uint32_t)kc << 16;
    return k;
}

// One workgroup per picture, one thread per macroblock: count the keys of the four lists, lay the key segments out (each
// padded to whole chunks), scatter the entries.  Up to MC_SORT_KEEP macroblocks per thread (1080p: 8) the classification
// stays in registers between the two passes: the macroblock arrays are read once.
#define MC_KEY_SLOTS (2 * MC_MAX_BANDS * MCY_KEYS + 2 * MC_MAX_BANDS * MCC_KEYS)
#define MC_SORT_KEEP 8
struct McSortCtx { const PicDev *pd; uint32_t *cnt, *pos, *out; int b_ym, b_yq, b_cm, b_cq; uint32_t l_ym, l_yq, l_cm, l_cq; int pass; };
__device__ __forceinline__ bool mcmb_luma_quad(const McMb &k, int q, int pass) { return !(pass && ((k.info >> (28 + q)) & 1u)); }
__device__ __forceinline__ void mc_count(const McSortCtx &c, const McMb &k)
{
    if (!(k.info & MCMB_INTER)) return;
    if (k.info & MCMB_WHOLE) { atomicAdd(&c.cnt[c.b_ym + (k.key[0] & 0xffffu)], 1u); atomicAdd(&c.cnt[c.b_cm + (k.key[0] >> 16)], 1u); }
    else {
#pragma unroll
        for (int q = 0; q < 4; q++) if (mcmb_luma_quad(k, q, c.pass)) atomicAdd(&c.cnt[c.b_yq + (k.key[q] & 0xffffu)], 1u);
        atomicAdd(&c.cnt[c.b_cq + (k.key[0] >> 16)], 4u);
    }
}
__device__ __forceinline__ void mc_scatter(const McSortCtx &c, const McMb &k, int mbi, const Geom &g, uint32_t inv_mbw)
{
    if (!(k.info & MCMB_INTER)) return;
    int mbx, mby;
    split_mb(mbi, g, inv_mbw, mbx, mby);
    if (!c.pass) {
        // first pass: 8-byte entries, nothing of the record in them
        if (k.info & MCMB_WHOLE) {
            const uint2 e = make_uint2(mcmb_entry(k, mbx, mby, 0, 0), k.vec[0]);
            gstore2(c.out + c.l_ym + 2u * atomicAdd(&c.pos[c.b_ym + (k.key[0] & 0xffffu)], 1u), e);
            gstore2(c.out + c.l_cm + 2u * atomicAdd(&c.pos[c.b_cm + (k.key[0] >> 16)], 1u), e);
        } else {
            const uint32_t cq = atomicAdd(&c.pos[c.b_cq + (k.key[0] >> 16)], 4u);     // (a multiple of 4: every segment starts on a chunk)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint2 e = make_uint2(mcmb_entry(k, mbx, mby, q, 0), k.vec[q]);
                gstore2(c.out + c.l_yq + 2u * atomicAdd(&c.pos[c.b_yq + (k.key[q] & 0xffffu)], 1u), e);
                gstore2(c.out + c.l_cq + 2u * (cq + (uint32_t)q), e);
            }
        }
        return;
    }
    // second pass: the record's residual fields and the weight of the pair of references (the table the parser derived,
    // core/macroblock.c:525-583; 32 = plain average) ride in the entry
    const uint4 rec = gload4(c.pd->mb + mbi);              // (second look at the record: out of the cache)
    const uint32_t ez = (rec.y & 0x03ffffffu) | ((rec.x >> 8) & 63u) << 26;
    uint32_t ew[4];
    const uint32_t refs = gload1(c.pd->ref_idx + mbi * 4);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        int w = 32;
        if (c.pd->weighted) {
            const int r0 = min(max((int)(int8_t)(refs >> (8 * q)), 0), c.pd->n_ref - 1), r1 = (int)((k.info >> (8 + 4 * q)) & 15u);
            w = (int)glob(c.pd->bipred_w)[r0 * P264HIP_MAX_REFS + r1];
        }
        ew[q] = (rec.z & ((1u << MCE_W_SHIFT) - 1u)) | (uint32_t)w << MCE_W_SHIFT;
    }
    if (k.info & MCMB_WHOLE) {
        const uint4 e = make_uint4(mcmb_entry(k, mbx, mby, 0, 1), k.vec[0], ((k.info >> 28) & 1u) ? (ez & 0xfc000000u) : ez, ew[0]);
        gstore4(c.out + c.l_ym + 4u * atomicAdd(&c.pos[c.b_ym + (k.key[0] & 0xffffu)], 1u), e);
        gstore4(c.out + c.l_cm + 4u * atomicAdd(&c.pos[c.b_cm + (k.key[0] >> 16)], 1u), e);
    } else {
        const uint32_t cq = atomicAdd(&c.pos[c.b_cq + (k.key[0] >> 16)], 4u);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint4 e = make_uint4(mcmb_entry(k, mbx, mby, q, 1), k.vec[q], ((k.info >> (28 + q)) & 1u) ? (ez & 0xfc000000u) : ez, ew[q]);
            if (mcmb_luma_quad(k, q, 1)) gstore4(c.out + c.l_yq + 4u * atomicAdd(&c.pos[c.b_yq + (k.key[q] & 0xffffu)], 1u), e);
            gstore4(c.out + c.l_cq + 4u * (cq + (uint32_t)q), e);
        }
    }
}
// One picture: P pictures one pass (the classification of up to MC_SORT_KEEP macroblocks per thread stays in registers between
// counting and scattering: the macroblock arrays are read once); B pictures both passes in one sweep (the arrays are read
// twice - keeping two classifications of eight macroblocks does not fit the register file).
template <bool BPIC>
__device__ __forceinline__ void mc_sort_picture(const PicDev *__restrict__ pics, uint32_t *__restrict__ mc_all, const Geom &g, const McLayout &ml, uint32_t inv_mbw,
                                                uint32_t *cnt, uint32_t *pos, uint8_t *__restrict__ is_intra_all)
{
    // by-product for the intra stage (k_intra's collect pass): one byte per macroblock, 1 = intra
    AS1 uint8_t *is_intra = glob(is_intra_all + (size_t)blockIdx.x * g.n_mb);
    constexpr int NP = BPIC ? 2 : 1;
    const PicDev *pd = pics + blockIdx.x;
    uint32_t *out = mc_all + (size_t)blockIdx.x * ml.words;
    const int tid = threadIdx.x;
    const int n_pass = pd->slice_type == P264_SLICE_I ? 0 : (BPIC && pd->slice_type == P264_SLICE_B) ? 2 : 1;       // wave-uniform
    if (tid < ML_LISTS * 2 && tid >= ML_LISTS * n_pass) gstore1(out + tid, 0);                            // the passes not taken: empty lists
    if (n_pass == 0) return;
    const int nky = (int)ml.n_bands * MCY_KEYS, nkc = (int)ml.n_bands * MCC_KEYS;
    const int b_ym = 0, b_yq = nky, b_cm = 2 * nky, b_cq = 2 * nky + nkc, b_end = 2 * nky + 2 * nkc;     // key slots of the four lists
    McSortCtx ctx[NP];
#pragma unroll
    for (int ps = 0; ps < NP; ps++)
        ctx[ps] = McSortCtx{ pd, cnt + ps * MC_KEY_SLOTS, pos + ps * MC_KEY_SLOTS, out, b_ym, b_yq, b_cm, b_cq, ml.off_list[ps * ML_LISTS + ML_YM], ml.off_list[ps * ML_LISTS + ML_YQ],
                             ml.off_list[ps * ML_LISTS + ML_CM], ml.off_list[ps * ML_LISTS + ML_CQ], ps };
    for (int k = tid; k < NP * MC_KEY_SLOTS; k += MC_SORT_THREADS) cnt[k] = 0;
    __syncthreads();
    const bool keep = !BPIC && g.n_mb <= MC_SORT_KEEP * MC_SORT_THREADS;
    McMb kept[BPIC ? 1 : MC_SORT_KEEP];
    if (keep) {
#pragma unroll
        for (int j = 0; j < (BPIC ? 1 : MC_SORT_KEEP); j++) {
            const int mbi = tid + j * MC_SORT_THREADS;
            kept[j].info = 0;
            if (mbi < g.n_mb) {
                kept[j] = mc_classify<BPIC>(pd, g, mc_load<BPIC>(pd, mbi), mbi, inv_mbw, (int)ml.band_log2, 0); mc_count(ctx[0], kept[j]);
                is_intra[mbi] = (uint8_t)!(kept[j].info & MCMB_INTER);
            }
        }
    } else {
        for (int mbi = tid; mbi < g.n_mb; mbi += MC_SORT_THREADS) {
            const McIn in = mc_load<BPIC>(pd, mbi);
            is_intra[mbi] = (uint8_t)(P264_MB_IS_INTRA(in.rec.x & 255) != 0);
#pragma unroll
            for (int ps = 0; ps < NP; ps++) if (ps < n_pass) mc_count(ctx[ps], mc_classify<BPIC>(pd, g, in, mbi, inv_mbw, (int)ml.band_log2, ps));
        }
    }
    __syncthreads();
    // segment starts: every thread sums the padded counts in front of its key (a few hundred LDS reads at most), notes the
    // key's class bits for each of its chunks, and the thread of a list's last key writes the number of chunks in use
#pragma unroll
    for (int ps = 0; ps < NP; ps++) {
        if (ps >= n_pass) break;
        const uint32_t *cn = cnt + ps * MC_KEY_SLOTS;
        uint32_t *po = pos + ps * MC_KEY_SLOTS;
        const int L0 = ps * ML_LISTS;
        for (int k = tid; k < b_end; k += MC_SORT_THREADS) {
            const int l = k < b_yq ? ML_YM : k < b_cm ? ML_YQ : k < b_cq ? ML_CM : ML_CQ;
            const int first = l == ML_YM ? b_ym : l == ML_YQ ? b_yq : l == ML_CM ? b_cm : b_cq;
            const int kk = k - first, nk = l < ML_CM ? nky : nkc;
            const uint32_t per = (uint32_t)mc_chunk_items(l);
            uint32_t start = 0;
            for (int j = 0; j < kk; j++) start += (cn[first + j] + per - 1) / per;
            const uint32_t n = (cn[k] + per - 1) / per;                         // chunks of this key, from chunk `start`
            po[k] = start * per;
            AS1 uint8_t *cls = glob((uint8_t *)(out + (l == ML_YM ? ml.off_cls[L0 + ML_YM] : l == ML_YQ ? ml.off_cls[L0 + ML_YQ] : l == ML_CM ? ml.off_cls[L0 + ML_CM] : ml.off_cls[L0 + ML_CQ])));
            const uint8_t v = (uint8_t)(kk % mc_list_keys(l));
            for (uint32_t c = 0; c < n; c++) cls[start + c] = v;
            if (kk == nk - 1) gstore1(out + L0 + l, start + n);
        }
    }
    __syncthreads();
    if (keep) {
#pragma unroll
        for (int j = 0; j < (BPIC ? 1 : MC_SORT_KEEP); j++) mc_scatter(ctx[0], kept[j], tid + j * MC_SORT_THREADS, g, inv_mbw);
    } else {
        for (int mbi = tid; mbi < g.n_mb; mbi += MC_SORT_THREADS) {
            const McIn in = mc_load<BPIC>(pd, mbi);
#pragma unroll
            for (int ps = 0; ps < NP; ps++) if (ps < n_pass) mc_scatter(ctx[ps], mc_classify<BPIC>(pd, g, in, mbi, inv_mbw, (int)ml.band_log2, ps), mbi, g, inv_mbw);
        }
    }
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < NP; ps++) {
        if (ps >= n_pass) break;
        const uint32_t *po = pos + ps * MC_KEY_SLOTS;
        const int L0 = ps * ML_LISTS;
        for (int k = tid; k < b_end; k += MC_SORT_THREADS) {                  // padding entries behind every segment
            const int l = k < b_yq ? ML_YM : k < b_cm ? ML_YQ : k < b_cq ? ML_CM : ML_CQ;
            const uint32_t per = (uint32_t)mc_chunk_items(l), end = po[k];
            // (selects on k itself: as a chain on l the compiler builds a four-entry table in scratch memory and indexes it)
            uint32_t lo = ml.off_list[L0 + ML_CQ];
            if (k < b_cq) lo = ml.off_list[L0 + ML_CM];
            if (k < b_cm) lo = ml.off_list[L0 + ML_YQ];
            if (k < b_yq) lo = ml.off_list[L0 + ML_YM];
            for (uint32_t p = end; p < (end + per - 1) / per * per; p++) {
                if (ps) gstore4(out + lo + 4u * p, make_uint4(0xffffffffu, 0, 0, 0));
                else gstore2(out + lo + 2u * p, make_uint2(0xffffffffu, 0));
            }
        }
    }
}
// batches without B pictures (the two-list classification costs registers the sort of a P picture does not have to pay for)
__global__ __launch_bounds__(MC_SORT_THREADS)
void k_mc_sort(const PicDev *__restrict__ pics, uint32_t *__restrict__ mc_all, Geom g, McLayout ml, uint32_t inv_mbw, uint8_t *__restrict__ is_intra)
{
    __shared__ uint32_t cnt[MC_KEY_SLOTS], pos[MC_KEY_SLOTS];
    mc_sort_picture<false>(pics, mc_all, g, ml, inv_mbw, cnt, pos, is_intra);
}
// batches with B pictures
#ifndef MC_SORT_B_WAVES_PER_EU
#define MC_SORT_B_WAVES_PER_EU 4       // 78 registers, no scratch, one workgroup per CU (round 4 shipped 8: 64 registers of which 9 spilled, two workgroups
                                       // per CU - the MC stage of a B launch 5.44 -> 5.37 ms, scratch/r4_sortb_occ.sh; given back for a binary without scratch)
#endif
__global__ __launch_bounds__(MC_SORT_THREADS, MC_SORT_B_WAVES_PER_EU)
void k_mc_sort_b(const PicDev *__restrict__ pics, uint32_t *__restrict__ mc_all, Geom g, McLayout ml, uint32_t inv_mbw, uint8_t *__restrict__ is_intra)
{
    __shared__ uint32_t cnt[2 * MC_KEY_SLOTS], pos[2 * MC_KEY_SLOTS];
    mc_sort_picture<true>(pics, mc_all, g, ml, inv_mbw, cnt, pos, is_intra);
}

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }
__device__ __forceinline__ s16x2 as_s16x2(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ uint32_t as_u32(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
// byte-parallel (a + b + 1) >> 1 (pixel_avg, core/mc.c:58-74)
__device__ __forceinline__ uint32_t avg4(uint32_t a, uint32_t b) { return (a | b) - (((a ^ b) & 0xfefefefeu) >> 1); }
__device__ __forceinline__ uint32_t pack4(int a, int b, int c, int d) { return (uint32_t)a | ((uint32_t)b << 8) | ((uint32_t)c << 16) | ((uint32_t)d << 24); }
__device__ __forceinline__ uint32_t sel32(bool c, uint32_t a, uint32_t b) { return c ? a : b; }

__device__ __forceinline__ rsrc_t make_rsrc(const void *base, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint32_t bload(rsrc_t r, uint32_t off) { return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0); }
__device__ __forceinline__ void bstore(rsrc_t r, uint32_t off, uint32_t v) { __builtin_amdgcn_raw_buffer_store_b32((int)v, r, (int)off, 0, 0); }
__device__ __forceinline__ u32x4 bload4(rsrc_t r, uint32_t off) { return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0)); }
// reconstructed samples are written once and not read again by this stage: non-temporal stores keep them from pushing
// reference lines out of L2 (measured: -3 % on the stage)
#ifndef MC_ST_AUX
#define MC_ST_AUX 2
#endif
__device__ __forceinline__ u32x2 bload2(rsrc_t r, uint32_t off) { return __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0)); }
// (the 8-byte stores of quadrant items fill half a 16-byte row each: left to L2 to merge - non-temporal they cost 0.27 GB of
// extra HBM writes per launch)
#ifndef MC_ST2_AUX
#define MC_ST2_AUX 0
#endif
__device__ __forceinline__ void bstore2(rsrc_t r, uint32_t off, uint32_t a, uint32_t b)
{
    if (EXPM_NO_STORE) { asm volatile("" :: "v"(a), "v"(b), "v"(off)); return; }
    const u32x2 v = { a, b }; __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)off, 0, MC_ST2_AUX);
}
// a reference-window load (timing switch EXPM_NO_WINDOW: no load, the offset stands in for the data)
__device__ __forceinline__ u32x4 wload4(rsrc_t r, uint32_t off)
{
    if (EXPM_NO_WINDOW) { u32x4 v = { off, off, off, off }; asm volatile("" : "+v"(v)); return v; }
    return bload4(r, off);
}
// the value of lane ^ 1 / lane ^ 2 (inside a group of four lanes: DPP quad_perm, no LDS traffic)
__device__ __forceinline__ uint32_t lane_xor1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t lane_xor2(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true); }
// 4x4 dword transpose among the four lanes of a quad: t[k] of lane L = x[L] of lane k (two exchange steps)
__device__ __forceinline__ void quad_transpose(uint32_t (&t)[4], const uint32_t (&x)[4], int lane)
{
    const bool b0 = lane & 1, b1 = lane & 2;
    uint32_t a[4];
#pragma unroll
    for (int i = 0; i < 4; i += 2) {                       // with lane ^ 1: a[i + j] = x[i + b0] of lane (L & ~1) + j
        const uint32_t r = lane_xor1(b0 ? x[i] : x[i + 1]);
        a[i] = b0 ? r : x[i]; a[i + 1] = b0 ? x[i + 1] : r;
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {                          // with lane ^ 2
        const uint32_t r = lane_xor2(b1 ? a[j] : a[2 + j]);
        t[j] = b1 ? r : a[j]; t[2 + j] = b1 ? a[2 + j] : r;
    }
}
__device__ __forceinline__ void bstore4(rsrc_t r, uint32_t off, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    if (EXPM_NO_STORE) { asm volatile("" :: "v"(a), "v"(b), "v"(c), "v"(d), "v"(off)); return; }
    const u32x4 v = { a, b, c, d }; __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)off, 0, MC_ST_AUX);
}

// ------------------------------------------------------------------------------------------
// reference windows in registers
// ------------------------------------------------------------------------------------------
// d[r][k] = the aligned dword k of window row r.  The window starts at sample (xw, yw); xa = xw & ~3 is the first dword.
// Rows R0 .. R0+NR-1 and NC dword columns are fetched.  Two sources:
// LWin - out of the work item's image in LDS (see the file header): one address register, rows and columns as immediates.
// GWin - straight from memory with clamped coordinates; only for quadrants whose 4x4 blocks have different vectors
//        (sub-8x8 partitions), where there is no common window.
//
// Luma work items: MB = a whole macroblock (16 lanes, window 21 rows x up to 3 strips) or an 8x8 quadrant (4 lanes, 13 rows x 2
// strips).  The image holds only the dword columns the item reads, ROTATED so that column 0 is the dword of the first
// sample the class needs (x0 = xs & ~3): a lane's reads are at column (its block column + k) whatever the vector, and pitch
// and item placement are chosen so that every read instruction of a wavefront is free of bank conflicts BY CONSTRUCTION.  For
// ds_read_b32 / ds_read2_b32 and every ds_write the LDS of gfx950 has 32 banks of 4 bytes and serves a wavefront as two groups
// of 32 lanes (MI355X_MICROARCH.md, LDS): the 32 lanes of a group must land on 32 different banks.
//   macroblock items: dword = item * 208 + (4 by + r) * 9 + bx + k   ->  bank = 16 item + 4 by + bx + const (two items per group);
//   quadrant items:   dword = (item / 2) * 264 + (item % 2) * 130 + (4 ly + r) * 9 + lx + k
//                                                                    ->  bank = 8 (item / 2) + 2 (item % 2) + 4 ly + lx + const
//                     (eight items per group).  Rounds 3 - 5 had a pitch of 8 dwords and 130 dwords per item, laid out for 64
//                     banks across all 64 lanes: 32 ly fell away modulo 32 and the two block rows of every item met on the
//                     same banks in every read - 150 M of the kernel's 271 M bank-conflict cycles per launch (scratch/r5_ldsconf.sh).
// (With the strips stored side by side at a pitch of 52 / 36 bytes, the lanes of different items met on the same banks at
// random: LDS bank-conflict cycles were twice the LDS issue cycles of these kernels.)  The staging stores of a 16-byte row
// piece go to columns 4 s - dx .. 4 s - dx + 3; columns outside 0 .. 5 (0 .. 3) fall into the row's padding or the padding of
// the row above - nobody reads them.  LEAD bytes in front of the first item take the negative columns of its first row.
template <bool MB> struct YItem {
    static constexpr int LANES = MB ? 16 : 4, PER_WAVE = 64 / LANES, STRIPS = MB ? 3 : 2, ROWS = MB ? 21 : 13;
    static constexpr int PITCH = 36, BYTES = MB ? 832 : 520, LEAD = 16;
    static constexpr int PAIR = 1056;                       // quadrant items: bytes per pair of items (264 dwords)
    static constexpr int WAVE_BYTES = MB ? PER_WAVE * BYTES : (PER_WAVE / 2) * PAIR;
    static_assert(ROWS * PITCH + 12 <= BYTES && (MB || BYTES + ROWS * PITCH + 12 <= PAIR), "item image");
    // first byte of item i (counted over the workgroup's wavefronts) behind LEAD
    __device__ static __forceinline__ int item_off(int i) { return MB ? i * BYTES : (i >> 1) * PAIR + (i & 1) * BYTES; }
};

// Stage rows R0S .. R0S+NRS-1 of the window whose first needed sample is (xs, wy); li = lane inside the item.  Piece p = strip
// p / NRS, row p % NRS: consecutive lanes fetch consecutive rows of one strip (64 contiguous bytes per four lanes).  A
// macroblock window reaches into its third strip only when it starts in the last dword of a strip: `third` says so.
// CLAMP: coordinates clamped to the picture = the reference's replicated borders (core/frame.c:183-222, A-Q9): a strip
// that lies outside the picture becomes the replicated first (last) sample of the row.
// Every piece is requested by the SAME straight-line sequence of loads, all in flight together, one wait: a piece a lane does
// not need (the third strip of an item that does not reach into it) is requested at an offset beyond the buffer, which the
// bounds check of the descriptor answers with zeros without going to memory.  (Written as `if (needed) load`, the compiler
// wrapped every such load into a branch of its own with a wait for the data inside it: up to three memory round trips one
// after the other per chunk instead of one.)  NS = strips the class can reach into (2: copy / vertical, which read no columns
// left of the blocks).
// "all of these loads have been issued, and their data is used HERE": keeps the compiler from sinking a load into the conditional
// block that stores its data (behind the waits and the stores of the loads in front of it)
__device__ __forceinline__ void loads_land(u32x4 &v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
#define MC_OOB 0xffffff00u             // raw buffer offset beyond any frame store (< 4 GiB, p264hip_create): reads return 0
template <bool MB, int R0S, int NRS, bool CLAMP, int NS>
__device__ __forceinline__ void stage_luma(uint8_t *img, rsrc_t rs, uint32_t roff, const Geom &g, int xs, int wy, int li, bool third)
{
    typedef YItem<MB> I;
    static_assert(NS == 2 || (MB && NS == 3), "strips");
    constexpr int NP = NS * NRS, NJ = (NP + I::LANES - 1) / I::LANES;
    const int sA = xs >> 4, dx4 = xs & 12;                  // first strip, byte offset of the first needed dword inside it
    u32x4 v[NJ];
    int dst[NJ];
    bool on[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int p = min(li + I::LANES * j, NP - 1), s = p >= 2 * NRS ? 2 : p >= NRS ? 1 : 0, row = R0S + p - s * NRS;
        dst[j] = row * I::PITCH + s * 16 - dx4;
        on[j] = NS == 2 || s < 2 || third;
        if (!CLAMP) {
            const uint32_t off = roff + strip_mul(sA + s, g.ystrip) + (uint32_t)((wy + row) * 16);
            v[j] = wload4(rs, on[j] ? off : MC_OOB);
        } else {
            const int st = sA + s, sc = clip3i(st, 0, g.mb_w - 1);
            const uint32_t off = roff + strip_mul(sc, g.ystrip) + (uint32_t)(clip3i(wy + row, 0, g.h - 1) * 16);
            v[j] = wload4(rs, on[j] ? off : MC_OOB);
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; j++) loads_land(v[j]);
    if (CLAMP) {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int p = min(li + I::LANES * j, NP - 1), s = p >= 2 * NRS ? 2 : p >= NRS ? 1 : 0, st = sA + s;
            u32x4 t = v[j];
            if (st < 0) { const uint32_t e = perm(t.x, t.x, 0x00000000u); t.x = t.y = t.z = t.w = e; }
            if (st >= g.mb_w) { const uint32_t e = perm(t.w, t.w, 0x03030303u); t.x = t.y = t.z = t.w = e; }
            v[j] = t;
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        if (on[j]) {
            uint32_t *o = (uint32_t *)(img + dst[j]);       // (4-byte aligned: pairs of dwords)
            o[0] = v[j].x; o[1] = v[j].y; o[2] = v[j].z; o[3] = v[j].w;
        }
    }
}
template <int PITCH> struct LWin {
    // (Round 5 tried dword reads at the window's own byte address - gfx950's LDS takes unaligned reads - to drop the v_alignbyte per
    // window dword: 6 % fewer vector instructions in the luma roles, and the MC stage went from 4.95 to 5.54 ms per 2048 pictures:
    // the compiler merges them into unaligned ds_read_b64 / b96 and the LDS pipe serves those far below its aligned rate.)
    const uint8_t *img; int x0, y0;                        // the image's sample (0,0) is (x0, y0) of the reference
    template <int R0, int NR, int NC> __device__ __forceinline__ void load(uint32_t (&d)[9][3], int xw, int yw) const
    {
        const uint8_t *b = img + __mul24(yw - y0, PITCH) + ((xw & ~3) - x0);
#pragma unroll
        for (int r = R0; r < R0 + NR; r++)
#pragma unroll
            for (int k = 0; k < NC; k++) d[r][k] = *(const uint32_t *)(b + r * PITCH + 4 * k);
    }
};
struct GWin {
    rsrc_t rs; uint32_t roff; const Geom &g;
    template <int R0, int NR, int NC> __device__ __forceinline__ void load(uint32_t (&d)[9][3], int xw, int yw) const
    {
        const int xa = xw & ~3;
        uint32_t col[NC], sel[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) {
            const int x = xa + 4 * k, xc = clip3i(x, 0, g.w - 4);
            col[k] = roff + strip_mul(xc >> 4, g.ystrip) + (uint32_t)(xc & 15);
            sel[k] = x < 0 ? 0x00000000u : x >= g.w ? 0x03030303u : 0x03020100u;
        }
#pragma unroll
        for (int r = R0; r < R0 + NR; r++) {
            const uint32_t ro = (uint32_t)clip3i(yw + r, 0, g.h - 1) * 16u;
#pragma unroll
            for (int k = 0; k < NC; k++) { const uint32_t v = bload(rs, col[k] + ro); d[r][k] = perm(v, v, sel[k]); }
        }
    }
};

// horizontal 6-tap sums (core/mc.c:53-56) for 4 adjacent samples; n0..n2 hold window bytes 0..11, output sample i uses
// bytes i..i+5.  Samples are taken as (s - 128) in int8: the taps sum to 32, so the true sum is the dot product + 4096;
// `bias` = 4096 + rounding term.
__device__ __forceinline__ void tap_h4(uint32_t n0, uint32_t n1, uint32_t n2, int bias, int t[4])
{
    // Sample i needs bytes i..i+5.  Instead of shifting the window to each sample (two v_alignbyte per sample) the TAPS are
    // shifted: every sample is a sum of dot products of the same aligned dwords with its own constant vectors.
    n0 ^= 0x80808080u; n1 ^= 0x80808080u; n2 ^= 0x80808080u;
    t[0] = __builtin_amdgcn_sdot4((int)n0, 0x1414fb01, __builtin_amdgcn_sdot4((int)n1, 0x000001fb, bias, false), false);   // (1,-5,20,20 | -5,1,0,0)
    t[1] = __builtin_amdgcn_sdot4((int)n0, 0x14fb0100, __builtin_amdgcn_sdot4((int)n1, 0x0001fb14, bias, false), false);   // (0,1,-5,20 | 20,-5,1,0)
    t[2] = __builtin_amdgcn_sdot4((int)n0, (int)0xfb010000u, __builtin_amdgcn_sdot4((int)n1, 0x01fb1414, bias, false), false);   // (0,0,1,-5 | 20,20,-5,1)
    t[3] = __builtin_amdgcn_sdot4((int)n0, 0x01000000, __builtin_amdgcn_sdot4((int)n1, (int)0xfb1414fbu, __builtin_amdgcn_sdot4((int)n2, 0x00000001, bias, false), false), false);   // (0,0,0,1 | -5,20,20,-5 | 1,0,0,0)
}
// (t >> sh) clipped to a byte, four at once: gfx950's v_ashr_pk_u8_i32 shifts, saturates and packs two values per
// instruction (result bits 16..31 are not defined: only its low two bytes are used)
template <int SH> __device__ __forceinline__ uint32_t round_pack4(const int t[4])
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_ashr_pk_u8_i32(t[0], t[1], SH), hi = (uint32_t)__builtin_amdgcn_ashr_pk_u8_i32(t[2], t[3], SH);
    return perm(hi, lo, 0x05040100u);
}
// window bytes 0..11 of a row, the window starting at byte s of dword 0
__device__ __forceinline__ void align_row(const uint32_t (&w)[3], uint32_t s, uint32_t &n0, uint32_t &n1, uint32_t &n2)
{
    n0 = alignbyte(w[1], w[0], s); n1 = alignbyte(w[2], w[1], s); n2 = w[2] >> (8 * s);
}
// vertical 6-tap of four columns over six rows of samples (mc_hv, core/mc.c:186-199), packed 16-bit
__device__ __forceinline__ uint32_t tap_v4(uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, uint32_t r4, uint32_t r5)
{
    // The pair sums of samples are 32-bit adds of two non-negative 16-bit halves (nothing carries across; v_add_u32 issues in 2.4
    // cycles per wavefront on gfx950, the packed add in 4.3 - scratch/r4_rates/), the rounding term rides on the first of them;
    // the taps are two packed multiply-adds.
    const uint32_t M = 0x00ff00ffu;
    const s16x2 c20 = { 20, 20 }, cm5 = { -5, -5 };
    const uint32_t a05 = (r0 & M) + (r5 & M) + 0x00100010u, a23 = (r2 & M) + (r3 & M), a14 = (r1 & M) + (r4 & M);
    const uint32_t b05 = ((r0 >> 8) & M) + ((r5 >> 8) & M) + 0x00100010u, b23 = ((r2 >> 8) & M) + ((r3 >> 8) & M), b14 = ((r1 >> 8) & M) + ((r4 >> 8) & M);
    s16x2 a = cm5 * as_s16x2(a14) + (c20 * as_s16x2(a23) + as_s16x2(a05));
    s16x2 b = cm5 * as_s16x2(b14) + (c20 * as_s16x2(b23) + as_s16x2(b05));
    // (v >> 5) clipped to a byte: v_sat_pk_u8_i16 saturates both halves of a pair into bytes 0 and 1 (two instructions per pair
    // instead of shift, max, min; one v_perm interleaves the even and the odd samples)
    a = a >> 5; b = b >> 5;
    return perm(sat_pk_u8_i16(as_u32(b)), sat_pk_u8_i16(as_u32(a)), 0x05010400u);
}

// ------------------------------------------------------------------------------------------
// bi-prediction (B pictures): the two ways the reference combines two predictions, four samples at a
// time; used by the two-list class of the luma and chroma quadrant kernels.  pixel_avg_wxh (core/mc.c:76-88) is the byte-parallel rounding average above; the implicit-weight
// form pixel_avg_weight_wxh (core/mc.c:106-132): clip((a * w1 + b * (64 - w1) + 32) >> 6), weights from -64 to 128.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t bipred_avg4(uint32_t a, uint32_t b) { return avg4(a, b); }
__device__ __forceinline__ uint32_t bipred_weight4(uint32_t a, uint32_t b, int w1)
{
    const int w2 = 64 - w1;
    int v[4];
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = (int)((a >> (8 * i)) & 255u) * w1 + (int)((b >> (8 * i)) & 255u) * w2 + 32;
    return round_pack4<6>(v);
}
// ---- the seven phase classes: out[y] = the four samples of row y of the lane's 4x4 block -----------------------------
// (ix, iy) = integer position of the block's first sample in the reference, (fx, fy) = quarter-pel phase.
template <class W> __device__ __forceinline__ void mc_copy(uint32_t (&out)[4], const W &w, int ix, int iy)
{
    uint32_t d[9][3];
    w.template load<0, 4, 2>(d, ix, iy);
    const uint32_t s = (uint32_t)ix & 3u;
#pragma unroll
    for (int y = 0; y < 4; y++) out[y] = alignbyte(d[y][1], d[y][0], s);
}
template <class W> __device__ __forceinline__ void mc_h(uint32_t (&out)[4], const W &w, int ix, int iy, int fx)
{   // fx = 1: avg(G, h), 2: h, 3: avg(h, G one to the right)
    uint32_t d[9][3];
    w.template load<0, 4, 3>(d, ix - 2, iy);
    const uint32_t s = (uint32_t)(ix - 2) & 3u, gs = 2u + (uint32_t)(fx == 3);
#pragma unroll
    for (int y = 0; y < 4; y++) {
        uint32_t n0, n1, n2; int t[4];
        align_row(d[y], s, n0, n1, n2);
        tap_h4(n0, n1, n2, 4096 + 16, t);
        const uint32_t hh = round_pack4<5>(t);
        out[y] = sel32(fx == 2, hh, avg4(hh, alignbyte(n1, n0, gs)));
    }
}
template <class W> __device__ __forceinline__ void mc_v(uint32_t (&out)[4], const W &w, int ix, int iy, int fy)
{   // fy = 1: avg(v, G), 2: v, 3: avg(G one down, v)
    uint32_t d[9][3], c[9];
    w.template load<0, 9, 2>(d, ix, iy - 2);
    const uint32_t s = (uint32_t)ix & 3u;
#pragma unroll
    for (int r = 0; r < 9; r++) c[r] = alignbyte(d[r][1], d[r][0], s);
#pragma unroll
    for (int y = 0; y < 4; y++) {
        const uint32_t vv = tap_v4(c[y], c[y + 1], c[y + 2], c[y + 3], c[y + 4], c[y + 5]);
        out[y] = sel32(fy == 2, vv, avg4(vv, sel32(fy == 3, c[y + 3], c[y + 2])));
    }
}
template <class W> __device__ __forceinline__ void mc_diag(uint32_t (&out)[4], const W &w, int ix, int iy, int fx, int fy)
{   // avg(h of row y + (fy == 3), v of column x + (fx == 3)); rows are consumed one at a time to keep the live set small
    uint32_t d[9][3], c[9], hh[5];
    w.template load<0, 9, 3>(d, ix - 2, iy - 2);
    const uint32_t s = (uint32_t)(ix - 2) & 3u, vs = 2u + (uint32_t)(fx == 3);
#pragma unroll
    for (int r = 0; r < 9; r++) {
        uint32_t n0, n1, n2;
        align_row(d[r], s, n0, n1, n2);
        c[r] = alignbyte(n1, n0, vs);
        if (r >= 2 && r <= 6) { int t[4]; tap_h4(n0, n1, n2, 4096 + 16, t); hh[r - 2] = round_pack4<5>(t); }
    }
#pragma unroll
    for (int y = 0; y < 4; y++)
        out[y] = avg4(sel32(fy == 3, hh[y + 1], hh[y]), tap_v4(c[y], c[y + 1], c[y + 2], c[y + 3], c[y + 4], c[y + 5]));
}
// centre position (mc_hc, core/mc.c:200-235): horizontal sums of nine rows, unrounded, then the vertical filter on them,
// (sum + 512) >> 10.  Each row's sum carries + 16, which the vertical taps (sum 32) turn into the + 512 - and which is
// also the rounding term of the horizontal half-pel sample of that row.  A row's sums are folded into the (up to four)
// output rows they belong to as soon as they exist: 16 accumulators instead of 36 sums kept alive.
// WITH: 0 centre only, 1 avg with h of row y + (fy == 3), 2 avg with v of column x + (fx == 3).
template <class W, int WITH> __device__ __forceinline__ void mc_centre(uint32_t (&out)[4], const W &w, int ix, int iy, int fx, int fy)
{
    uint32_t d[9][3];
    w.template load<0, 9, 3>(d, ix - 2, iy - 2);
    const uint32_t s = (uint32_t)(ix - 2) & 3u;
    int acc[4][4];
    uint32_t c[9], hsel[4];
#pragma unroll
    for (int r = 0; r < 9; r++) {
        uint32_t n0, n1, n2; int t[4];
        align_row(d[r], s, n0, n1, n2);
        tap_h4(n0, n1, n2, 4096 + 16, t);
        if (WITH == 2) c[r] = alignbyte(n1, n0, 2u + (uint32_t)(fx == 3));
#pragma unroll
        for (int y = 0; y < 4; y++) {
            const int k = r - y;                           // tap of output row y this window row meets
            if (k < 0 || k > 5) continue;
            const int cv = (k == 0 || k == 5) ? 1 : (k == 1 || k == 4) ? -5 : 20;
#pragma unroll
            for (int x = 0; x < 4; x++) acc[y][x] = k == 0 ? t[x] : acc[y][x] + cv * t[x];
        }
        if (WITH == 1 && r >= 2 && r <= 6) {               // horizontal half-pel sample of window row r: output row r-2 (fy = 1) or r-3 (fy = 3)
            const uint32_t hh = round_pack4<5>(t);
            if (r <= 5) hsel[r - 2] = hh;
            if (r >= 3) hsel[r - 3] = sel32(fy == 3, hh, hsel[r - 3]);
        }
    }
#pragma unroll
    for (int y = 0; y < 4; y++) {
        const uint32_t cc = round_pack4<10>(acc[y]);
        out[y] = WITH == 0 ? cc : WITH == 1 ? avg4(cc, hsel[y]) : avg4(cc, tap_v4(c[y], c[y + 1], c[y + 2], c[y + 3], c[y + 4], c[y + 5]));
    }
}

template <class W> __device__ __forceinline__ void mc_luma_class(int pc, uint32_t (&out)[4], const W &w, int ix, int iy, int fx, int fy)
{
    switch (pc) {                                          // wave-uniform
    case PC_COPY: mc_copy(out, w, ix, iy); break;
    case PC_H:    mc_h(out, w, ix, iy, fx); break;
    case PC_V:    mc_v(out, w, ix, iy, fy); break;
    case PC_DIAG: mc_diag(out, w, ix, iy, fx, fy); break;
    case PC_C:    mc_centre<W, 0>(out, w, ix, iy, fx, fy); break;
    case PC_CH:   mc_centre<W, 1>(out, w, ix, iy, fx, fy); break;
    default:      mc_centre<W, 2>(out, w, ix, iy, fx, fy); break;
    }
}

// ------------------------------------------------------------------------------------------
// residual of one 4x4 block, entirely in the lane's registers
// ------------------------------------------------------------------------------------------
// dequantisation scale of a position class (core/set.c:27-35, without the factor 16 the reference folds in)
__device__ __forceinline__ uint32_t dq_s(int cls, int rem)
{
    const uint32_t k = cls == 0 ? (10u | 11u << 5 | 13u << 10 | 14u << 15 | 16u << 20 | 18u << 25)
                     : cls == 1 ? (13u | 14u << 5 | 16u << 10 | 18u << 15 | 20u << 20 | 23u << 25)
                                : (16u | 18u << 5 | 20u << 10 | 23u << 15 | 25u << 20 | 29u << 25);
    return (k >> (5 * rem)) & 31u;
}
// half `ha` of word pa into the low half, half `hb` of word pb into the high half
template <int HA, int HB> __device__ __forceinline__ uint32_t pick2(uint32_t pa, uint32_t pb)
{
    return perm(pb, pa, (uint32_t)(2 * HA) | (uint32_t)(2 * HA + 1) << 8 | (uint32_t)(4 + 2 * HB) << 16 | (uint32_t)(5 + 2 * HB) << 24);
}
// Levels in scan order, two per word (lv[j] = levels 2j, 2j+1) -> the block in COLUMNS: col[x][0] = (d[0][x], d[1][x]),
// col[x][1] = (d[2][x], d[3][x]) (zig-zag, decoder/macroblock.c:602-603).  AC: the 15 levels sit at scan positions 1..15
// (level k-1 at position k); position 0 is filled by the caller.
template <bool AC> __device__ __forceinline__ void unscan_cols(const uint32_t (&lv)[8], uint32_t (&col)[4][2])
{
    // scan index of raster position (y,x): 0 1 5 6 / 2 4 7 12 / 3 8 11 13 / 9 10 14 15
    if (!AC) {
        col[0][0] = pick2<0, 0>(lv[0], lv[1]);  col[0][1] = pick2<1, 1>(lv[1], lv[4]);      // s0 s2 | s3 s9
        col[1][0] = pick2<1, 0>(lv[0], lv[2]);  col[1][1] = pick2<0, 0>(lv[4], lv[5]);      // s1 s4 | s8 s10
        col[2][0] = pick2<1, 1>(lv[2], lv[3]);  col[2][1] = pick2<1, 0>(lv[5], lv[7]);      // s5 s7 | s11 s14
        col[3][0] = pick2<0, 0>(lv[3], lv[6]);  col[3][1] = pick2<1, 1>(lv[6], lv[7]);      // s6 s12 | s13 s15
    } else {   // level index = scan index - 1
        col[0][0] = lv[0] & 0xffff0000u;        col[0][1] = pick2<0, 0>(lv[1], lv[4]);      // -  l1 | l2 l8
        col[1][0] = pick2<0, 1>(lv[0], lv[1]);  col[1][1] = pick2<1, 1>(lv[3], lv[4]);      // l0 l3 | l7 l9
        col[2][0] = pick2<0, 0>(lv[2], lv[3]);  col[2][1] = pick2<0, 1>(lv[5], lv[6]);      // l4 l6 | l10 l13
        col[3][0] = pick2<1, 1>(lv[2], lv[5]);  col[3][1] = pick2<0, 0>(lv[6], lv[7]);      // l5 l11 | l12 l14
    }
}
// dequant_4x4 (core/quant.c:66-99) on the column form.  For every QP the reference's value is level * (scale << qp/6)
// truncated to int16: below QP 24 its rounding shift divides a multiple of 16 exactly (16 * scale * level >> n, n <= 4).
__device__ __forceinline__ void dequant_cols(uint32_t (&col)[4][2], int qp)
{
    const int per = (qp * 43) >> 8, rem = qp - per * 6;
    const uint32_t m0 = dq_s(0, rem) << per, m1 = dq_s(1, rem) << per, m2 = dq_s(2, rem) << per;
    const u16x2 even = __builtin_bit_cast(u16x2, m0 | (m1 << 16)), odd = __builtin_bit_cast(u16x2, m1 | (m2 << 16));
#pragma unroll
    for (int x = 0; x < 4; x++)
#pragma unroll
        for (int j = 0; j < 2; j++)
            col[x][j] = __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, col[x][j]) * ((x & 1) ? odd : even)));
}
// add4x4_idct (core/dct.c:205-247): first pass along the rows in packed 16-bit (int16 stores, as the reference's tmp),
// second pass along the columns in 32-bit ((sum + 32) >> 6 is taken before the int16 store), added to the prediction.
__device__ __forceinline__ void idct_add(const uint32_t (&col)[4][2], uint32_t (&px)[4])
{
    s16x2 T[4][2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const s16x2 c0 = as_s16x2(col[0][j]), c1 = as_s16x2(col[1][j]), c2 = as_s16x2(col[2][j]), c3 = as_s16x2(col[3][j]);
        const s16x2 s02 = c0 + c2, d02 = c0 - c2, s13 = c1 + (c3 >> 1), d13 = (c1 >> 1) - c3;
        T[0][j] = s02 + s13; T[1][j] = d02 + d13; T[2][j] = d02 - d13; T[3][j] = s02 - s13;
    }
    // (sum + 32) >> 6 added to the prediction and clipped = (sum + 32 + 64 * prediction) >> 6 clipped: shift, saturation
    // and packing in one instruction per two samples
    int res[4][4];
#pragma unroll
    for (int x = 0; x < 4; x++) {
        const int a0 = T[x][0].x, a1 = T[x][0].y, a2 = T[x][1].x, a3 = T[x][1].y;
        const int s02 = a0 + a2 + 32, d02 = a0 - a2 + 32, s13 = a1 + (a3 >> 1), d13 = (a1 >> 1) - a3;
        res[0][x] = s02 + s13; res[1][x] = d02 + d13; res[2][x] = d02 - d13; res[3][x] = s02 - s13;
    }
#pragma unroll
    for (int y = 0; y < 4; y++) {
        const uint32_t p = px[y];
        const int v[4] = { res[y][0] + (int)((p & 255u) << 6), res[y][1] + (int)(((p >> 8) & 255u) << 6),
                           res[y][2] + (int)(((p >> 16) & 255u) << 6), res[y][3] + (int)((p >> 24) << 6) };
        px[y] = round_pack4<6>(v);
    }
}


// XCD-aware block mapping: the dispatcher deals workgroups round-robin over the 8 XCDs; every XCD gets one contiguous
// eighth of the batch, so that the windows of a picture meet in ONE L2 (speed only, never correctness).
__device__ __forceinline__ int xcd_logical_block()
{
    const int per_xcd = gridDim.x >> 3;
    return (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
}

// A list entry as the consumers keep it (first pass: x, y; second pass: all four words), and the macroblock record behind a
// first-pass entry of a chunk with residual (other chunks and padding entries: record 0, one cached request per wavefront -
// asked for without a condition so that the wait for it stays where the record is used).
template <bool PB>
__device__ __forceinline__ uint4 mc_entry_load(const uint32_t *list, int item)
{
    if (PB) return gload4(list + (size_t)item * 4);
    const uint2 t = gload2(list + (size_t)item * 2);
    return make_uint4(t.x, t.y, 0, 0);
}
__device__ __forceinline__ uint3 mc_entry_record(const PicDev *__restrict__ pd, const Geom &g, uint32_t ex, bool resid)
{
    const bool use = resid && (ex & MC_ITEM_MASK) != MC_ITEM_MASK;
    const int mbi = use ? (int)((ex >> 13) & 1023u) * g.mb_w + (int)((ex >> 2) & 2047u) : 0;
    return gload3(pd->mb + mbi);          // (three words: asked for all four, the compiler recycles the unused register at once and waits for the load there)
}

// ------------------------------------------------------------------------------------------
// mc_luma_body<MB, PB> (a role of k_mc / k_mc_second): one wavefront = one chunk of one key: 4 macroblock items of 16 lanes, or 16 quadrant items of 4 lanes;
// the lane is one 4x4 block
// ------------------------------------------------------------------------------------------
template <bool MB, bool PB>
__device__ __forceinline__ void mc_luma_body(uint8_t *images, const uint32_t *ref_tab, const PicDev *__restrict__ pd, const uint32_t *__restrict__ mc, const Geom &g,
                                             const McLayout &ml, int sub, int role_wgs)
{
    typedef YItem<MB> I;
    constexpr int LIST = (MB ? ML_YM : ML_YQ) + (PB ? ML_LISTS : 0);
    const int wave = rfl((int)(threadIdx.x >> 6));
    // A wavefront walks the list's chunks with a stride of all the wavefronts that work on this list of this picture; the key
    // and the list entries of its NEXT chunk are requested before it starts on the current one, so that only the window
    // fetch itself is a memory round trip the wavefront has to sit through.
    const int stride = role_wgs * 4, n_chunks = (int)mc[LIST];
    int chunk = sub * 4 + wave;
    if (chunk >= n_chunks) return;
    const int lane = threadIdx.x & 63, li = lane & (I::LANES - 1), it = lane / I::LANES;
    const uint32_t *cls_w = mc + ml.off_cls[LIST], *list = mc + ml.off_list[LIST];
    const rsrc_t rs = make_rsrc(pd->store, pd->store_bytes);
    uint32_t key_w = cls_w[chunk >> 2];
    uint4 e = mc_entry_load<PB>(list, chunk * I::PER_WAVE + it);
    uint3 rec = make_uint3(0, 0, 0);
    if (!PB) rec = mc_entry_record(pd, g, e.x, (MC_CHUNK_KEY((int)((key_w >> (8 * (chunk & 3))) & 255u)) & MCY_RESID) != 0);
  for (;;) {
    const int next = chunk + stride;
    const bool more = next < n_chunks;
    // (requested without a condition - past the list's end the last chunk once more: under `if (more)` the compiler waits for the
    // data inside the branch, one memory round trip per chunk that is no prefetch at all)
    const int nx = min(next, n_chunks - 1);
    const uint32_t key_w_next = cls_w[nx >> 2];
    const uint4 e_next = mc_entry_load<PB>(list, nx * I::PER_WAVE + it);
    const int key = MC_CHUNK_KEY((int)((key_w >> (8 * (chunk & 3))) & 255u));                   // scalar: the chunk's key bits
    const int pc = key & 7;
    wave_lds_fence();                                      // the previous chunk's image has been read
    const bool valid = (e.x & MC_ITEM_MASK) != MC_ITEM_MASK;
    const int mbx = valid ? (int)((e.x >> 2) & 2047u) : 0, mby = valid ? (int)((e.x >> 13) & 1023u) : 0;
    // block position inside the macroblock
    const int q = MB ? 0 : (valid ? (int)(e.x & 3) : 0);
    const int bx = MB ? (li & 3) : (q & 1) * 2 + (li & 1), by = MB ? (li >> 2) : (q >> 1) * 2 + (li >> 1);
    // chunks with residual: QP, coded-block mask and place in the coefficient stream come with the entry (second pass) or
    // with the macroblock's record, requested while the previous chunk was at work
    int mvp = (int)e.y;
    if (!MB && pc == PC_GEN) mvp = (int)gload1(pd->mv + (mby * g.mb_w + mbx) * 16 + by * 4 + bx);   // sub-8x8 partitions: the block's own vector
    // the reference frame of the entry's (list, index): out of the workgroup's table in LDS (mc_roles) - as a load from the
    // picture's tables it was a memory round trip in front of every chunk's window requests (and the wait for it also waited
    // for the next chunk's entries, requested just before: no prefetch)
    const uint32_t roff = ref_tab[(e.x >> 28) | ((e.x >> 19) & 16u)];
    const int ix = mbx * 16 + bx * 4 + (mv_x(mvp) >> 2), iy = mby * 16 + by * 4 + (mv_y(mvp) >> 2);
    const int fx = mv_x(mvp) & 3, fy = mv_y(mvp) & 3;
    const int blk = blk_at(bx, by);                        // decode-order index: bit of coef_mask, position in the packed stream
    // (filled in where the chunk asks for its levels, BEHIND its window requests: the record is waited for there and nowhere
    // else - read up here, the wait sat in front of the windows, a memory round trip per wavefront at its first chunk)
    unsigned mask = 0; int qp = 0; uint32_t cidx = 0; bool coded = false;
    auto resid_fields = [&]() {
        if (PB) { cidx = e.w & ((1u << MCE_W_SHIFT) - 1u); mask = e.z & 0x03ffffffu; qp = (int)(e.z >> 26); }
        else {
            asm volatile("" : "+v"(rec.x), "+v"(rec.y), "+v"(rec.z));
            cidx = rec.z; mask = (e.x & MCE_Z) ? 0u : (rec.y & 0x03ffffffu); qp = (int)((rec.x >> 8) & 63u);
        }
        coded = valid && ((mask >> blk) & 1);
    };
    if (PB) resid_fields();
    // coded levels (requested right behind the windows: they fly while the prediction is computed)
    uint4 la = make_uint4(0, 0, 0, 0), lb = la;
    // second pass: the list-0 prediction the first pass left in the destination, as the lane will store it (requested in
    // front of the windows: it has arrived when they have)
    const uint32_t dsto = pd->dst_off + mb_luma_off(g, mbx, mby) + (MB ? (uint32_t)((by * 4 + bx) * 16) : (uint32_t)((by * 4 + ((bx & 1) ? 2 : 0)) * 16 + (bx >> 1) * 8));
    u32x4 prev = { 0, 0, 0, 0 };
    if (PB) {                                              // (compile time; padding lanes read beyond the buffer: zeros, no branch)
        const uint32_t po = valid ? dsto : MC_OOB;
        if (MB) prev = bload4(rs, po);
        else { const u32x2 a = bload2(rs, po), b = bload2(rs, po + 16); prev = u32x4{ a.x, a.y, b.x, b.y }; }
    }
    // ---- prediction ----
    uint32_t out[4];
    if (MB || pc != PC_GEN) {
        // the item's window (all its lanes hold the same vector): top-left sample (wx, wy), staged in LDS
        const int ox = MB ? bx * 4 : (li & 1) * 4, oy = MB ? by * 4 : (li >> 1) * 4;
        const int wx = ix - ox - 2, wy = iy - oy - 2;
        uint8_t *img = images + I::LEAD + I::item_off(wave * I::PER_WAVE + it);
        const bool rows_mid = pc <= PC_H;                  // copy / horizontal: no rows above and below the blocks
        // first sample the class reads in a row (copy / vertical: no columns left of the blocks); image column 0 = its dword
        const int xs = (pc == PC_COPY || pc == PC_V) ? wx + 2 : wx;
        // (macroblock items) six dwords from the first one: the third strip is needed when they start in a strip's last dword
        const bool third = MB && (xs & 12) == 12 && !(pc == PC_COPY || pc == PC_V);
        constexpr int RM = MB ? 16 : 8;
        // (scalar branches: one straight-line staging sequence per shape - rows, strips, clamped or not)
        const bool two = !MB || pc == PC_COPY || pc == PC_V;     // the class reads no columns left of the blocks: two strips at most
        constexpr int S3 = MB ? 3 : 2;
        if (!(key & MCY_CLAMP)) {
            if (rows_mid) { if (two) stage_luma<MB, 2, RM, false, 2>(img, rs, roff, g, xs, wy, li, third); else stage_luma<MB, 2, RM, false, S3>(img, rs, roff, g, xs, wy, li, third); }
            else          { if (two) stage_luma<MB, 0, I::ROWS, false, 2>(img, rs, roff, g, xs, wy, li, third); else stage_luma<MB, 0, I::ROWS, false, S3>(img, rs, roff, g, xs, wy, li, third); }
        } else {
            if (rows_mid) { if (two) stage_luma<MB, 2, RM, true, 2>(img, rs, roff, g, xs, wy, li, third); else stage_luma<MB, 2, RM, true, S3>(img, rs, roff, g, xs, wy, li, third); }
            else          { if (two) stage_luma<MB, 0, I::ROWS, true, 2>(img, rs, roff, g, xs, wy, li, third); else stage_luma<MB, 0, I::ROWS, true, S3>(img, rs, roff, g, xs, wy, li, third); }
        }
        if (key & MCY_RESID) {
            if (!PB) resid_fields();
            if (coded) {
                const int16_t *cf = pd->coefs + ((size_t)cidx + coef_slot(mask, blk)) * 16;
                MC_LOAD_LEVELS(cf, la, lb);
            }
        }
        wave_lds_fence();
        const LWin<I::PITCH> w = { img, xs & ~3, wy };
        mc_luma_class(EXPM_LUMA_COPY ? PC_COPY : pc, out, w, ix, iy, fx, fy);
    } else {
        // The vectors differ inside the quadrant (sub-8x8 partitions), every lane has its own window and phase: windows
        // straight from memory, one pass per phase class present among the lanes.
        if (key & MCY_RESID) {
            if (!PB) resid_fields();
            if (coded) {
                const int16_t *cf = pd->coefs + ((size_t)cidx + coef_slot(mask, blk)) * 16;
                MC_LOAD_LEVELS(cf, la, lb);
            }
        }
        // one prediction per lane from reference frame `ro` with vector `mv` (lanes with use = false are left alone)
        auto predict = [&](uint32_t ro, int mv, bool use, uint32_t (&o4)[4]) {
            const GWin w = { rs, ro, g };
            const int px = mbx * 16 + bx * 4 + (mv_x(mv) >> 2), py = mby * 16 + by * 4 + (mv_y(mv) >> 2);
            const int qx = mv_x(mv) & 3, qy = mv_y(mv) & 3;
            const int mine = phase_class(qx, qy);
            o4[0] = o4[1] = o4[2] = o4[3] = 0;
#pragma unroll 1
            for (int c = 0; c < 7; c++) {
                if (__ballot(use && mine == c) == 0) continue;
                // (the window addresses do not depend on c: without this the compiler computes the clamped addresses of every
                // class in front of the loop and keeps all of them alive - far more registers than the rest of the kernel needs)
                int jx = px, jy = py;
                asm volatile("" : "+v"(jx), "+v"(jy));
                uint32_t o[4];
                mc_luma_class(c, o, w, jx, jy, qx, qy);
                if (mine == c) { o4[0] = o[0]; o4[1] = o[1]; o4[2] = o[2]; o4[3] = o[3]; }
            }
        };
        if (key & MCY_CLAMP) predict(roff, mvp, valid, out);
        else {
            // B picture, two lists (core/macroblock.c:525-583): which lists the lane's quadrant uses, a prediction from each,
            // then pixel_avg / pixel_avg_weight (core/mc.c:76-132)
            const int mbi = mby * g.mb_w + mbx;
            const int r0 = (int)glob(pd->ref_idx)[mbi * 4 + q], r1 = (int)glob(pd->ref_idx_l1)[mbi * 4 + q];
            const bool u0 = valid && r0 >= 0, u1 = valid && r1 >= 0;
            const int r0c = min(max(r0, 0), pd->n_ref - 1), r1c = min(max(r1, 0), pd->n_ref_l1 - 1);
            const int mv1 = (int)gload1(pd->mv_l1 + mbi * 16 + by * 4 + bx);
            const uint32_t ro0 = glob(pd->ref_off)[r0c], ro1 = glob(pd->ref_off_l1)[r1c];
            int wgt = 32;
            if (pd->weighted) wgt = (int)glob(pd->bipred_w)[r0c * P264HIP_MAX_REFS + r1c];
            uint32_t p0[4], p1[4];
            predict(ro0, mvp, u0, p0);
            predict(ro1, mv1, u1, p1);
#pragma unroll
            for (int y = 0; y < 4; y++) {
                const uint32_t both = pd->weighted ? bipred_weight4(p0[y], p1[y], wgt) : bipred_avg4(p0[y], p1[y]);
                out[y] = (u0 && u1) ? both : u0 ? p0[y] : p1[y];
            }
        }
    }
    // the next chunk's record (its entries have long arrived: they were requested in front of this chunk's windows), in flight
    // while this chunk adds its residual and stores
    if (!PB) rec = mc_entry_record(pd, g, e_next.x, (MC_CHUNK_KEY((int)((key_w_next >> (8 * (nx & 3))) & 255u)) & MCY_RESID) != 0);
    if (PB) {
        // (core/macroblock.c:525-583, core/mc.c:76-132) the first pass's rows back into block order, then the mean or the weighted sum
        uint32_t p0[4];
        if (MB) { const uint32_t pr[4] = { prev.x, prev.y, prev.z, prev.w }; quad_transpose(p0, pr, lane); }
        else {
            // the lane loaded rows (2 * right, 2 * right + 1) of the pair of blocks: its own halves stay, the others swap
            const bool right = bx & 1;
            const uint32_t g0 = lane_xor1(right ? prev.x : prev.y), g1 = lane_xor1(right ? prev.z : prev.w);
            p0[0] = right ? g0 : prev.x; p0[1] = right ? g1 : prev.z; p0[2] = right ? prev.y : g0; p0[3] = right ? prev.w : g1;
        }
        const int wgt = (int)e.w >> MCE_W_SHIFT;
#pragma unroll
        for (int y = 0; y < 4; y++) out[y] = pd->weighted ? bipred_weight4(p0[y], out[y], wgt) : bipred_avg4(p0[y], out[y]);
    }
    // ---- residual (decoder/macroblock.c:839-847) ----
    if (EXPM_RESID && (key & MCY_RESID) && __ballot(coded)) {
        const uint32_t lv[8] = { la.x, la.y, la.z, la.w, lb.x, lb.y, lb.z, lb.w };
        uint32_t col[4][2];
        unscan_cols<false>(lv, col);
        dequant_cols(col, qp);
        uint32_t px[4] = { out[0], out[1], out[2], out[3] };
        idct_add(col, px);
        if (coded) { out[0] = px[0]; out[1] = px[1]; out[2] = px[2]; out[3] = px[3]; }
    }
    // ---- store: the two lanes of a block row swap halves, so that a lane writes two rows of 8 samples (four lanes: eight
    // consecutive rows, one or two cache lines) instead of four rows of 4 ----
    if (MB) {
        // macroblock items: the four lanes of a quad are the four blocks of a block row - transposed, a lane writes one
        // whole 16-byte row and the macroblock's sixteen lanes two whole cache lines
        uint32_t t[4];
        quad_transpose(t, out, lane);
        if (valid) bstore4(rs, dsto, t[0], t[1], t[2], t[3]);
    } else {
        const bool right = bx & 1;
        const uint32_t g0 = lane_xor1(right ? out[0] : out[2]), g1 = lane_xor1(right ? out[1] : out[3]);
        const uint32_t r0a = right ? g0 : out[0], r0b = right ? out[2] : g0;       // row 4*by + 2*right: samples 0-3, 4-7
        const uint32_t r1a = right ? g1 : out[1], r1b = right ? out[3] : g1;       // the row below
        if (valid) {
            bstore2(rs, dsto, r0a, r0b);
            bstore2(rs, dsto + 16, r1a, r1b);
        }
    }
    if (!more) break;
    chunk = next; key_w = key_w_next; e = e_next;
  }
}
// ------------------------------------------------------------------------------------------
// mc_chroma_body<MB, PB> (a role of k_mc / k_mc_second): one wavefront = one chunk: 8 macroblock items of 8 lanes or 32 quadrant items of 2 lanes;
// the lane is one 4x4 chroma block (quadrant, plane)
// ------------------------------------------------------------------------------------------
// 1/8-pel bilinear (core/mc.c:303-334) of one 4x4 block from the two aligned dwords d0, d1 of five window rows; the
// block's first sample is byte s of d0; weights (dx, dy)
__device__ __forceinline__ void chroma_bilinear(uint32_t (&out)[4], const uint32_t (&d0)[5], const uint32_t (&d1)[5], uint32_t s, int dx, int dy)
{
    uint32_t a[5], b[5];                                   // bytes cx..cx+3 and cx+1..cx+4 of rows cy..cy+4
#pragma unroll
    for (int r = 0; r < 5; r++) { a[r] = alignbyte(d1[r], d0[r], s); b[r] = alignbyte(d1[r] >> (8 * s), a[r], 1); }
    // sample x of row y: (8-dx)(8-dy) A + dx (8-dy) B + (8-dx) dy C + dx dy D + 32 >> 6 as one 4-byte dot product with
    // the weights (they fit a byte: at most 64), A B adjacent in row y, C D in row y + 1
    const uint32_t wts = (uint32_t)((8 - dx) * (8 - dy)) | (uint32_t)(dx * (8 - dy)) << 8 | (uint32_t)((8 - dx) * dy) << 16 | (uint32_t)(dx * dy) << 24;
#pragma unroll
    for (int y = 0; y < 4; y++) {
        const uint32_t v0 = __builtin_amdgcn_udot4(perm(a[y + 1], a[y], 0x05040100u), wts, 32u, false);
        const uint32_t v1 = __builtin_amdgcn_udot4(perm(a[y + 1], a[y], 0x06050201u), wts, 32u, false);
        const uint32_t v2 = __builtin_amdgcn_udot4(perm(a[y + 1], a[y], 0x07060302u), wts, 32u, false);
        const uint32_t v3 = __builtin_amdgcn_udot4(perm(b[y + 1], b[y], 0x07060302u), wts, 32u, false);
        // v < 2^14: (v0 | v1 << 16) >> 6 leaves sample 0 in byte 0 and sample 1 in byte 2
        const uint32_t w01 = (v0 | (v1 << 16)) >> 6, w23 = (v2 | (v3 << 16)) >> 6;
        out[y] = perm(w23, w01, 0x06040200u);
    }
}
// the window straight from memory, coordinates clamped to the picture (slow class only)
__device__ __forceinline__ void mc_chroma_clamped(uint32_t (&out)[4], rsrc_t rs, uint32_t roff, const Geom &g, int p, int cx, int cy, int dx, int dy)
{
    const int xa = cx & ~3;
    uint32_t d0[5], d1[5], col[2], sel[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int x = xa + 4 * k, xc = clip3i(x, 0, g.cw - 4);
        col[k] = roff + g.coff + strip_mul(xc >> 3, g.cstrip) + (uint32_t)(p * 8 + (xc & 7));
        sel[k] = x < 0 ? 0x00000000u : x >= g.cw ? 0x03030303u : 0x03020100u;
    }
#pragma unroll
    for (int r = 0; r < 5; r++) {
        const uint32_t ro = (uint32_t)clip3i(cy + r, 0, g.ch - 1) * 16u;
        const uint32_t v0 = bload(rs, col[0] + ro), v1 = bload(rs, col[1] + ro);
        d0[r] = perm(v0, v0, sel[0]); d1[r] = perm(v1, v1, sel[1]);
    }
    chroma_bilinear(out, d0, d1, (uint32_t)cx & 3u, dx, dy);
}


// Chroma work items: image row = strip A (8 U, 8 V) | strip B (8 U, 8 V) + 8 bytes; a macroblock window has 9 rows, a
// quadrant window 5.
template <bool MB> struct CItem {
    static constexpr int LANES = MB ? 8 : 2, PER_WAVE = 64 / LANES, ROWS = MB ? 9 : 5;
    static constexpr int PITCH = 40, BYTES = ROWS * PITCH + 8;
    // macroblock items: a lane reads dword row * 10 + 2 plane + (block column) of its item, the block rows 40 dwords apart: the
    // eight lanes of an item cover banks {0 .. 3} and {8 .. 11} (of 32, see YItem).  Items in pairs, 100 dwords apart inside a
    // pair and 208 from pair to pair: bank = 16 (item / 2) + 4 (item % 2) + ... - the four items of a 32-lane group tile the 32
    // banks (92 dwords per item as before put items 0 and 2, 1 and 3 on the same banks).  What is left of this role's conflicts
    // comes from the windows' own column offset (0, 1, 4 or 5 dwords by the vector), which differs from item to item.
    static constexpr int PAIR = 832, SECOND = 400;
    static constexpr int WAVE_BYTES = MB ? (PER_WAVE / 2) * PAIR : PER_WAVE * BYTES;
    static_assert(!MB || (BYTES <= SECOND && SECOND + BYTES <= PAIR), "item image");
    __device__ static __forceinline__ int item_off(int i) { return MB ? (i >> 1) * PAIR + (i & 1) * SECOND : i * BYTES; }
};
// one 16-byte row piece of a chroma strip, coordinates clamped to the picture if CLAMP (a strip outside becomes the
// replicated first / last sample of each plane's 8 bytes)
template <bool CLAMP> __device__ __forceinline__ u32x4 chroma_piece_load(rsrc_t rs, uint32_t roff, const Geom &g, int strip, int y, bool need)
{
    if (!CLAMP) return wload4(rs, need ? roff + g.coff + strip_mul(strip, g.cstrip) + (uint32_t)(y * 16) : MC_OOB);
    const int sc = clip3i(strip, 0, g.mb_w - 1);             // (a chroma strip is 8 samples wide: one per macroblock column)
    return wload4(rs, need ? roff + g.coff + strip_mul(sc, g.cstrip) + (uint32_t)(clip3i(y, 0, g.ch - 1) * 16) : MC_OOB);
}
// a strip outside the picture becomes the replicated first / last sample of each plane's 8 bytes
__device__ __forceinline__ u32x4 chroma_piece_clamp(u32x4 t, const Geom &g, int strip)
{
    if (strip < 0) { const uint32_t u = perm(t.x, t.x, 0x00000000u), v = perm(t.z, t.z, 0x00000000u); t.x = t.y = u; t.z = t.w = v; }
    if (strip >= g.mb_w) { const uint32_t u = perm(t.y, t.y, 0x03030303u), v = perm(t.w, t.w, 0x03030303u); t.x = t.y = u; t.z = t.w = v; }
    return t;
}
__device__ __forceinline__ void lds_put16(uint8_t *p, u32x4 v) { *(uint2 *)p = make_uint2(v.x, v.y); *(uint2 *)(p + 8) = make_uint2(v.z, v.w); }

template <bool MB, bool PB>
__device__ __forceinline__ void mc_chroma_body(uint8_t *images, const uint32_t *ref_tab, const PicDev *__restrict__ pd, const uint32_t *__restrict__ mc, const Geom &g,
                                               const McLayout &ml, int sub, int role_wgs)
{
    typedef CItem<MB> I;
    constexpr int LIST = (MB ? ML_CM : ML_CQ) + (PB ? ML_LISTS : 0);
    const int wave = rfl((int)(threadIdx.x >> 6));
    // (chunk walk with the next chunk's key and entries requested ahead, as in mc_luma_body)
    const int stride = role_wgs * 4, n_chunks = (int)mc[LIST];
    int chunk = sub * 4 + wave;
    if (chunk >= n_chunks) return;
    const int lane = threadIdx.x & 63, p = lane & 1, li = lane & (I::LANES - 1), it = lane / I::LANES;
    const uint32_t *cls_w = mc + ml.off_cls[LIST], *list = mc + ml.off_list[LIST];
    const rsrc_t rs = make_rsrc(pd->store, pd->store_bytes);
    uint32_t key_w = cls_w[chunk >> 2];
    uint4 e = mc_entry_load<PB>(list, chunk * I::PER_WAVE + it);
    uint3 rec = make_uint3(0, 0, 0);
    if (!PB) rec = mc_entry_record(pd, g, e.x, (MC_CHUNK_KEY((int)((key_w >> (8 * (chunk & 3))) & 255u)) & MCC_RESID) != 0);
  for (;;) {
    const int next = chunk + stride;
    const bool more = next < n_chunks;
    const int nx = min(next, n_chunks - 1);                // (unconditional, as in mc_luma_body)
    const uint32_t key_w_next = cls_w[nx >> 2];
    const uint4 e_next = mc_entry_load<PB>(list, nx * I::PER_WAVE + it);
    const int key = MC_CHUNK_KEY((int)((key_w >> (8 * (chunk & 3))) & 255u));
    wave_lds_fence();                                      // the previous chunk's image has been read
    const bool valid = (e.x & MC_ITEM_MASK) != MC_ITEM_MASK;
    const int mbx = valid ? (int)((e.x >> 2) & 2047u) : 0, mby = valid ? (int)((e.x >> 13) & 1023u) : 0;
    const int q = MB ? (li >> 1) : (valid ? (int)(e.x & 3) : 0);
    const uint32_t roff = ref_tab[(e.x >> 28) | ((e.x >> 19) & 16u)];         // (as in mc_luma_body)
    // the 16-byte row this lane stores (see the end of the loop); second pass: what the first pass left there
    const uint32_t dsto = pd->dst_off + mb_chroma_off(g, mbx, mby) + (uint32_t)((q >> 1) * 64 + (lane & 3) * 16);
    u32x4 prev = { 0, 0, 0, 0 };
    if (PB) prev = bload4(rs, valid ? dsto : MC_OOB);
    const int CX = mbx * 8 + (q & 1) * 4, CY = mby * 8 + (q >> 1) * 4;      // the block's first sample
    const int cb = 16 + 4 * p + q;                           // this block's bit of coef_mask
    unsigned mask = 0; int qp = 0; uint32_t cidx = 0; bool has_res = false;      // (filled in behind the window requests, as in mc_luma_body)
    auto resid_fields = [&]() {
        if (PB) { cidx = e.w & ((1u << MCE_W_SHIFT) - 1u); mask = e.z & 0x03ffffffu; qp = (int)(e.z >> 26); }
        else {
            asm volatile("" : "+v"(rec.x), "+v"(rec.y), "+v"(rec.z));
            cidx = rec.z; mask = (e.x & MCE_Z) ? 0u : (rec.y & 0x03ffffffu); qp = (int)((rec.x >> 8) & 63u);
        }
        has_res = valid && (mask & (0x00ff0000u | P264_COEF_CHROMA_DC)) != 0;
    };
    if (PB) resid_fields();
    uint4 la = make_uint4(0, 0, 0, 0), lb = la; uint2 dcl = make_uint2(0, 0);
    // ---- prediction ----
    uint32_t out[4];
    const bool staged = MB || !(key & (MCC_CLAMP | MCC_BI));
    if (staged) {
        // the item's window goes through LDS like the luma windows: 16-byte pieces (8 U, 8 V) of consecutive rows
        const int mvx = mv_x((int)e.y), mvy = mv_y((int)e.y);
        const int cx = CX + (mvx >> 3), cy = CY + (mvy >> 3);                // the block's window
        const int wx = cx - (q & 1) * 4, wy = cy - (q >> 1) * 4;             // the macroblock's window (MB); equals (cx, cy) for q = 0
        uint8_t *img;
        int x0, y0;                                                          // image origin
        if (MB) {
            // 9 rows x 2 strips = 18 pieces over 8 lanes: piece li + 8j = strip (p / 9), row p % 9
            img = images + I::item_off(wave * I::PER_WAVE + it);
            const int sA = wx >> 3;
            x0 = sA * 8; y0 = wy;
            // (ONE scalar branch around the whole sequence, every piece requested in a row and waited for once: with the clamped /
            // plain choice made per piece the compiler waited for every piece inside its own branch - three round trips)
            auto stage = [&](auto clamp_tag) {
                constexpr bool CL = decltype(clamp_tag)::value;
                u32x4 v[3]; int dst[3];
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const int pp = min(li + 8 * j, 17), s = pp >= 9 ? 1 : 0, row = pp - 9 * s;
                    dst[j] = row * I::PITCH + s * 16;
                    // (pieces 16 and 17 only in the third round: the other lanes ask beyond the buffer - zeros, no memory request)
                    v[j] = chroma_piece_load<CL>(rs, roff, g, sA + s, wy + row, j < 2 || li < 2);
                }
#pragma unroll
                for (int j = 0; j < 3; j++) loads_land(v[j]);
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    if (CL) v[j] = chroma_piece_clamp(v[j], g, sA + (min(li + 8 * j, 17) >= 9 ? 1 : 0));
                    if (j < 2 || li < 2) lds_put16(img + dst[j], v[j]);
                }
            };
            if (key & MCC_CLAMP) stage(std::true_type{}); else stage(std::false_type{});
        } else {
            // Four lanes = two quadrants stage together: rows 0..3 of one strip of one quadrant per load (64 contiguous
            // bytes), then row 4 of their own strips.
            const int sA = cx >> 3;
            x0 = sA * 8; y0 = cy;
            const uint32_t own = roff + g.coff + strip_mul(sA, g.cstrip) + (uint32_t)(cy * 16), other = lane_xor2(own);
            const int i = lane & 3;
            uint8_t *pimg = images + (wave * I::PER_WAVE + (lane >> 2) * 2) * I::BYTES;
            u32x4 v[5];
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = wload4(rs, (((j >> 1) == (i >> 1)) ? own : other) + (uint32_t)(j & 1) * g.cstrip + (uint32_t)(i * 16));
            v[4] = wload4(rs, own + (uint32_t)(i & 1) * g.cstrip + 64u);
#pragma unroll
            for (int j = 0; j < 4; j++) lds_put16(pimg + (j >> 1) * I::BYTES + i * I::PITCH + (j & 1) * 16, v[j]);
            lds_put16(pimg + (i >> 1) * I::BYTES + 4 * I::PITCH + (i & 1) * 16, v[4]);
            img = pimg + (i >> 1) * I::BYTES;
        }
        if (key & MCC_RESID) {
            if (!PB) resid_fields();
            const int16_t *cf = pd->coefs + (size_t)cidx * 16;
            if (has_res && ((mask >> cb) & 1)) { const int16_t *c = cf + coef_slot(mask, cb) * 16; MC_LOAD_LEVELS(c, la, lb); }
            if (has_res && (mask & P264_COEF_CHROMA_DC)) dcl = gload2(cf + ((mask >> 24) & 1) * 16 + p * 4);
        }
        wave_lds_fence();
        // the lane's five rows, two aligned dwords each: sample x of plane p sits at ((x - x0) >> 3) * 16 + p * 8 + (x & 7)
        const int xa = cx & ~3;
        const uint8_t *b = img + (cy - y0) * I::PITCH + p * 8;
        const int o0 = ((xa - x0) >> 3) * 16 + (xa & 7), o1 = ((xa + 4 - x0) >> 3) * 16 + ((xa + 4) & 7);
        uint32_t d0[5], d1[5];
#pragma unroll
        for (int r = 0; r < 5; r++) { d0[r] = *(const uint32_t *)(b + r * I::PITCH + o0); d1[r] = *(const uint32_t *)(b + r * I::PITCH + o1); }
        chroma_bilinear(out, d0, d1, (uint32_t)cx & 3u, mvx & 7, mvy & 7);
    } else {
        // quadrant items outside the picture or with different vectors inside (sub-8x8 partitions: every 2x2 chroma piece
        // follows its own vector): clamped windows straight from memory, one pass per piece, wave-uniformly skipped when
        // nobody needs it
        if (key & MCC_RESID) {
            if (!PB) resid_fields();
            const int16_t *cf = pd->coefs + (size_t)cidx * 16;
            if (has_res && ((mask >> cb) & 1)) { const int16_t *c = cf + coef_slot(mask, cb) * 16; MC_LOAD_LEVELS(c, la, lb); }
            if (has_res && (mask & P264_COEF_CHROMA_DC)) dcl = gload2(cf + ((mask >> 24) & 1) * 16 + p * 4);
        }
        const int b0 = (q >> 1) * 8 + (q & 1) * 2;
        const int mbi = mby * g.mb_w + mbx;
        // the lane's 4x4 chroma block from reference frame `ro`, the vectors of the quadrant's four luma blocks at `mvs`
        auto predict = [&](uint32_t ro, const int *mvs, bool use, uint32_t (&o4)[4]) {
            const uint2 va = gload2(mvs + mbi * 16 + b0), vb = gload2(mvs + mbi * 16 + b0 + 4);
            const bool uniform = va.x == va.y && va.x == vb.x && va.x == vb.y;
            const int v4[4] = { (int)va.x, (int)va.y, (int)vb.x, (int)vb.y };
            o4[0] = o4[1] = o4[2] = o4[3] = 0;
#pragma unroll 1
            for (int sb = 0; sb < 4; sb++) {
                if (sb > 0 && __ballot(use && !uniform) == 0) break;
                const int mv = sb == 0 ? v4[0] : sb == 1 ? v4[1] : sb == 2 ? v4[2] : v4[3];
                uint32_t o[4];
                mc_chroma_clamped(o, rs, ro, g, p, CX + (mv_x(mv) >> 3), CY + (mv_y(mv) >> 3), mv_x(mv) & 7, mv_y(mv) & 7);
                if (uniform) { if (sb == 0) { o4[0] = o[0]; o4[1] = o[1]; o4[2] = o[2]; o4[3] = o[3]; } }
                else {
                    const uint32_t m = (sb & 1) ? 0xffff0000u : 0x0000ffffu;
                    const int r0 = (sb >> 1) * 2;
                    o4[0] = r0 == 0 ? (o4[0] & ~m) | (o[0] & m) : o4[0]; o4[1] = r0 == 0 ? (o4[1] & ~m) | (o[1] & m) : o4[1];
                    o4[2] = r0 == 2 ? (o4[2] & ~m) | (o[2] & m) : o4[2]; o4[3] = r0 == 2 ? (o4[3] & ~m) | (o[3] & m) : o4[3];
                }
            }
        };
        if (!(key & MCC_BI)) predict(roff, (valid && (e.x & MCE_LIST1)) ? pd->mv_l1 : pd->mv, valid, out);      // (padding lanes: list 0, mv_l1 is null outside B pictures)
        else {
            // B picture, two lists: as in mc_luma_body
            const int r0 = (int)glob(pd->ref_idx)[mbi * 4 + q], r1 = (int)glob(pd->ref_idx_l1)[mbi * 4 + q];
            const bool u0 = valid && r0 >= 0, u1 = valid && r1 >= 0;
            const int r0c = min(max(r0, 0), pd->n_ref - 1), r1c = min(max(r1, 0), pd->n_ref_l1 - 1);
            const uint32_t ro0 = glob(pd->ref_off)[r0c], ro1 = glob(pd->ref_off_l1)[r1c];
            int wgt = 32;
            if (pd->weighted) wgt = (int)glob(pd->bipred_w)[r0c * P264HIP_MAX_REFS + r1c];
            uint32_t p0[4], p1[4];
            predict(ro0, pd->mv, u0, p0);
            predict(ro1, pd->mv_l1, u1, p1);
#pragma unroll
            for (int y = 0; y < 4; y++) {
                const uint32_t both = pd->weighted ? bipred_weight4(p0[y], p1[y], wgt) : bipred_avg4(p0[y], p1[y]);
                out[y] = (u0 && u1) ? both : u0 ? p0[y] : p1[y];
            }
        }
    }
    // (the next chunk's record, as in mc_luma_body)
    if (!PB) rec = mc_entry_record(pd, g, e_next.x, (MC_CHUNK_KEY((int)((key_w_next >> (8 * (nx & 3))) & 255u)) & MCC_RESID) != 0);
    if (PB) {
        // the first pass's rows (dwords U left, U right, V left, V right) back into block order; then the mean or the weighted
        // sum, or - quadrants that were finished in the first pass - the samples as they are
        const uint32_t pr[4] = { prev.x, prev.z, prev.y, prev.w };
        uint32_t p0[4];
        quad_transpose(p0, pr, lane);
        const int wgt = (int)e.w >> MCE_W_SHIFT;
        const bool keep = (e.x & MCE_KEEP) != 0;
#pragma unroll
        for (int y = 0; y < 4; y++) {
            const uint32_t both = pd->weighted ? bipred_weight4(p0[y], out[y], wgt) : bipred_avg4(p0[y], out[y]);
            out[y] = keep ? p0[y] : both;
        }
    }
    // ---- residual (decoder/macroblock.c:851-890): chroma DC through the 2x2 transform, AC, inverse transform ----
    if (EXPM_RESID && (key & MCC_RESID) && __ballot(has_res)) {
        const int qpc = chroma_qp(clip3i(qp + pd->chroma_qp_offset, 0, 51));
        const uint32_t lv[8] = { la.x, la.y, la.z, la.w, lb.x, lb.y, lb.z, lb.w };
        uint32_t col[4][2];
        unscan_cols<true>(lv, col);
        dequant_cols(col, qpc);
        // DC of block q: idct2x2dc (core/dct.c:55-68, int16 stores), then p264_mb_dequant_2x2_dc (core/quant.c:138-159):
        // for every QP its value is (f * (scale << qp/6)) >> 1 - exact for the shifts left, the truncating shift below QP 6
        const int d0 = (int)(int16_t)(dcl.x & 0xffff), d1 = (int)dcl.x >> 16, d2 = (int)(int16_t)(dcl.y & 0xffff), d3 = (int)dcl.y >> 16;
        const int t0 = d0 + d1, t1 = d0 - d1, t2 = d2 + d3, t3 = d2 - d3;
        int f = pick_addsub(t0, t1, t2, t3, q & 1, q & 2);                    // {t0+t2, t1+t3, t0-t2, t1-t3}[q]
        f = (int)(int16_t)f;
        const int per = (qpc * 43) >> 8, rem = qpc - per * 6;
        const int dc = (f * (int)(dq_s(0, rem) << per)) >> 1;
        col[0][0] = (col[0][0] & 0xffff0000u) | ((uint32_t)dc & 0xffffu);
        uint32_t px[4] = { out[0], out[1], out[2], out[3] };
        idct_add(col, px);
        if (has_res) { out[0] = px[0]; out[1] = px[1]; out[2] = px[2]; out[3] = px[3]; }
    }
    {
        // (Quadrant items: the four entries of a split macroblock are consecutive, so its eight lanes sit exactly like the
        // eight lanes of a macroblock item.)  The four lanes of a quad hold the four dwords of the same four rows (U left, V left, U right, V right block):
        // transpose among them, so that a lane writes ONE whole 16-byte row and a macroblock's eight lanes one cache line
        // (four 4-byte stores per lane kept the address unit busier than anything else in this kernel).
        uint32_t t[4];
        quad_transpose(t, out, lane);
        if (valid) bstore4(rs, dsto, t[0], t[2], t[1], t[3]);      // lane-in-quad = plane + 2 * (block column): dwords U0 U1 V0 V1
    }
    if (!more) break;
    chunk = next; key_w = key_w_next; e = e_next;
  }
}
// ------------------------------------------------------------------------------------------
// k_mc: ONE launch for the four kinds of work.  A picture gets `wgs_per_pic` consecutive (logical) workgroups; they split
// into the four roles - luma macroblock items, luma quadrant items, chroma macroblock items, chroma quadrant items - in
// proportion to the work the picture's lists hold (chunks in use x a cost weight per chunk, measured), decided on the device
// from the counts k_mc_sort left.  All four roles of a picture run at the same time on the same XCD and walk the picture's
// bands in step: a reference line fetched for a macroblock item is in L2 when the quadrant item next to it and the chroma
// windows ask for it (as four launches every kernel pulled the whole reference through HBM again: measured 7.0 GB of reads
// against 3.1 GB of reference samples), and the bandwidth-bound chroma work shares its CU with the issue-bound luma work.
// ------------------------------------------------------------------------------------------
#ifndef MC_COST_YM
#define MC_COST_YM 7u               // relative cost of one chunk (wavefront pass) per role, from the round-2 profiles (swept again in round 5: scratch/r5_mccost.sh)
#define MC_COST_YQ 10u
#define MC_COST_CM 5u
#define MC_COST_CQ 9u
#endif
template <bool PB>
__device__ __forceinline__ void mc_roles(uint8_t *images, uint32_t *ref_tab, const PicDev *__restrict__ pics, const uint32_t *__restrict__ mc_all, const Geom &g, const McLayout &ml,
                                         int wgs_per_pic, int n_wgs, uint32_t inv_wgs)
{
    constexpr int L0 = PB ? ML_LISTS : 0;
    const int logical = xcd_logical_block();
    if (logical >= n_wgs) return;
    int pic = (int)__umulhi((unsigned)logical, inv_wgs);
    if (logical - pic * wgs_per_pic >= wgs_per_pic) pic++;
    const int s = logical - pic * wgs_per_pic;
    const PicDev *pd = pics + pic;
    const uint32_t *mc = mc_all + (size_t)pic * ml.words;  // (addresses from kernel arguments only: the first loads depend on nothing)
    // the picture's reference frames by (list, index) for the workgroup's wavefronts: entries past a list's end repeat entry 0
    // (p264hip_reconstruct), list 1 exists for B pictures only
    static_assert(MCE_LIST1 == 1u << 23 && P264HIP_MAX_REFS == 16, "table index of mc_luma_body / mc_chroma_body");
    if (threadIdx.x < 2 * P264HIP_MAX_REFS) {
        const int k = threadIdx.x & (P264HIP_MAX_REFS - 1);
        ref_tab[threadIdx.x] = threadIdx.x < P264HIP_MAX_REFS ? pd->ref_off[k] : (pd->slice_type == P264_SLICE_B ? pd->ref_off_l1[k] : pd->ref_off[0]);
    }
    __syncthreads();
    // role split (scalar): every non-empty list gets one workgroup, the rest go by cost
    const uint32_t n0 = mc[L0 + ML_YM], n1 = mc[L0 + ML_YQ], n2 = mc[L0 + ML_CM], n3 = mc[L0 + ML_CQ];
    const uint32_t t0 = n0 * MC_COST_YM, t1 = n1 * MC_COST_YQ, t2 = n2 * MC_COST_CM, t3 = n3 * MC_COST_CQ, tt = t0 + t1 + t2 + t3;
    if (tt == 0) return;
    const uint32_t nz = (n0 != 0) + (n1 != 0) + (n2 != 0) + (n3 != 0), spare = (uint32_t)wgs_per_pic - nz;       // wgs_per_pic >= 4
    const float inv = (float)spare / (float)tt;
    const int w1 = (n1 != 0) + (int)((float)t1 * inv), w2 = (n2 != 0) + (int)((float)t2 * inv), w3 = (n3 != 0) + (int)((float)t3 * inv);
    const int w0 = wgs_per_pic - w1 - w2 - w3;             // (luma macroblock items take the rounding remainder)
    if (EXPM_ONLY == 1 && s >= w0 + w1) return;                                 // (timing switches)
    if (EXPM_ONLY == 2 && s < w0 + w1) return;
    if (EXPM_ONLY == 3 && !(s < w0 || (s >= w0 + w1 && s < w0 + w1 + w2))) return;   // macroblock items only
    if (s < w0) mc_luma_body<true, PB>(images, ref_tab, pd, mc, g, ml, s, w0);
    else if (s < w0 + w1) mc_luma_body<false, PB>(images, ref_tab, pd, mc, g, ml, s - w0, w1);
    else if (s < w0 + w1 + w2) mc_chroma_body<true, PB>(images, ref_tab, pd, mc, g, ml, s - w0 - w1, w2);
    else mc_chroma_body<false, PB>(images, ref_tab, pd, mc, g, ml, s - w0 - w1 - w2, w3);
}
#ifndef MC_IMAGE_BYTES
#define MC_IMAGE_BYTES (YItem<false>::LEAD + 4 * YItem<false>::WAVE_BYTES)         // the largest of the four roles' images
#endif
static_assert(EXPM_ONLY == 3 || (MC_IMAGE_BYTES >= YItem<true>::LEAD + 4 * YItem<true>::WAVE_BYTES && MC_IMAGE_BYTES >= 4 * CItem<true>::WAVE_BYTES && MC_IMAGE_BYTES >= 4 * CItem<false>::WAVE_BYTES), "image space");
static_assert(MC_IMAGE_BYTES >= YItem<true>::LEAD + 4 * YItem<true>::WAVE_BYTES && MC_IMAGE_BYTES >= 4 * CItem<true>::WAVE_BYTES, "image space of the macroblock roles");
#ifndef MC_WAVES_PER_EU
#define MC_WAVES_PER_EU 4
#endif
__global__ __launch_bounds__(256, MC_WAVES_PER_EU)
void k_mc(const PicDev *__restrict__ pics, const uint32_t *__restrict__ mc_all, Geom g, McLayout ml, int wgs_per_pic, int n_wgs, uint32_t inv_wgs)
{
    __shared__ __attribute__((aligned(16))) uint8_t images[MC_IMAGE_BYTES];
    __shared__ uint32_t ref_tab[2 * P264HIP_MAX_REFS];
    mc_roles<false>(images, ref_tab, pics, mc_all, g, ml, wgs_per_pic, n_wgs, inv_wgs);
}
// B pictures: the second pass (list-1 predictions of the blocks that use both lists), behind k_mc
__global__ __launch_bounds__(256, 4)
void k_mc_second(const PicDev *__restrict__ pics, const uint32_t *__restrict__ mc_all, Geom g, McLayout ml, int wgs_per_pic, int n_wgs, uint32_t inv_wgs)
{
    __shared__ __attribute__((aligned(16))) uint8_t images[MC_IMAGE_BYTES];
    __shared__ uint32_t ref_tab[2 * P264HIP_MAX_REFS];
    mc_roles<true>(images, ref_tab, pics, mc_all, g, ml, wgs_per_pic, n_wgs, inv_wgs);
}
