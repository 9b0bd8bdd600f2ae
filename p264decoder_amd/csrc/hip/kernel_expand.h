// kernel_expand.h - the compact link format of a picture's arrays (include/p264hip.h: p264hip_compact_hdr_t) expanded into the
// slot layout the reconstruction kernels read.  One workgroup of 1024 threads per picture; a launch takes every picture that
// was uploaded compact since the last launch (p264hip_upload_compact queues, p264hip_reconstruct expands).  The host reference
// of this kernel is p264hip_expand_compact (csrc/host/compact.c).  The host has checked the header only
// (p264hip_compact_header_ok: the sections lie inside the block, in order, large enough for the header's counts); every place
// derived from the block's bits is clamped to its section here, so that a block whose bits do not add up to its header gives a
// wrong picture, never an access outside the block or the slot.
//
// Bound by memory: ~0.5 MB read and 1.4 MB written per 1080p picture.  Records and reference indices are copied as they are;
// a macroblock's vectors and Intra4x4 modes sit behind those of all macroblocks before it, so their places come out of a
// prefix sum over the shape / flag bits (list 0, then list 1 of a B picture through the same table) (kept in LDS: one entry per macroblock, P264HIP_COMPACT_MAX_MB of them), then one thread per
// macroblock writes its 64 + 16 bytes (neighbouring threads, neighbouring macroblocks: whole cache lines); the coded levels the
// same way over the flag bits, one thread per run of consecutive blocks.
#pragma once
#include "device_common.h"

struct ExpandJob { const uint8_t *src; uint8_t *dst; uint32_t off_mv, off_ref, off_i4, off_coef, off_mv_l1, off_ref_l1, off_weights, pad; };
#define EXPAND_THREADS 1024

// exclusive prefix sum of one value per thread over the workgroup (v in, the sum of the threads before this one out)
__device__ __forceinline__ uint32_t wg_exclusive_scan(uint32_t v, uint32_t *wave_sums)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(inc, d); if (lane >= d) inc += t; }
    __syncthreads();                                        // (the sums of the scan before this one have been read)
    if (lane == 63) wave_sums[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; w++) base += wave_sums[w];
    return base + inc - v;
}
__device__ __forceinline__ uint32_t shape_words(uint32_t sh) { return sh == 1 ? 1u : sh == 2 ? 4u : sh == 3 ? 16u : 0u; }

__global__ __launch_bounds__(EXPAND_THREADS)
void k_expand_compact(const ExpandJob *__restrict__ jobs)
{
    __shared__ uint32_t voff[P264HIP_COMPACT_MAX_MB];      // macroblock -> first vector (dword index) | shape << 30
    __shared__ uint16_t ioff[P264HIP_COMPACT_MAX_MB];      // macroblock -> its place among the Intra4x4 entries | has one << 15
    __shared__ uint32_t wave_sums[EXPAND_THREADS / 64];
    const ExpandJob job = jobs[blockIdx.x];
    const uint8_t *src = job.src;
    const p264hip_compact_hdr_t h = *(const p264hip_compact_hdr_t *)src;
    const int tid = threadIdx.x, n = (int)h.n_mb;
    // ---- records: as they are (16-byte pieces; every section starts on a 16-byte boundary on both sides) ----
    for (int k = tid; k < n; k += EXPAND_THREADS) gstore4(job.dst + (size_t)k * 16, gload4(src + h.off_rec + (size_t)k * 16));
    // thread t owns macroblocks [t * M, t * M + M) for the prefix sums
    const int M = (n + EXPAND_THREADS - 1) / EXPAND_THREADS, lo = min(tid * M, n), hi = min(lo + M, n);
    const uint32_t i4_last = max((h.off_lvflag - h.off_i4) / 16u, 1u) - 1u;
    for (uint32_t l = 0; l < h.n_lists; l++) {
        const uint32_t off_ref = l ? h.list[1].off_ref : h.list[0].off_ref, off_shape = l ? h.list[1].off_shape : h.list[0].off_shape;
        const uint32_t off_vec = l ? h.list[1].off_vec : h.list[0].off_vec, vec_end = l ? (h.n_lists > 1 ? h.off_i4flag : 0u) : (h.n_lists > 1 ? h.list[1].off_ref : h.off_i4flag);
        uint8_t *d_ref = job.dst + (l ? job.off_ref_l1 : job.off_ref), *d_mv = job.dst + (l ? job.off_mv_l1 : job.off_mv);
        // reference indices: as they are (the last piece may reach past n * 4 bytes: the slot's sections are 256-byte aligned)
        for (int k = tid; k < (n + 3) / 4; k += EXPAND_THREADS) gstore4(d_ref + (size_t)k * 16, gload4(src + off_ref + (size_t)k * 16));
        // ---- places of the vectors (and, with list 0, of the Intra4x4 entries) ----
        const uint8_t *shape = src + off_shape, *i4flag = src + h.off_i4flag;
        uint32_t cnt = 0;                                   // vectors | Intra4x4 entries << 18 (8192 x 16 vectors < 2^18)
        for (int m = lo; m < hi; m++) {
            cnt += shape_words((glob(shape)[m >> 2] >> (2 * (m & 3))) & 3u);
            if (l == 0) cnt += ((glob(i4flag)[m >> 3] >> (m & 7)) & 1u) << 18;
        }
        uint32_t at = wg_exclusive_scan(cnt, wave_sums);
        __syncthreads();                                    // (the table's previous readers - list 0's expansion - are through)
        for (int m = lo; m < hi; m++) {
            const uint32_t sh = (glob(shape)[m >> 2] >> (2 * (m & 3))) & 3u;
            voff[m] = (at & 0x3ffffu) | sh << 30;
            at += shape_words(sh);
            if (l == 0) { const uint32_t f = (glob(i4flag)[m >> 3] >> (m & 7)) & 1u; ioff[m] = (uint16_t)(((at >> 18) & 0x3fffu) | f << 15); at += f << 18; }
        }
        __syncthreads();
        // ---- one thread per macroblock: sixteen vectors (and, with list 0, sixteen modes) ----
        const uint32_t *vec = (const uint32_t *)(src + off_vec);
        const uint32_t vec_words = max((vec_end - off_vec) / 4u, 16u);           // (the header check: the next section starts behind this one)
        for (int m = tid; m < n; m += EXPAND_THREADS) {
            const uint32_t e = voff[m], sh = e >> 30, o = min(e & 0x3ffffu, vec_words - max(shape_words(sh), 1u));
            uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0, r2 = r0, r3 = r0;      // rows of the macroblock's 4x4 grid of vectors
            if (sh == 1) { const uint32_t v = gload1(vec + o); r0 = r1 = r2 = r3 = make_uint4(v, v, v, v); }
            else if (sh == 2) {
                const uint32_t q0 = gload1(vec + o), q1 = gload1(vec + o + 1), q2 = gload1(vec + o + 2), q3 = gload1(vec + o + 3);
                r0 = r1 = make_uint4(q0, q0, q1, q1); r2 = r3 = make_uint4(q2, q2, q3, q3);
            } else if (sh == 3) {
                uint32_t t[16];                                                   // (a macroblock's vectors start on any dword)
#pragma unroll
                for (int k = 0; k < 16; k++) t[k] = gload1(vec + o + k);
                r0 = make_uint4(t[0], t[1], t[2], t[3]); r1 = make_uint4(t[4], t[5], t[6], t[7]);
                r2 = make_uint4(t[8], t[9], t[10], t[11]); r3 = make_uint4(t[12], t[13], t[14], t[15]);
            }
            uint8_t *mv = d_mv + (size_t)m * 64;
            gstore4(mv, r0); gstore4(mv + 16, r1); gstore4(mv + 32, r2); gstore4(mv + 48, r3);
            if (l == 0) {
                uint4 modes = make_uint4(0x02020202u, 0x02020202u, 0x02020202u, 0x02020202u);
                const uint32_t io = ioff[m];
                if (io & 0x8000u) modes = gload4(src + h.off_i4 + (size_t)min(io & 0x7fffu, i4_last) * 16);
                gstore4(job.dst + job.off_i4 + (size_t)m * 16, modes);
            }
        }
    }
    if (h.n_lists > 1 && tid < 32) gstore4(job.dst + job.off_weights + (size_t)tid * 16, gload4(src + h.off_weights + (size_t)tid * 16));   // bipred_weight[]: 512 bytes
    // ---- coded levels: thread t owns blocks [t * B, t * B + B); a block is sixteen int8 (flag bit set) or sixteen int16 ----
    const int nb = (int)h.n_coef_blocks, B = (nb + EXPAND_THREADS - 1) / EXPAND_THREADS, b0 = min(tid * B, nb), b1 = min(b0 + B, nb);
    const uint8_t *flag = src + h.off_lvflag, *lv = src + h.off_levels;
    uint32_t bytes = 0;
    for (int b = b0; b < b1; b++) bytes += ((glob(flag)[b >> 3] >> (b & 7)) & 1) ? 16u : 32u;
    uint32_t from = wg_exclusive_scan(bytes, wave_sums);
    const uint32_t lv_last = h.level_bytes & ~15u;           // (the header check: 32 readable bytes behind the levels)
    for (int b = b0; b < b1; b++) {
        const bool narrow = (glob(flag)[b >> 3] >> (b & 7)) & 1;
        uint8_t *o = job.dst + job.off_coef + (size_t)b * 32;
        from = min(from, lv_last);
        const uint4 a = gload4(lv + from);                 // (16-byte aligned: every block is a multiple of 16 bytes)
        if (narrow) {
            const uint32_t w[4] = { a.x, a.y, a.z, a.w };
            uint32_t e[8];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                // two int8 -> two int16 in one word
                e[2 * k]     = (uint32_t)(((int32_t)(w[k] << 24) >> 24) & 0xffff) | (uint32_t)((int32_t)(w[k] << 16) >> 24) << 16;
                e[2 * k + 1] = (uint32_t)(((int32_t)(w[k] << 8) >> 24) & 0xffff) | (uint32_t)((int32_t)w[k] >> 24) << 16;
            }
            gstore4(o, make_uint4(e[0], e[1], e[2], e[3])); gstore4(o + 16, make_uint4(e[4], e[5], e[6], e[7]));
            from += 16;
        } else {
            gstore4(o, a); gstore4(o + 16, gload4(lv + from + 16));
            from += 32;
        }
    }
}
