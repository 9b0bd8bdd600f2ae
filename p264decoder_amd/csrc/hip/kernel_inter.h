// kernel_inter.h - K1: inter prediction + residual for every non-intra macroblock of a batch.
//
// Replaces p264_mb_mc / p264_mb_mc_0xywh (core/macroblock.c:506-524,633-676), mc_luma /
// pixel_avg / mc_copy (core/mc.c:58-74,160-171,237-266), the half-pel plane generator
// p264_frame_filter (core/mc.c:172-235,409-451 - computed on the fly here, never stored),
// motion_compensation_chroma (core/mc.c:303-334), p264_macroblock_decode_skip
// (decoder/macroblock.c:895-934) and the inter half of p264_macroblock_decode
// (decoder/macroblock.c:832-890: unscan, dequant_4x4, add4x4_idct, chroma DC).
//
// Shape: one 64-lane wavefront per macroblock, four macroblocks (256 threads) per workgroup,
// all inter MBs of all pictures of the batch in one launch (they are independent).
// Per 8x8 quadrant the 13x13 reference window is staged in LDS with ONE aligned dword load
// per lane, then every lane produces one sample.  Output is collected in LDS and leaves as
// one coalesced dword store per lane.  Border padding (core/frame.c:183-222) is replaced by
// coordinate clamping, which is equivalent inside the reference's pads (SURVEY A-Q9).
#pragma once
#include "device_common.h"

struct InterLds {                 // per wavefront
    uint32_t win[13 * 4];         // 13 rows x 16 bytes of reference luma
    int16_t  coef[4 * 16];        // dequantised coefficients of the four 4x4 blocks of a quadrant / plane
    uint32_t outY[64];            // 16x16 luma, raster
    uint32_t outC[32];            // two 8x8 chroma planes
};

// -- sample fetch policies ------------------------------------------------------------------
struct LdsWin {                   // window staged in LDS; (x,y) relative to the window origin
    const uint8_t *w;
    __device__ __forceinline__ int operator()(int x, int y) const { return w[y * 16 + x]; }
};
struct ClampedPlane {             // direct global reads with clamped coordinates
    const uint8_t *p; int w, h;
    __device__ __forceinline__ int operator()(int x, int y) const
    { return p[clip3i(y, 0, h - 1) * w + clip3i(x, 0, w - 1)]; }
};

template <class F> __device__ __forceinline__ int tap_h(const F &f, int x, int y)
{   // core/mc.c:53-56
    return f(x-2, y) - 5*f(x-1, y) + 20*(f(x, y) + f(x+1, y)) - 5*f(x+2, y) + f(x+3, y);
}
template <class F> __device__ __forceinline__ int tap_v(const F &f, int x, int y)
{   // core/mc.c:49-52
    return f(x, y-2) - 5*f(x, y-1) + 20*(f(x, y) + f(x, y+1)) - 5*f(x, y+2) + f(x, y+3);
}

// Sample at half-pel coordinate (2*x + hx, 2*y + hy): the value the reference would read from
// plane (hx&1)+2*(hy&1) of {integer, H, V, HV} (core/mc.c:180,194,213-223).
template <class F> __device__ __forceinline__ int half_sample(const F &f, int x, int y, int hx, int hy)
{
    x += hx >> 1; y += hy >> 1;
    int which = (hx & 1) | ((hy & 1) << 1);
    if (which == 0) return f(x, y);
    if (which == 1) return clip255((tap_h(f, x, y) + 16) >> 5);
    if (which == 2) return clip255((tap_v(f, x, y) + 16) >> 5);
    int t = tap_h(f, x, y-2) - 5*tap_h(f, x, y-1) + 20*(tap_h(f, x, y) + tap_h(f, x, y+1)) - 5*tap_h(f, x, y+2) + tap_h(f, x, y+3);
    return clip255((t + 512) >> 10);
}

// Quarter-pel luma sample (core/mc.c:244-265): one half-pel-grid sample, or the rounded mean of two.
template <class F> __device__ __forceinline__ int qpel_sample(const F &f, int x, int y, int fx, int fy)
{
    int corr = (fx & 1) && (fy & 1) && ((fx & 2) ^ (fy & 2));
    int a = half_sample(f, x, y, fx >> 1, (fy + 1 - corr) >> 1);
    if ((fx | fy) & 1) {
        int b = half_sample(f, x, y, (fx + 1) >> 1, (fy + corr) >> 1);
        a = (a + b + 1) >> 1;
    }
    return a;
}

__device__ __forceinline__ int mv_x(int packed) { return (int)(int16_t)(packed & 0xffff); }
__device__ __forceinline__ int mv_y(int packed) { return packed >> 16; }

// Add the residual of one plane's four 4x4 blocks (an 8x8 area) held dequantised in L.coef.
// present: bit j set -> block j contributes.  (px,py): this lane's sample inside the 8x8.
__device__ __forceinline__ int add_residual8x8(const int16_t *coef, unsigned present, int px, int py, int pred)
{
    int j = ((py >> 2) << 1) | (px >> 2);
    if (!((present >> j) & 1)) return pred;
    return clip255(pred + idct4x4_sample(coef + j * 16, px & 3, py & 3));
}

__global__ __launch_bounds__(256)
void k_inter(const PicDev *__restrict__ pics, Geom g, int blocks_per_pic, int n_blocks)
{
    __shared__ InterLds lds[4];
    // XCD-aware remap: the dispatcher deals workgroups round-robin over the 8 XCDs; give every XCD
    // one contiguous eighth of the batch so that overlapping reference windows share an L2.
    int per_xcd = gridDim.x >> 3;
    int logical = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (logical >= n_blocks) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pic = logical / blocks_per_pic;
    const int mbi = rfl((logical - pic * blocks_per_pic) * 4 + wave);
    if (mbi >= g.n_mb) return;
    const PicDev *pd = pics + pic;
    const p264hip_mb_t m = pd->mb[mbi];
    if (P264_MB_IS_INTRA(m.mb_type)) return;

    InterLds &L = lds[wave];
    const int mbx = mbi % g.mb_w, mby = mbi / g.mb_w;
    const int X0 = mbx * 16, Y0 = mby * 16;
    const int mvreg = lane < 16 ? pd->mv[mbi * 16 + lane] : 0;
    const int refs4 = *(const int *)(pd->ref_idx + mbi * 4);
    const int px = lane & 7, py = lane >> 3;

    // ---------------- luma: four 8x8 quadrants ----------------
    for (int q = 0; q < 4; q++) {
        const int qx = (q & 1) * 8, qy = (q >> 1) * 8;
        const int b0 = (qy >> 2) * 4 + (qx >> 2);
        const int mv0 = __builtin_amdgcn_readlane(mvreg, b0), mv1 = __builtin_amdgcn_readlane(mvreg, b0 + 1);
        const int mv2 = __builtin_amdgcn_readlane(mvreg, b0 + 4), mv3 = __builtin_amdgcn_readlane(mvreg, b0 + 5);
        int ri = (int)(int8_t)(refs4 >> (8 * q));
        if (ri < 0 || ri >= pd->n_ref) ri = 0;
        const uint8_t *refY = pd->ref[ri];
        const int mvx = mv_x(mv0), mvy = mv_y(mv0);
        const int ix = X0 + qx + (mvx >> 2), iy = Y0 + qy + (mvy >> 2);
        const int wx0 = ix - 2;
        int val;
        if (mv0 == mv1 && mv0 == mv2 && mv0 == mv3 && wx0 >= 0 && wx0 + 12 < g.w) {
            if (lane < 52) {
                int r = lane >> 2, d = lane & 3;
                int yy = clip3i(iy - 2 + r, 0, g.h - 1);
                L.win[lane] = *(const uint32_t *)(refY + (size_t)yy * g.w + (wx0 & ~3) + d * 4);
            }
            wave_lds_fence();
            LdsWin f = { (const uint8_t *)L.win };
            val = qpel_sample(f, px + 2 + (wx0 & 3), py + 2, mvx & 3, mvy & 3);
            wave_lds_fence();
        } else {
            // sub-8x8 partitions with differing vectors, or a window that crosses the left/right
            // picture edge: every lane samples the clamped plane directly with its own vector
            int mvl = __shfl(mvreg, b0 + (py >> 2) * 4 + (px >> 2));
            int lx = mv_x(mvl), ly = mv_y(mvl);
            ClampedPlane f = { refY, g.w, g.h };
            val = qpel_sample(f, X0 + qx + px + (lx >> 2), Y0 + qy + py + (ly >> 2), lx & 3, ly & 3);
        }
        ((uint8_t *)L.outY)[(qy + py) * 16 + qx + px] = (uint8_t)val;
    }

    // ---------------- chroma: 8x8 per plane, one sample per lane (core/mc.c:303-334) ----------------
    {
        int mvl = __shfl(mvreg, (py >> 1) * 4 + (px >> 1));
        int lx = mv_x(mvl), ly = mv_y(mvl);
        int ri = (int)(int8_t)(refs4 >> (8 * (((py >> 2) << 1) | (px >> 2))));
        if (ri < 0 || ri >= pd->n_ref) ri = 0;
        const uint8_t *rf = pd->ref[ri];
        int dx = lx & 7, dy = ly & 7;
        int cA = (8 - dx) * (8 - dy), cB = dx * (8 - dy), cC = (8 - dx) * dy, cD = dx * dy;
        int sx = X0 / 2 + px + (lx >> 3), sy = Y0 / 2 + py + (ly >> 3);
        int x0 = clip3i(sx, 0, g.cw - 1), x1 = clip3i(sx + 1, 0, g.cw - 1);
        int y0 = clip3i(sy, 0, g.ch - 1) * g.cw, y1 = clip3i(sy + 1, 0, g.ch - 1) * g.cw;
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const uint8_t *c = rf + (p ? g.off_v : g.off_u);
            int v = (cA * c[y0 + x0] + cB * c[y0 + x1] + cC * c[y1 + x0] + cD * c[y1 + x1] + 32) >> 6;
            ((uint8_t *)L.outC)[p * 64 + py * 8 + px] = (uint8_t)v;
        }
    }

    // ---------------- residual (decoder/macroblock.c:832-890) ----------------
    const unsigned mask = m.coef_mask;
    if (mask) {
        const int16_t *cf = pd->coefs + (size_t)m.coef_index * 16;
        const int j = lane >> 4, k = lane & 15, pos = c_zigzag[k];
        if (mask & 0xffff) {
            for (int q = 0; q < 4; q++) {
                unsigned present = (mask >> (4 * q)) & 15;
                if (!present) continue;
                int blk = 4 * q + j;
                int c = (present >> j) & 1 ? cf[coef_slot(mask, blk) * 16 + k] : 0;
                L.coef[j * 16 + pos] = (int16_t)dequant_coef(c, pos, m.qp);
                wave_lds_fence();
                const int qx = (q & 1) * 8, qy = (q >> 1) * 8;
                uint8_t *o = (uint8_t *)L.outY + (qy + py) * 16 + qx + px;
                *o = (uint8_t)add_residual8x8(L.coef, present, px, py, *o);
                wave_lds_fence();
            }
        }
        if (m.cbp >> 4) {
            const int qpc = c_chroma_qp[clip3i(m.qp + pd->chroma_qp_offset, 0, 51)];
            const int16_t *dcp = cf + ((mask >> 24) & 1) * 16;
            for (int p = 0; p < 2; p++) {
                int v;
                if (k == 0) {
                    // chroma DC of block j: idct2x2dc (core/dct.c:55-68) then truncating dequant (core/quant.c:138-159)
                    int d0 = 0, d1 = 0, d2 = 0, d3 = 0;
                    if (mask & P264_COEF_CHROMA_DC) { d0 = dcp[p*4]; d1 = dcp[p*4+1]; d2 = dcp[p*4+2]; d3 = dcp[p*4+3]; }
                    int t0 = d0 + d1, t1 = d0 - d1, t2 = d2 + d3, t3 = d2 - d3;
                    int f = j == 0 ? t0 + t2 : j == 1 ? t1 + t3 : j == 2 ? t0 - t2 : t1 - t3;
                    f = (int)(int16_t)f;
                    int qbits = qpc / 6 - 5, mf = c_dqmf[qpc % 6][0];
                    v = qbits >= 0 ? f * (int)((unsigned)mf << qbits) : (f * mf) >> (-qbits);
                    v = (int)(int16_t)v;
                } else {
                    int blk = 16 + 4 * p + j;
                    int c = (mask >> blk) & 1 ? cf[coef_slot(mask, blk) * 16 + k - 1] : 0;
                    v = dequant_coef(c, pos, qpc);
                }
                L.coef[j * 16 + pos] = (int16_t)v;
                wave_lds_fence();
                uint8_t *o = (uint8_t *)L.outC + p * 64 + py * 8 + px;
                *o = (uint8_t)add_residual8x8(L.coef, 15u, px, py, *o);
                wave_lds_fence();
            }
        }
    } else wave_lds_fence();

    // ---------------- coalesced write-out ----------------
    wave_lds_fence();
    {
        int row = lane >> 2, d = lane & 3;
        *(uint32_t *)(pd->dst + (size_t)(Y0 + row) * g.w + X0 + d * 4) = L.outY[lane];
        if (lane < 32) {
            int p = lane >> 4, r = (lane >> 1) & 7, dd = lane & 1;
            *(uint32_t *)(pd->dst + (p ? g.off_v : g.off_u) + (size_t)(Y0 / 2 + r) * g.cw + X0 / 2 + dd * 4) = L.outC[lane];
        }
    }
}
