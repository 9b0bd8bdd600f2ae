/* oracle/cpu_recon.h - TEST INFRASTRUCTURE ONLY.
 *
 * Scalar-C restatement of the reference's macroblock-reconstruction hot path
 * (dequant + inverse transforms, intra prediction, motion compensation, deblocking).
 * It consumes exactly the per-picture buffers the GPU layer consumes (include/p264hip.h),
 * so a test can run both on the same input and compare bytes.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * Pinned against the real reference (oracle/_ref) - see tests/test_oracle_vs_reference.py
 * and the golden fixtures under tests/golden/.
 */
#ifndef ORACLE_CPU_RECON_H
#define ORACLE_CPU_RECON_H
#include <stdint.h>
#include "../include/p264hip.h"

/* planes[slot*3 + {0,1,2}] = Y,U,V of frame-store slot `slot`; luma stride = 16*mb_w,
 * chroma stride = 8*mb_w, no padding (references are read with clamped coordinates,
 * which equals the reference's padded planes inside its pads - SURVEY A-Q9). */
int oracle_reconstruct(const p264hip_picture_t *pic, uint8_t **planes);
/* same without the loop filter (to test stages separately) */
int oracle_reconstruct_nodeblock(const p264hip_picture_t *pic, uint8_t **planes);
int oracle_deblock_picture(const p264hip_picture_t *pic, uint8_t **planes);

/* coverage counters, see cpu_recon.c */
void oracle_stats_reset(void);
void oracle_stats_get(long long out[8]);
long long oracle_bs_by_picture(void);
long long oracle_bipred_blocks(void);      /* bi-predicted 4x4 blocks since the last reset (B pictures) */

/* kernel-level entry points, one per reference function table entry (for known-answer tests) */
void oracle_dequant4x4(int16_t d[16], int qp);                               /* core/quant.c:72-99  */
void oracle_idct4x4dc(int16_t d[16]);                                        /* core/dct.c:104-136  */
void oracle_dequant4x4_dc(int16_t d[16], int qp);                            /* core/quant.c:161-191 */
void oracle_idct2x2dc(int16_t d[4]);                                         /* core/dct.c:55-68    */
void oracle_dequant2x2_dc(int16_t d[4], int qp);                             /* core/quant.c:138-159 */
void oracle_add4x4_idct(uint8_t *dst, int stride, const int16_t d[16]);      /* core/dct.c:205-247  */
/* mode numbering = H.264 / core/predict.h; neighbours are read from dst[-1], dst[-stride] */
void oracle_pred16x16(uint8_t *dst, int stride, int mode);                   /* core/predict.c:55-193  (0 V 1 H 2 DC 3 P 4 DC_LEFT 5 DC_TOP 6 DC_128) */
void oracle_pred8x8c(uint8_t *dst, int stride, int mode);                    /* core/predict.c:199-361 (0 DC 1 H 2 V 3 P 4 DC_LEFT 5 DC_TOP 6 DC_128) */
void oracle_pred4x4(uint8_t *dst, int stride, int mode);                     /* core/predict.c:366-638 (0 V 1 H 2 DC 3 DDL 4 DDR 5 VR 6 HD 7 VL 8 HU 9 DC_LEFT 10 DC_TOP 11 DC_128) */
void oracle_mc_luma(const uint8_t *ref, int w, int h, int x, int y, int mvx, int mvy,
                    int bw, int bh, uint8_t *dst, int dst_stride);           /* core/mc.c:237-266 + 172-235 */
void oracle_mc_chroma(const uint8_t *ref, int w, int h, int x, int y, int mvx, int mvy,
                      int bw, int bh, uint8_t *dst, int dst_stride);         /* core/mc.c:303-334 */
/* bi-prediction (not reachable through the reference's decoder, which has no B slices - pinned through its
 * function tables): dst = (dst + src + 1) >> 1, and the implicit-weight form clip((dst*w1 + src*(64-w1) + 32) >> 6) */
void oracle_bipred_avg(uint8_t *dst, int dst_stride, const uint8_t *src, int src_stride, int w, int h);                     /* core/mc.c:76-88  */
void oracle_bipred_weight(uint8_t *dst, int dst_stride, const uint8_t *src, int src_stride, int w, int h, int weight1);     /* core/mc.c:106-132 */
void oracle_deblock_luma(uint8_t *pix, int xstride, int ystride, int alpha, int beta, const int8_t tc0[4]);   /* core/frame.c:302-341 */
void oracle_deblock_chroma(uint8_t *pix, int xstride, int ystride, int alpha, int beta, const int8_t tc[4]);  /* core/frame.c:351-377 */
void oracle_deblock_luma_intra(uint8_t *pix, int xstride, int ystride, int alpha, int beta);                  /* core/frame.c:387-433 */
void oracle_deblock_chroma_intra(uint8_t *pix, int xstride, int ystride, int alpha, int beta);                /* core/frame.c:443-462 */
#endif
