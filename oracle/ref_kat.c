/* oracle/ref_kat.c - TEST INFRASTRUCTURE ONLY.
 *
 * Known-answer harness of OUR OWN around the REAL reference objects (oracle/_ref): it fills the
 * reference's own function tables (decoder/decoder.c:701-711) and exposes each hot-path entry as a
 * flat C function, so tests/golden/make_kat.py can record (input, output) vectors.  The vectors
 * are committed; this file and the reference are not needed to run the tests.
 */
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "core/core.h"

static p264_t *g_h;
static p264_pps_t g_pps;

int refk_init(void)
{
    if (g_h) return 0;
    g_h = calloc(1, sizeof(p264_t));
    if (!g_h) return -1;
    for (int i = 0; i < 6; i++) g_pps.scaling_list[i] = p264_cqm_flat16;   /* decoder/set.c:261-263 */
    g_h->pps = &g_pps;
    p264_cqm_init(g_h);                                                    /* core/set.c:69-117 */
    p264_predict_16x16_init(0, g_h->predict_16x16);
    p264_predict_8x8c_init(0, g_h->predict_8x8c);
    p264_predict_4x4_init(0, g_h->predict_4x4);
    p264_dct_init(0, &g_h->dctf);
    p264_mc_init(0, &g_h->mc);
    p264_quant_init(g_h, 0, &g_h->quantf);
    p264_deblock_init(0, &g_h->loopf);
    return 0;
}

/* dequant_4x4 (list: 0 = CQM_4IY, 1 = 4IC, 2 = 4PY, 3 = 4PC) then add4x4_idct */
void refk_dequant_idct_add(int16_t coef[16], int qp, int list, uint8_t *dst, int stride)
{
    int16_t d[4][4];
    memcpy(d, coef, sizeof d);
    g_h->quantf.dequant_4x4(d, g_h->dequant4_mf[list], qp);
    memcpy(coef, d, sizeof d);
    g_h->dctf.add4x4_idct(dst, stride, d);
}
void refk_luma_dc(int16_t d16[16], int qp)
{
    int16_t d[4][4];
    memcpy(d, d16, sizeof d);
    g_h->dctf.idct4x4dc(d);
    p264_mb_dequant_4x4_dc(d, g_h->dequant4_mf[0], qp);
    memcpy(d16, d, sizeof d);
}
void refk_chroma_dc(int16_t d4[4], int qp)
{
    int16_t d[2][2];
    memcpy(d, d4, sizeof d);
    g_h->dctf.idct2x2dc(d);
    p264_mb_dequant_2x2_dc(d, g_h->dequant4_mf[1], qp);
    memcpy(d4, d, sizeof d);
}
void refk_pred16x16(uint8_t *dst, int stride, int mode) { g_h->predict_16x16[mode](dst, stride); }
void refk_pred8x8c(uint8_t *dst, int stride, int mode)  { g_h->predict_8x8c[mode](dst, stride); }
void refk_pred4x4(uint8_t *dst, int stride, int mode)   { g_h->predict_4x4[mode](dst, stride); }

/* ---- motion compensation on a real reference frame (pads + half-pel planes) ---------------- */
static p264_frame_t *g_frame;
static int g_fw, g_fh;

int refk_frame_set(const uint8_t *y, const uint8_t *u, const uint8_t *v, int w, int h)
{
    if (g_frame && (g_fw != w || g_fh != h)) { p264_frame_delete(g_frame); g_frame = NULL; }
    if (!g_frame) {
        g_h->param.i_width = w; g_h->param.i_height = h; g_h->param.i_csp = P264_CSP_I420;
        g_h->mb.i_mb_count = (w / 16) * (h / 16);
        g_frame = p264_frame_new(g_h);                                     /* core/frame.c:30-121 */
        g_fw = w; g_fh = h;
    }
    const uint8_t *src[3] = { y, u, v };
    for (int p = 0; p < 3; p++) {
        int pw = p ? w / 2 : w, ph = p ? h / 2 : h;
        for (int r = 0; r < ph; r++) memcpy(g_frame->plane[p] + r * g_frame->i_stride[p], src[p] + r * pw, pw);
    }
    p264_frame_expand_border(g_frame);                                     /* decoder/decoder.c:644-649 */
    p264_frame_filter(0, g_frame);
    p264_frame_expand_border_filtered(g_frame);
    return 0;
}

/* exactly the calls of p264_mb_mc_0xywh (core/macroblock.c:506-524) for a block at MB (mbx,mby),
 * 4x4-unit offset (x,y), size (bw,bh) in 4x4 units; outputs packed bw*4 x bh*4 and 2 x (bw*2 x bh*2) */
void refk_mc_block(int mbx, int mby, int x, int y, int bw, int bh, int mvx, int mvy,
                   uint8_t *oy, uint8_t *ou, uint8_t *ov)
{
    int s0 = g_frame->i_stride[0], s1 = g_frame->i_stride[1];
    uint8_t *src[4];
    for (int i = 0; i < 4; i++) src[i] = g_frame->filtered[i] + 16 * (mbx + mby * s0);
    g_h->mc.mc_luma(src, s0, oy, 4 * bw, mvx + 4 * 4 * x, mvy + 4 * 4 * y, 4 * bw, 4 * bh);
    uint8_t *cu = g_frame->plane[1] + 8 * (mbx + mby * s1) + 2 * y * s1 + 2 * x;
    uint8_t *cv = g_frame->plane[2] + 8 * (mbx + mby * s1) + 2 * y * s1 + 2 * x;
    g_h->mc.mc_chroma(cu, s1, ou, 2 * bw, mvx, mvy, 2 * bw, 2 * bh);
    g_h->mc.mc_chroma(cv, s1, ov, 2 * bw, mvx, mvy, 2 * bw, 2 * bh);
}

/* ---- deblocking sample filters: which = 0 v_luma, 1 h_luma, 2 v_chroma, 3 h_chroma (inter, tc[4]);
 *      4..7 the same order, intra ------------------------------------------------------------- */
void refk_deblock(int which, uint8_t *pix, int stride, int alpha, int beta, int8_t *tc)
{
    p264_deblock_function_t *f = &g_h->loopf;
    switch (which) {
    case 0: f->deblock_v_luma(pix, stride, alpha, beta, tc); break;
    case 1: f->deblock_h_luma(pix, stride, alpha, beta, tc); break;
    case 2: f->deblock_v_chroma(pix, stride, alpha, beta, tc); break;
    case 3: f->deblock_h_chroma(pix, stride, alpha, beta, tc); break;
    case 4: f->deblock_v_luma_intra(pix, stride, alpha, beta); break;
    case 5: f->deblock_h_luma_intra(pix, stride, alpha, beta); break;
    case 6: f->deblock_v_chroma_intra(pix, stride, alpha, beta); break;
    case 7: f->deblock_h_chroma_intra(pix, stride, alpha, beta); break;
    }
}

/* ---- bi-prediction: pf->avg[] / pf->avg_weight[] (core/mc.c:76-155, tables :366-379),
 *      which = PIXEL_16x16 .. PIXEL_2x2 (core/pixel.h) --------------------------------------------- */
void refk_avg(int which, uint8_t *dst, int dst_stride, uint8_t *src, int src_stride) { g_h->mc.avg[which](dst, dst_stride, src, src_stride); }
void refk_avg_weight(int which, uint8_t *dst, int dst_stride, uint8_t *src, int src_stride, int weight1) { g_h->mc.avg_weight[which](dst, dst_stride, src, src_stride, weight1); }

/* ---- CABAC (SURVEY 8c "partial pins"): the reference ENCODES a sequence of bins with its arithmetic coder
 *      (p264_cabac_context_init / _encode_init / _encode_decision / _encode_bypass / _encode_terminal / _encode_flush,
 *      core/cabac.c:819-837, 907-1018); a decoder must get the same bins back out of the bytes.  ops[i] >= 0: decision with
 *      context ops[i] (< 436: the reference initialises no more, core/cabac.c:835); -1 bypass; -2 terminal.
 *      Returns the number of bytes written. ------------------------------------------------------------------------ */
int refk_cabac_encode(int slice_type, int qp, int model, const int16_t *ops, const uint8_t *bins, int n, uint8_t *out, int cap)
{
    p264_cabac_t cb;
    bs_t s;
    memset(out, 0, (size_t)cap);
    bs_init(&s, out, cap);
    memset(&cb, 0, sizeof cb);
    p264_cabac_context_init(&cb, slice_type, qp, model);
    p264_cabac_encode_init(&cb, &s);
    for (int i = 0; i < n; i++) {
        if (ops[i] >= 0) p264_cabac_encode_decision(&cb, ops[i], bins[i]);
        else if (ops[i] == -1) p264_cabac_encode_bypass(&cb, bins[i]);
        else p264_cabac_encode_terminal(&cb, bins[i]);
    }
    p264_cabac_encode_flush(&cb);
    return bs_pos(&s) / 8;
}
int refk_slice_type_i(void) { return SLICE_TYPE_I; }
int refk_slice_type_p(void) { return SLICE_TYPE_P; }
