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

/* ---- B pictures: implicit weights and direct prediction (SURVEY 8c "partial pins" for configs 4-5) -----------------------
 *      p264_macroblock_bipred_init (core/macroblock.c:1400-1430) and p264_mb_predict_mv_direct16x16 (:254-413) are encoder-side
 *      code the reference's decoder never reaches, but they are non-static: driven here on hand-made state. */
static p264_frame_t g_l0[16], g_l1[16], g_cur, g_col;
static int8_t g_col_type[1], g_col_ref[2][4];
static int16_t g_col_mv[2][16][2];

/* weights / distance scale factors of every (list-0, list-1) index pair from the pictures' order counts */
void refk_bipred_init(int n0, const int *poc0, int n1, const int *poc1, int cur_poc, int weighted, int *dsf_out, int *w_out)
{
    for (int i = 0; i < n0; i++) { g_l0[i].i_poc = poc0[i]; g_h->fref0[i] = &g_l0[i]; }
    for (int i = 0; i < n1; i++) { g_l1[i].i_poc = poc1[i]; g_h->fref1[i] = &g_l1[i]; }
    g_h->i_ref0 = n0; g_h->i_ref1 = n1;
    g_cur.i_poc = cur_poc; g_h->fdec = &g_cur;
    g_h->param.analyse.b_weighted_bipred = weighted;
    p264_macroblock_bipred_init(g_h);
    for (int i = 0; i < 16; i++)
        for (int k = 0; k < 16; k++) { dsf_out[i * 16 + k] = g_h->mb.dist_scale_factor[i][k]; w_out[i * 16 + k] = g_h->mb.bipred_weight[i][k]; }
}

/* Direct prediction of one macroblock.
 *   nb_ref[l][n], nb_mv[l][n][2]: the neighbours A (left), B (top), C (top right), D (top left) of the macroblock per list;
 *                                 ref -2 = not available, -1 = intra / list unused
 *   col_*: the co-located macroblock in RefPicList1[0]: intra or not, reference indices per 8x8 and vectors per 4x4 (raster),
 *          both lists
 *   temporal only: map_col[i] = index in the current list 0 of the picture the co-located picture's list-0 index i names
 *          (-2 = not there), dsf[i] = dist_scale_factor[i][0] (refk_bipred_init)
 * Returns what the function returns (0 = "direct prediction not available"); out_ref[l][q], out_mv[l][4x4 raster][2]. */
int refk_direct(int spatial, const int8_t *nb_ref, const int16_t *nb_mv, int col_intra, const int8_t *col_ref, const int16_t *col_mv,
                const int *map_col, const int *dsf, int8_t *out_ref, int16_t *out_mv)
{
    p264_t *h = g_h;
    static const int nb_at[4] = { P264_SCAN8_0 - 1, P264_SCAN8_0 - 8, P264_SCAN8_0 - 8 + 4, P264_SCAN8_0 - 8 - 1 };
    memset(&h->mb.cache, 0, sizeof h->mb.cache);
    for (int l = 0; l < 2; l++) {
        memset(h->mb.cache.ref[l], -2, sizeof h->mb.cache.ref[l]);
        for (int n = 0; n < 4; n++) {
            h->mb.cache.ref[l][nb_at[n]] = nb_ref[l * 4 + n];
            h->mb.cache.mv[l][nb_at[n]][0] = nb_mv[(l * 4 + n) * 2]; h->mb.cache.mv[l][nb_at[n]][1] = nb_mv[(l * 4 + n) * 2 + 1];
        }
    }
    g_col_type[0] = col_intra ? I_4x4 : P_L0;
    memcpy(g_col_ref, col_ref, sizeof g_col_ref);
    memcpy(g_col_mv, col_mv, sizeof g_col_mv);
    g_col.mb_type = g_col_type;
    for (int l = 0; l < 2; l++) { g_col.ref[l] = g_col_ref[l]; g_col.mv[l] = g_col_mv[l]; }
    h->fref1[0] = &g_col;
    h->mb.i_mb_x = h->mb.i_mb_y = h->mb.i_mb_xy = h->mb.i_b8_xy = h->mb.i_b4_xy = 0;
    h->mb.i_mb_stride = 1; h->mb.i_b8_stride = 2; h->mb.i_b4_stride = 4;
    h->mb.i_partition = D_16x16;
    h->param.analyse.i_direct_mv_pred = spatial ? P264_DIRECT_PRED_SPATIAL : P264_DIRECT_PRED_TEMPORAL;
    h->sh.b_direct_spatial_mv_pred = spatial;
    h->mb.map_col_to_list0[-1] = -1; h->mb.map_col_to_list0[-2] = -2;
    for (int i = 0; i < 16; i++) { h->mb.map_col_to_list0[i] = map_col[i]; h->mb.dist_scale_factor[i][0] = dsf[i]; }
    const int ok = p264_mb_predict_mv_direct16x16(h);
    for (int l = 0; l < 2; l++) {
        for (int q = 0; q < 4; q++) out_ref[l * 4 + q] = h->mb.cache.ref[l][P264_SCAN8_0 + (q & 1) * 2 + (q >> 1) * 16];
        for (int b = 0; b < 16; b++) {
            out_mv[(l * 16 + b) * 2] = h->mb.cache.mv[l][P264_SCAN8_0 + (b & 3) + (b >> 2) * 8][0];
            out_mv[(l * 16 + b) * 2 + 1] = h->mb.cache.mv[l][P264_SCAN8_0 + (b & 3) + (b >> 2) * 8][1];
        }
    }
    return ok;
}

/* ---- the whole deblocking driver (core/frame.c:490-643: raster order, bS derivation :535-581, edge QPs :593-601, the table
 *      look-ups and the tc = tc0 (+1 for chroma) of deblock_edge :472-488, the eight sample filters) on hand-made state.
 *      Planes are MB-aligned without pads (stride = width), filtered in place.  Per macroblock: intra or not, QP, the 16 luma
 *      4x4 blocks with coefficients as a bit mask in decode order (= the index of non_zero_count), list-0 reference index per
 *      8x8 and vector per 4x4 (raster inside the macroblock).  P slices. ------------------------------------------------ */
int refk_deblock_frame(int mb_w, int mb_h, uint8_t *y, uint8_t *u, uint8_t *v, const uint8_t *intra, const uint8_t *qp,
                       const uint32_t *nnz_mask, const int8_t *ref8, const int16_t *mv4, int chroma_qp_offset, int alpha_off, int beta_off)
{
    static p264_sps_t sps;
    p264_t *h = g_h;
    const int n = mb_w * mb_h;
    p264_frame_t fr;
    memset(&fr, 0, sizeof fr);
    int8_t *type = malloc(n), *q = malloc(n), *ts = calloc(n, 1), *ref = malloc(n * 4);
    uint8_t (*nzc)[24] = calloc(n, 24);
    int16_t (*mv)[2] = calloc(n * 16, sizeof(int16_t[2]));
    if (!type || !q || !ts || !ref || !nzc || !mv) return -1;
    for (int m = 0; m < n; m++) {
        const int mx = m % mb_w, my = m / mb_w;
        type[m] = intra[m] ? I_16x16 : P_L0;
        q[m] = (int8_t)qp[m];
        for (int b = 0; b < 16; b++) nzc[m][b] = (nnz_mask[m] >> b) & 1;
        for (int k = 0; k < 4; k++) ref[(2 * my + (k >> 1)) * 2 * mb_w + 2 * mx + (k & 1)] = ref8[m * 4 + k];
        for (int k = 0; k < 16; k++) {
            int16_t *d = mv[(4 * my + (k >> 2)) * 4 * mb_w + 4 * mx + (k & 3)];
            d[0] = mv4[(m * 16 + k) * 2]; d[1] = mv4[(m * 16 + k) * 2 + 1];
        }
    }
    sps.i_mb_width = mb_w; sps.i_mb_height = mb_h;
    h->sps = &sps;
    g_pps.i_chroma_qp_index_offset = chroma_qp_offset;
    h->sh.i_alpha_c0_offset = alpha_off; h->sh.i_beta_offset = beta_off;
    h->param.b_cabac = 0;
    h->mb.i_mb_stride = mb_w;
    h->mb.type = type; h->mb.qp = q; h->mb.mb_transform_size = ts; h->mb.non_zero_count = nzc;
    h->mb.ref[0] = ref; h->mb.mv[0] = mv;
    fr.plane[0] = y; fr.plane[1] = u; fr.plane[2] = v;
    fr.i_stride[0] = 16 * mb_w; fr.i_stride[1] = fr.i_stride[2] = 8 * mb_w;
    h->fdec = &fr;
    p264_frame_deblocking_filter(h, SLICE_TYPE_P);
    h->fdec = NULL; h->mb.type = NULL; h->mb.qp = NULL; h->mb.mb_transform_size = NULL; h->mb.non_zero_count = NULL; h->mb.ref[0] = NULL; h->mb.mv[0] = NULL;
    free(type); free(q); free(ts); free(ref); free(nzc); free(mv);
    return 0;
}
