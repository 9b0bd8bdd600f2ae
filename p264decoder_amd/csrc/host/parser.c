/* parser.c - host-side H.264 bitstream layer (see include/p264parse.h).
 *
 * CPU work by design: entropy decoding is bit-serial.  Everything it learns about a picture
 * is written straight into the structure-of-arrays buffers of p264hip_picture_t, which the
 * HIP layer uploads as they are.
 *
 * Supported subset = what the reference decodes: CAVLC, I and P slices,
 * frame MBs, one reference list.  Beyond it we follow ITU-T H.264 (several slices per picture,
 * multiple reference frames, list reordering, sub-8x8 partitions, memory management control
 * operations, and - SURVEY 8f rank 4, where the reference stops at decoder/lists.c:136 and
 * decoder/macroblock.c:168-171 - CAVLC B slices: two lists ordered by picture order count,
 * every B macroblock and sub-macroblock type, spatial and temporal direct prediction, implicit
 * bi-prediction weights); everything else is rejected with -1 and a line on stderr, the
 * reference's error convention (decoder/decoder.c:558-577,780-795).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "p264parse.h"
#include "host_cpu.h"
#include "bits.h"
#include "vlc.h"
#include "cavlc_tables.h"
#include "cabac.h"

#define NAL_SLICE     1
#define NAL_SLICE_DPA 2
#define NAL_SLICE_DPB 3
#define NAL_SLICE_DPC 4
#define NAL_SLICE_IDR 5
#define NAL_SPS       7
#define NAL_PPS       8

typedef struct {
    int valid, profile_idc, level_idc;
    int log2_max_frame_num, poc_type, log2_max_poc_lsb;
    int delta_pic_order_always_zero, num_ref_frames_in_poc_cycle;
    int num_ref_frames, gaps_allowed, mb_w, mb_h, frame_mbs_only, direct_8x8_inference;
    int crop[4];
} sps_t;

typedef struct {
    int valid, sps_id, cabac, pic_order_present, num_slice_groups;
    int num_ref_idx_l0, num_ref_idx_l1, weighted_pred, weighted_bipred;
    int pic_init_qp, chroma_qp_offset, deblock_ctrl, constrained_intra, redundant_pic_cnt;
} pps_t;

typedef struct {
    int first_mb, type, pps_id, frame_num, idr_pic_id;
    int num_ref_idx, qp, disable_deblock, alpha_off, beta_off;
    int n_reorder; struct { int idc, arg; } reorder[34];
    int num_ref_idx_l1, n_reorder1; struct { int idc, arg; } reorder1[34];   /* B slices: list 1 */
    int poc_lsb, delta_poc_bottom, direct_spatial, cabac_init_idc;
    int no_output_of_prior, long_term_flag, adaptive_marking;
    int n_mmco; struct { int op, a, b; } mmco[34];   /* memory_management_control_operation 1..6 and its operands */
} slice_t;

typedef struct { int used, frame_num, pic_num, is_long, long_idx;   /* long_idx = LongTermFrameIdx (= LongTermPicNum for frames) */
                 int poc; uint32_t uid; } dpb_frame_t;              /* picture order count; uid: which decoded picture the slot holds */

typedef struct {
    p264hip_mb_t *mb; int16_t *mv; int8_t *ref; uint8_t *i4; int16_t *coef;
    int16_t *mv1; int8_t *ref1;               /* list 1 (B pictures; allocated with the others for non-Baseline streams) */
    size_t coef_cap, coef_n;
    /* mb, mv, ref, i4 and coef are sections of ONE allocation laid out like an input slot of the HIP layer (p264hip_input_layout:
     * records | vectors | reference indices | intra 4x4 modes | coded levels, each on a 256-byte boundary), so that a picture goes
     * host -> HBM as one copy (p264hip_upload / _upload_async notice it); coef_own: the coded levels outgrew their section and
     * moved to an allocation of their own (the picture then travels in pieces, as every picture did until round 5) */
    uint8_t *block; int coef_own;
    void *(*alloc)(size_t); void (*release)(void *);   /* where the arrays live (p264parse_set_allocator) */
} picbuf_t;

struct p264parse {
    int opts;
    sps_t sps[32];
    pps_t pps[256];
    int active_sps, active_pps, generation;
    int mb_w, mb_h, n_mb, slots;

    picbuf_t buf[2]; int cur;                 /* cur: being built; 1-cur: last completed */
    picbuf_t retired[2];                      /* the buffers of the previous context: released at the NEXT re-init (see free_context) */
    void *(*alloc)(size_t); void (*release)(void *);
    p264hip_picture_t desc[2];
    uint8_t  *nnz;                            /* [n_mb][24] total_coeff per 4x4 block */
    uint16_t *slice_of;                       /* [n_mb] slice number inside the picture, 0xffff = not decoded */

    int pic_open, next_mb, slice_no;
    int pic_is_idr, pic_ref_idc;
    slice_t sh;                               /* current slice */
    slice_t sh0;                              /* first slice of the picture */
    int pic_deblock, pic_alpha, pic_beta;     /* loop filter of the picture: on if any slice enables it, offsets of the first such slice */
    int list0[P264HIP_MAX_REFS], n_list0;
    int list1[P264HIP_MAX_REFS], n_list1;     /* B pictures */
    int16_t bipred_weight[P264HIP_MAX_REFS * P264HIP_MAX_REFS];   /* implicit weights of the picture (8.4.2.3.1) */
    int weighted_bipred;
    /* picture order count (8.2.1) */
    int cur_poc, prev_poc_msb, prev_poc_lsb, prev_frame_num, frame_num_offset;
    uint32_t next_uid, cur_uid;
    /* motion of every reference picture, kept for the direct prediction of later B pictures (the co-located picture is
     * RefPicList1[0]): per frame-store slot, per 4x4 block the vector, per 8x8 the reference index it used and the uid of the
     * picture that index meant (-1 = intra) */
    int has_col;
    int16_t *col_mv[P264HIP_MAX_REFS + 1]; int8_t *col_ref[P264HIP_MAX_REFS + 1]; int32_t *col_uid[P264HIP_MAX_REFS + 1];

    dpb_frame_t dpb[P264HIP_MAX_REFS + 1];
    int cur_slot;
    int last_qp;                              /* never reset, like h->mb.i_last_qp (core/macroblock.c:1248-1252) */
    int qp_pred;                              /* conformant chain only (strict_qp) */
    int strict_qp;                            /* this slice: QP_Y = (QP_Y,PRED + mb_qp_delta + 52) % 52 (H.264 7.4.5) */

    /* current MB */
    int mbx, mby, mbi;
    unsigned mv_done, mv_done1;               /* bit (y*4+x): that 4x4 of the current MB has its list-0 / list-1 motion */
    int      cur_avail;                       /* P264_AVAIL_* of the current MB (set by begin_mb) */
    int skip_run;
    /* CABAC (parser_cabac.h): the engine, and what context selection needs from earlier macroblocks */
    int cabac_on, last_dqp;
    p264cabac_t cb;
    uint16_t *cinfo;                          /* [n_mb] CI_* */
    uint8_t *mvd_abs[2];                      /* [n_mb][16][2] |mvd| per list, 4x4 block and component, saturated at 255 */
};

#define ERR(p, ...) do { fprintf(stderr, "p264amd: " __VA_ARGS__); fputc('\n', stderr); } while (0)
#define INFO(p, ...) do { if (!((p)->opts & P264PARSE_OPT_QUIET)) { fprintf(stderr, __VA_ARGS__); } } while (0)

static inline int clip3i(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
static inline int median3(int a, int b, int c)
{
    int lo = a < b ? a : b, hi = a < b ? b : a;
    return c < lo ? lo : c > hi ? hi : c;
}

/* ---------------------------------------------------------------- parameter sets -------- */
/* decoder/set.c:37-167 */
static int parse_sps(p264parse *p, bitrd_t *b)
{
    int profile = (int)br_u(b, 8);
    br_skip(b, 8);                                  /* constraint flags + reserved */
    int level = (int)br_u(b, 8);
    unsigned id = br_ue(b);
    if (br_eof(b) || id >= 32) return -1;
    sps_t *s = &p->sps[id];
    sps_t old = *s;
    memset(s, 0, sizeof *s);
    s->profile_idc = profile; s->level_idc = level;
    s->log2_max_frame_num = (int)br_ue(b) + 4;
    s->poc_type = (int)br_ue(b);
    if (s->poc_type == 0) s->log2_max_poc_lsb = (int)br_ue(b) + 4;
    else if (s->poc_type == 1) {
        s->delta_pic_order_always_zero = (int)br_u1(b);
        br_se(b); br_se(b);
        s->num_ref_frames_in_poc_cycle = (int)br_ue(b);
        if (s->num_ref_frames_in_poc_cycle > 256) s->num_ref_frames_in_poc_cycle = 256;
        for (int i = 0; i < s->num_ref_frames_in_poc_cycle; i++) br_se(b);
    } else if (s->poc_type > 2) return -1;
    s->num_ref_frames = (int)br_ue(b);
    s->gaps_allowed = (int)br_u1(b);
    s->mb_w = (int)br_ue(b) + 1;
    s->mb_h = (int)br_ue(b) + 1;
    s->frame_mbs_only = (int)br_u1(b);
    if (!s->frame_mbs_only) br_u1(b);
    s->direct_8x8_inference = (int)br_u1(b);
    if (br_u1(b)) for (int i = 0; i < 4; i++) s->crop[i] = (int)br_ue(b);   /* parsed, never applied (A-Q1) */
    br_u1(b);                                       /* vui_parameters_present: not parsed, like set.c:136-144 */
    if (br_eof(b)) { ERR(p, "incomplete SPS"); return -1; }
    /* untrusted input: H.264 7.4.2.1 ranges (the slice header reads fields of these widths; sizes drive every allocation) */
    if (s->log2_max_frame_num < 4 || s->log2_max_frame_num > 16 || (s->poc_type == 0 && (s->log2_max_poc_lsb < 4 || s->log2_max_poc_lsb > 16)) ||
        s->mb_w < 1 || s->mb_w > 1024 || s->mb_h < 1 || s->mb_h > 512 || s->num_ref_frames < 0 || s->num_ref_frames > 16) {
        ERR(p, "SPS field out of range (frame_num bits %d, poc bits %d, %dx%d macroblocks, %d reference frames)",
            s->log2_max_frame_num, s->log2_max_poc_lsb, s->mb_w, s->mb_h, s->num_ref_frames);
        *s = old;                                   /* keep the set we had */
        return -1;
    }
    s->valid = 1;
    if ((int)id == p->active_sps && (old.mb_w != s->mb_w || old.mb_h != s->mb_h || old.num_ref_frames != s->num_ref_frames))
        p->active_sps = -1;                         /* same id, new geometry: force a context re-init */
    INFO(p, "p264amd: sps:%u profile:%d/%d poc:%d ref:%d %dx%d crop:%d-%d-%d-%d\n", id, profile, level,
         s->poc_type, s->num_ref_frames, s->mb_w, s->mb_h, s->crop[0], s->crop[1], s->crop[2], s->crop[3]);
    return (int)id;
}

/* decoder/set.c:171-272; scaling lists are forced flat there (:261-263), so none are parsed */
static int parse_pps(p264parse *p, bitrd_t *b)
{
    unsigned id = br_ue(b);
    if (br_eof(b) || id >= 256) { ERR(p, "pps id invalid"); return -1; }
    pps_t *q = &p->pps[id];
    memset(q, 0, sizeof *q);
    q->sps_id = (int)br_ue(b);
    if (q->sps_id < 0 || q->sps_id >= 32) return -1;
    q->cabac = (int)br_u1(b);
    q->pic_order_present = (int)br_u1(b);
    q->num_slice_groups = (int)br_ue(b) + 1;
    if (q->num_slice_groups > 1) { ERR(p, "FMO unsupported"); return -1; }
    q->num_ref_idx_l0 = (int)br_ue(b) + 1;
    q->num_ref_idx_l1 = (int)br_ue(b) + 1;
    if (q->num_ref_idx_l0 < 1 || q->num_ref_idx_l0 > 32 || q->num_ref_idx_l1 < 1 || q->num_ref_idx_l1 > 32) { ERR(p, "pps: num_ref_idx out of range"); return -1; }
    q->weighted_pred = (int)br_u1(b);
    q->weighted_bipred = (int)br_u(b, 2);
    q->pic_init_qp = br_se(b) + 26;
    br_se(b);                                       /* pic_init_qs */
    q->chroma_qp_offset = br_se(b);
    q->deblock_ctrl = (int)br_u1(b);
    q->constrained_intra = (int)br_u1(b);
    q->redundant_pic_cnt = (int)br_u1(b);
    if (br_eof(b)) { ERR(p, "incomplete PPS"); return -1; }
    q->valid = 1;
    INFO(p, "p264amd: pps:%u sps:%d %s ref0:%d QP:%d QC=%d DFC:%d CIP:%d\n", id, q->sps_id,
         q->cabac ? "CABAC" : "CAVLC", q->num_ref_idx_l0, q->pic_init_qp, q->chroma_qp_offset,
         q->deblock_ctrl, q->constrained_intra);
    return (int)id;
}

/* ---------------------------------------------------------------- context --------------- */
static void release_bufs(picbuf_t *q)
{
    if (q->release) { q->release(q->block); if (q->coef_own) q->release(q->coef); if (q->mv1) q->release(q->mv1); if (q->ref1) q->release(q->ref1); }
    memset(q, 0, sizeof *q);
}
/* A caller may still be reading the last completed picture's arrays when the next slice re-initialises the context (the
 * pipeline issues its asynchronous uploads while the parser threads are already on the next picture), so a re-init only
 * RETIRES the current buffers; they are released by the re-init after that, or by close. */
static void free_context(p264parse *p, int final)
{
    for (int i = 0; i < 2; i++) {
        release_bufs(&p->retired[i]);
        if (final) release_bufs(&p->buf[i]);
        else { p->retired[i] = p->buf[i]; memset(&p->buf[i], 0, sizeof p->buf[i]); }
    }
    free(p->nnz); p->nnz = NULL;
    free(p->slice_of); p->slice_of = NULL;
    free(p->cinfo); p->cinfo = NULL; free(p->mvd_abs[0]); free(p->mvd_abs[1]); p->mvd_abs[0] = p->mvd_abs[1] = NULL;
    for (int i = 0; i <= P264HIP_MAX_REFS; i++) { free(p->col_mv[i]); free(p->col_ref[i]); free(p->col_uid[i]); p->col_mv[i] = NULL; p->col_ref[i] = NULL; p->col_uid[i] = NULL; }
    p->has_col = 0;
}

/* decoder/decoder.c:304-343: (re)size everything when the active SPS/PPS pair changes */
static int init_context(p264parse *p, int sps_id, int pps_id)
{
    const sps_t *s = &p->sps[sps_id];
    free_context(p, 0);
    p->mb_w = s->mb_w; p->mb_h = s->mb_h; p->n_mb = s->mb_w * s->mb_h;
    p->slots = s->num_ref_frames + 1;
    if (p->slots < 2) p->slots = 2;
    size_t n = (size_t)p->n_mb;
    for (int i = 0; i < 2; i++) {
        picbuf_t *q = &p->buf[i];
        q->alloc = p->alloc ? p->alloc : malloc; q->release = p->release ? p->release : free;
        q->coef_cap = n * 8 + 64;                   /* (at most 26 per macroblock; beyond the section: coef_reserve) */
        p264hip_picture_t shape; p264hip_input_layout_t lay;
        memset(&shape, 0, sizeof shape);
        shape.mb_w = s->mb_w; shape.mb_h = s->mb_h; shape.slice_type = P264_SLICE_P; shape.n_coef_blocks = (uint32_t)q->coef_cap;
        if (p264hip_input_layout(&shape, &lay)) return -1;
        q->block = (uint8_t *)q->alloc(lay.bytes);
        if (!q->block) return -1;
        q->mb  = (p264hip_mb_t *)(q->block + lay.off_mb);
        q->mv  = (int16_t *)(q->block + lay.off_mv);
        q->ref = (int8_t *)(q->block + lay.off_ref);
        q->i4  = (uint8_t *)(q->block + lay.off_i4);
        q->coef = (int16_t *)(q->block + lay.off_coef); q->coef_own = 0;
        memset(q->mb, 0, n * sizeof(p264hip_mb_t)); memset(q->mv, 0, n * 32 * sizeof(int16_t));
        memset(q->ref, 0, n * 4); memset(q->i4, 0, n * 16);
        if (s->profile_idc != 66) {                 /* anything but Baseline may carry B slices: list-1 arrays */
            q->mv1 = (int16_t *)q->alloc(n * 32 * sizeof(int16_t)); q->ref1 = (int8_t *)q->alloc(n * 4);
            if (!q->mv1 || !q->ref1) return -1;
            memset(q->mv1, 0, n * 32 * sizeof(int16_t)); memset(q->ref1, -1, n * 4);
        }
    }
    if (s->profile_idc != 66) {
        for (int i = 0; i < p->slots; i++) {
            p->col_mv[i] = (int16_t *)calloc(n * 32, sizeof(int16_t)); p->col_ref[i] = (int8_t *)malloc(n * 4); p->col_uid[i] = (int32_t *)malloc(n * 4 * sizeof(int32_t));
            if (!p->col_mv[i] || !p->col_ref[i] || !p->col_uid[i]) return -1;
            memset(p->col_ref[i], -1, n * 4); memset(p->col_uid[i], 0xff, n * 4 * sizeof(int32_t));
        }
        p->has_col = 1;
        p->cinfo = (uint16_t *)calloc(n, sizeof(uint16_t)); p->mvd_abs[0] = (uint8_t *)calloc(n, 32); p->mvd_abs[1] = (uint8_t *)calloc(n, 32);
        if (!p->cinfo || !p->mvd_abs[0] || !p->mvd_abs[1]) return -1;
    }
    p->nnz = (uint8_t *)calloc(n, 24);
    p->slice_of = (uint16_t *)malloc(n * sizeof(uint16_t));
    if (!p->nnz || !p->slice_of) return -1;
    memset(p->dpb, 0, sizeof p->dpb);
    p->cur_slot = 0;
    p->active_sps = sps_id; p->active_pps = pps_id;
    p->generation++;
    p->pic_open = 0;
    INFO(p, "p264amd: %dx%d\n", 16 * p->mb_w, 16 * p->mb_h);
    return 0;
}

/* ---------------------------------------------------------------- slice header ---------- */
/* decoder/decoder.c:70-301,368-488.  Fields the reconstruction does not need are skipped. */
static int parse_slice_header(p264parse *p, bitrd_t *b, int nal_type, int nal_ref_idc, slice_t *sh)
{
    memset(sh, 0, sizeof *sh);
    sh->first_mb = (int)br_ue(b);
    sh->type = (int)br_ue(b);
    if (sh->type >= 5) sh->type -= 5;
    sh->pps_id = (int)br_ue(b);
    if (br_eof(b) || sh->pps_id < 0 || sh->pps_id >= 256 || !p->pps[sh->pps_id].valid) {
        ERR(p, "invalid pps_id %d in slice header", sh->pps_id); return -1;
    }
    const pps_t *pps = &p->pps[sh->pps_id];
    if (!p->sps[pps->sps_id].valid) { ERR(p, "slice refers to missing sps %d", pps->sps_id); return -1; }
    const sps_t *sps = &p->sps[pps->sps_id];
    if (sh->type != P264_SLICE_P && sh->type != P264_SLICE_I && sh->type != P264_SLICE_B) { ERR(p, "only I, P and B slices supported (type %d)", sh->type); return -1; }
    if (sh->type == P264_SLICE_B && (sps->profile_idc == 66 || nal_type == NAL_SLICE_IDR)) { ERR(p, "B slice in a Baseline stream or an IDR picture"); return -1; }
    if (sh->type == P264_SLICE_B && sps->poc_type == 1) { ERR(p, "B slices with pic_order_cnt_type 1 unsupported"); return -1; }
    if (pps->cabac && sps->profile_idc == 66) { ERR(p, "CABAC in a Baseline stream"); return -1; }
    if (!sps->frame_mbs_only) { ERR(p, "field/MBAFF coding unsupported"); return -1; }

    sh->frame_num = (int)br_u(b, sps->log2_max_frame_num);
    if (nal_type == NAL_SLICE_IDR) sh->idr_pic_id = (int)br_ue(b);
    if (sps->poc_type == 0) {
        sh->poc_lsb = (int)br_u(b, sps->log2_max_poc_lsb);
        if (pps->pic_order_present) sh->delta_poc_bottom = br_se(b);
    } else if (sps->poc_type == 1 && !sps->delta_pic_order_always_zero) {
        br_se(b);
        if (pps->pic_order_present) br_se(b);
    }
    if (pps->redundant_pic_cnt && br_ue(b) != 0) return 1;       /* redundant picture: ignore the slice */
    sh->num_ref_idx = 0;
    if (sh->type == P264_SLICE_B) sh->direct_spatial = (int)br_u1(b);
    if (sh->type == P264_SLICE_P || sh->type == P264_SLICE_B) {
        sh->num_ref_idx = pps->num_ref_idx_l0; sh->num_ref_idx_l1 = pps->num_ref_idx_l1;
        if (br_u1(b)) { sh->num_ref_idx = (int)br_ue(b) + 1; if (sh->type == P264_SLICE_B) sh->num_ref_idx_l1 = (int)br_ue(b) + 1; }
        if (sh->num_ref_idx < 1 || sh->num_ref_idx > P264HIP_MAX_REFS) { ERR(p, "num_ref_idx_l0_active %d too large", sh->num_ref_idx); return -1; }
        if (sh->type == P264_SLICE_B && (sh->num_ref_idx_l1 < 1 || sh->num_ref_idx_l1 > P264HIP_MAX_REFS)) { ERR(p, "num_ref_idx_l1_active %d too large", sh->num_ref_idx_l1); return -1; }
        for (int l = 0; l < (sh->type == P264_SLICE_B ? 2 : 1); l++) {
            if (!br_u1(b)) continue;                              /* ref_pic_list_reordering_flag_l0 / _l1 */
            for (;;) {
                unsigned idc = br_ue(b);
                int *n = l ? &sh->n_reorder1 : &sh->n_reorder;
                if (idc == 3) break;
                if (idc > 3 || *n >= 33 || br_overrun(b)) { ERR(p, "wrong reordering of pic nums idc"); return -1; }
                if (l) { sh->reorder1[*n].idc = (int)idc; sh->reorder1[*n].arg = (int)br_ue(b); }
                else   { sh->reorder[*n].idc = (int)idc; sh->reorder[*n].arg = (int)br_ue(b); }
                (*n)++;
            }
        }
        if (pps->weighted_pred && sh->type == P264_SLICE_P) { ERR(p, "weighted prediction unsupported (decoder/decoder.c:259-262)"); return -1; }
        if (pps->weighted_bipred == 1 && sh->type == P264_SLICE_B) { ERR(p, "explicit weighted bi-prediction unsupported (decoder/decoder.c:259-262)"); return -1; }
    }
    if (nal_ref_idc != 0) {
        if (nal_type == NAL_SLICE_IDR) { sh->no_output_of_prior = (int)br_u1(b); sh->long_term_flag = (int)br_u1(b); }
        else if (br_u1(b)) {
            /* dec_ref_pic_marking with adaptive_ref_pic_marking_mode_flag (H.264 7.3.3.3; the reference parses the
             * commands and then ignores them, decoder/decoder.c:264-297, decoder/lists.c:183-187) */
            sh->adaptive_marking = 1;
            for (;;) {
                const unsigned op = br_ue(b);
                if (op == 0) break;
                if (op > 6 || sh->n_mmco >= 33 || br_overrun(b)) { ERR(p, "bad memory_management_control_operation %u", op); return -1; }
                /* operands are bounded by the syntax (7.4.3.3): picture-number differences below MaxFrameNum, long-term
                 * indices at most num_ref_frames - anything else is a broken stream, not something to compute with */
                unsigned a = 0, c = 0;
                const unsigned max_fn_u = 1u << sps->log2_max_frame_num, max_lt = (unsigned)(sps->num_ref_frames > 0 ? sps->num_ref_frames : 1);
                if (op == 1 || op == 3) { a = br_ue(b); if (a >= max_fn_u) { ERR(p, "difference_of_pic_nums_minus1 %u out of range", a); return -1; } }
                if (op == 2) { a = br_ue(b); if (a >= 2 * max_lt) { ERR(p, "long_term_pic_num %u out of range", a); return -1; } }
                if (op == 3 || op == 6) { c = br_ue(b); if (c >= max_lt) { ERR(p, "long_term_frame_idx %u out of range (num_ref_frames %u)", c, max_lt); return -1; } }
                if (op == 4) { a = br_ue(b); if (a > max_lt) { ERR(p, "max_long_term_frame_idx_plus1 %u out of range", a); return -1; } }
                sh->mmco[sh->n_mmco].op = (int)op; sh->mmco[sh->n_mmco].a = (int)a; sh->mmco[sh->n_mmco].b = (int)c;
                sh->n_mmco++;
            }
        }
    }
    if (pps->cabac && sh->type != P264_SLICE_I) { sh->cabac_init_idc = (int)br_ue(b); if (sh->cabac_init_idc > 2) { ERR(p, "cabac_init_idc %d out of range", sh->cabac_init_idc); return -1; } }
    sh->qp = pps->pic_init_qp + br_se(b);
    if (pps->deblock_ctrl) {
        sh->disable_deblock = (int)br_ue(b);
        if (sh->disable_deblock != 1) { sh->alpha_off = br_se(b); sh->beta_off = br_se(b); }
    }
    if (br_overrun(b)) { ERR(p, "slice header overruns the NAL"); return -1; }
    return 0;
}

/* ---------------------------------------------------------------- frame store ----------- */
/* Picture order count of the picture whose first slice header is sh (H.264 8.2.1; frames only: PicOrderCnt =
 * Min(TopFieldOrderCnt, BottomFieldOrderCnt)).  Types 0 and 2; type 1 is only ever met in streams without B slices,
 * where nothing depends on the count.  The state behind it (prevPicOrderCntMsb / Lsb, prevFrameNumOffset) moves on in
 * finish_picture_marking. */
static int poc_msb_of(const p264parse *p, const sps_t *sps, const slice_t *sh, int idr)
{
    const int max_lsb = 1 << sps->log2_max_poc_lsb, prev_msb = idr ? 0 : p->prev_poc_msb, prev_lsb = idr ? 0 : p->prev_poc_lsb;
    if (sh->poc_lsb < prev_lsb && prev_lsb - sh->poc_lsb >= max_lsb / 2) return prev_msb + max_lsb;
    if (sh->poc_lsb > prev_lsb && sh->poc_lsb - prev_lsb > max_lsb / 2) return prev_msb - max_lsb;
    return prev_msb;
}
static int frame_num_offset_of(const p264parse *p, const sps_t *sps, const slice_t *sh, int idr)
{
    if (idr) return 0;
    return p->prev_frame_num > sh->frame_num ? p->frame_num_offset + (1 << sps->log2_max_frame_num) : p->frame_num_offset;
}
static int picture_order_count(const p264parse *p, const slice_t *sh, int idr, int nal_ref_idc)
{
    const sps_t *sps = &p->sps[p->active_sps];
    if (sps->poc_type == 0) {
        const int top = poc_msb_of(p, sps, sh, idr) + sh->poc_lsb, bottom = top + sh->delta_poc_bottom;
        return top < bottom ? top : bottom;
    }
    if (sps->poc_type == 2) return idr ? 0 : 2 * (frame_num_offset_of(p, sps, sh, idr) + sh->frame_num) - (nal_ref_idc == 0);
    return 0;
}

/* Reference list X of the slice (H.264 8.2.4.2, 8.2.4.3).  Initial order - P slices: short-term pictures by descending
 * PicNum (decoder/lists.c:72-143); B slices (the reference stops at decoder/lists.c:136): list 0 the short-term pictures
 * before the current one in output order, nearest first, then those after it, nearest first - list 1 the other way round;
 * both: then the long-term pictures by ascending LongTermPicNum; a list 1 of more than one entry that equals list 0 gets its
 * first two entries swapped.  Then the slice's reordering commands (short-term: idc 0 / 1, long-term: idc 2; the reference
 * ignores them, decoder/lists.c:146-149). */
static int build_list(p264parse *p, const slice_t *sh, int X, int *out)
{
    const sps_t *sps = &p->sps[p->active_sps];
    const int max_fn = 1 << sps->log2_max_frame_num, isB = sh->type == P264_SLICE_B;
    int idx[2][P264HIP_MAX_REFS + 1], n = 0, n_short;
    for (int L = 0; L < (isB ? 2 : 1); L++) {
        n = 0;
        for (int i = 0; i < p->slots; i++) {
            if (!p->dpb[i].used || p->dpb[i].is_long || i == p->cur_slot) continue;
            p->dpb[i].pic_num = p->dpb[i].frame_num > sh->frame_num ? p->dpb[i].frame_num - max_fn : p->dpb[i].frame_num;
            int j = n++;
            if (!isB) while (j > 0 && p->dpb[idx[L][j-1]].pic_num < p->dpb[i].pic_num) { idx[L][j] = idx[L][j-1]; j--; }
            else {
                /* order key: list 0 wants POC below the current one first, descending, then the rest ascending; list 1 the mirror image */
                const int before_i = p->dpb[i].poc < p->cur_poc;
                for (; j > 0; j--) {
                    const dpb_frame_t *o = &p->dpb[idx[L][j-1]];
                    const int before_o = o->poc < p->cur_poc;
                    int i_first;
                    if (before_i != before_o) i_first = L == 0 ? before_i : !before_i;
                    else i_first = before_i ? p->dpb[i].poc > o->poc : p->dpb[i].poc < o->poc;     /* nearest first on either side */
                    if (!i_first) break;
                    idx[L][j] = idx[L][j-1];
                }
            }
            idx[L][j] = i;
        }
        n_short = n;
        for (int i = 0; i < p->slots; i++) {
            if (!p->dpb[i].used || !p->dpb[i].is_long || i == p->cur_slot) continue;
            int j = n++;
            while (j > n_short && p->dpb[idx[L][j-1]].long_idx > p->dpb[i].long_idx) { idx[L][j] = idx[L][j-1]; j--; }
            idx[L][j] = i;
        }
    }
    n_short = 0;
    for (int i = 0; i < p->slots; i++) if (p->dpb[i].used && !p->dpb[i].is_long && i != p->cur_slot) n_short++;
    if (n == 0) { ERR(p, "%s slice without a reference picture", isB ? "B" : "P"); return -1; }
    if (isB && n > 1 && !memcmp(idx[0], idx[1], sizeof(int) * (size_t)n)) { const int t = idx[1][0]; idx[1][0] = idx[1][1]; idx[1][1] = t; }
    const int *ini = idx[isB ? X : 0];
    const int len = X ? sh->num_ref_idx_l1 : sh->num_ref_idx;
    const int n_cmd = X ? sh->n_reorder1 : sh->n_reorder;
    int list[P264HIP_MAX_REFS + 1];
    for (int i = 0; i < len; i++) list[i] = ini[i < n ? i : n - 1];
    int pred = sh->frame_num, at = 0;
    for (int k = 0; k < n_cmd && at < len; k++) {
        const int idc = X ? sh->reorder1[k].idc : sh->reorder[k].idc, arg = X ? sh->reorder1[k].arg : sh->reorder[k].arg;
        int slot = -1;
        if (idc == 2) {                                   /* long_term_pic_num */
            for (int i = n_short; i < n; i++) if (p->dpb[ini[i]].long_idx == arg) slot = ini[i];
        } else {
            int d = arg + 1;
            pred = idc == 0 ? pred - d : pred + d;
            if (pred < 0) pred += max_fn;
            if (pred >= max_fn) pred -= max_fn;
            int want = pred > sh->frame_num ? pred - max_fn : pred;
            for (int i = 0; i < p->slots; i++) if (p->dpb[i].used && !p->dpb[i].is_long && i != p->cur_slot && p->dpb[i].pic_num == want) slot = i;
        }
        if (slot < 0) { ERR(p, "reordering names a picture that is not in the frame store"); return -1; }
        for (int i = len; i > at; i--) list[i] = list[i-1];
        list[at++] = slot;
        int w = at;
        for (int r = at; r <= len; r++) if (list[r] != slot) list[w++] = list[r];
    }
    for (int i = 0; i < len; i++) out[i] = list[i];
    return len;
}

/* Implicit bi-prediction weights of the picture (H.264 8.4.2.3.1 with weighted_bipred_idc 2; the reference computes the same
 * numbers in p264_macroblock_bipred_init, core/macroblock.c:1400-1430): weight of the list-0 prediction for every pair of
 * reference indices; 32 (the plain average) where the distances do not give a weight in -64 .. 128. */
static void implicit_weights(p264parse *p)
{
    for (int r0 = 0; r0 < P264HIP_MAX_REFS; r0++)
        for (int r1 = 0; r1 < P264HIP_MAX_REFS; r1++) {
            int w0 = 32;
            if (p->weighted_bipred && r0 < p->n_list0 && r1 < p->n_list1) {
                const dpb_frame_t *f0 = &p->dpb[p->list0[r0]], *f1 = &p->dpb[p->list1[r1]];
                const int td = clip3i(f1->poc - f0->poc, -128, 127), tb = clip3i(p->cur_poc - f0->poc, -128, 127);
                if (td != 0 && !f0->is_long && !f1->is_long) {
                    const int tx = (16384 + (td < 0 ? -td : td) / 2) / td;
                    const int dsf = clip3i((tb * tx + 32) >> 6, -1024, 1023) >> 2;
                    if (dsf >= -64 && dsf <= 128) w0 = 64 - dsf;
                }
            }
            p->bipred_weight[r0 * P264HIP_MAX_REFS + r1] = (int16_t)w0;
        }
}

/* Reference picture marking (H.264 8.2.5; the reference implements the sliding window only, decoder/lists.c:152-228) and
 * the choice of the next slot. */
static void finish_picture_marking(p264parse *p)
{
    const sps_t *sps = &p->sps[p->active_sps];
    int max_fn = 1 << sps->log2_max_frame_num;
    dpb_frame_t *cur = &p->dpb[p->cur_slot];
    int cur_long = 0, cur_long_idx = 0, had_mmco5 = 0;
    if (p->pic_is_idr) {
        for (int i = 0; i < p->slots; i++) if (i != p->cur_slot) p->dpb[i].used = 0;
        if (p->sh0.long_term_flag) { cur_long = 1; cur_long_idx = 0; }
    } else if (p->pic_ref_idc && p->sh0.adaptive_marking) {
        /* 8.2.5.4: the commands in order; PicNum relative to the current picture's frame_num */
        for (int k = 0; k < p->sh0.n_mmco; k++) {
            const int op = p->sh0.mmco[k].op, a = p->sh0.mmco[k].a, b = p->sh0.mmco[k].b;
            if (op == 1 || op == 3) {
                const int want = p->sh0.frame_num - (a + 1);
                for (int i = 0; i < p->slots; i++) {
                    dpb_frame_t *f = &p->dpb[i];
                    if (!f->used || f->is_long || i == p->cur_slot) continue;
                    const int num = f->frame_num > p->sh0.frame_num ? f->frame_num - max_fn : f->frame_num;
                    if (num != want) continue;
                    if (op == 1) f->used = 0;
                    else {                                  /* 3: the index is taken away from whoever holds it, then assigned */
                        for (int j = 0; j < p->slots; j++) if (j != i && p->dpb[j].used && p->dpb[j].is_long && p->dpb[j].long_idx == b) p->dpb[j].used = 0;
                        f->is_long = 1; f->long_idx = b;
                    }
                }
            } else if (op == 2) {
                for (int i = 0; i < p->slots; i++) if (p->dpb[i].used && p->dpb[i].is_long && p->dpb[i].long_idx == a && i != p->cur_slot) p->dpb[i].used = 0;
            } else if (op == 4) {
                for (int i = 0; i < p->slots; i++) if (p->dpb[i].used && p->dpb[i].is_long && p->dpb[i].long_idx >= a && i != p->cur_slot) p->dpb[i].used = 0;
            } else if (op == 5) {
                for (int i = 0; i < p->slots; i++) if (i != p->cur_slot) p->dpb[i].used = 0;
                had_mmco5 = 1;
            } else if (op == 6) {
                for (int j = 0; j < p->slots; j++) if (j != p->cur_slot && p->dpb[j].used && p->dpb[j].is_long && p->dpb[j].long_idx == b) p->dpb[j].used = 0;
                cur_long = 1; cur_long_idx = b;
            }
        }
    } else if (p->pic_ref_idc) {
        /* 8.2.5.3 sliding window: when short-term + long-term pictures fill num_ref_frames, the oldest short-term one goes */
        int cnt = 0, oldest = -1, oldest_num = 0;
        for (int i = 0; i < p->slots; i++) {
            if (!p->dpb[i].used || i == p->cur_slot) continue;
            cnt++;
            if (p->dpb[i].is_long) continue;
            int num = p->dpb[i].frame_num > p->sh0.frame_num ? p->dpb[i].frame_num - max_fn : p->dpb[i].frame_num;
            if (oldest < 0 || num < oldest_num) { oldest = i; oldest_num = num; }
        }
        int cap = sps->num_ref_frames > 0 ? sps->num_ref_frames : 1;
        if (cnt >= cap && oldest >= 0) p->dpb[oldest].used = 0;
    }
    /* after memory_management_control_operation 5 the picture is inferred to have had frame_num 0 (H.264 7.4.3, 8.2.1): the
     * pictures that follow compute their PicNums against that */
    if (p->pic_ref_idc) { cur->used = 1; cur->frame_num = had_mmco5 ? 0 : p->sh0.frame_num; cur->is_long = cur_long; cur->long_idx = cur_long_idx; }
    /* picture order count: what the pictures behind this one measure theirs against (8.2.1; after operation 5 the picture
     * counts as POC 0 - frames, bottom not below top) */
    cur->poc = had_mmco5 ? 0 : p->cur_poc; cur->uid = p->cur_uid;
    if (sps->poc_type == 0 && p->pic_ref_idc) {
        p->prev_poc_msb = had_mmco5 ? 0 : poc_msb_of(p, sps, &p->sh0, p->pic_is_idr);
        p->prev_poc_lsb = had_mmco5 ? 0 : p->sh0.poc_lsb;
    }
    if (sps->poc_type == 2) { p->frame_num_offset = had_mmco5 ? 0 : frame_num_offset_of(p, sps, &p->sh0, p->pic_is_idr); p->prev_frame_num = had_mmco5 ? 0 : p->sh0.frame_num; }
    /* the motion of a reference picture is what the direct prediction of later B pictures reads (8.4.1.2) */
    if (p->pic_ref_idc && p->has_col) {
        const picbuf_t *q = &p->buf[p->cur];
        int16_t *cm = p->col_mv[p->cur_slot]; int8_t *cr = p->col_ref[p->cur_slot]; int32_t *cu = p->col_uid[p->cur_slot];
        const int isB = p->sh0.type == P264_SLICE_B;
        for (int i = 0; i < p->n_mb * 4; i++) {
            const int r0 = q->ref[i], r1 = isB ? q->ref1[i] : -1;
            const int mbi = i >> 2, q8 = i & 3, b0 = (q8 >> 1) * 8 + (q8 & 1) * 2;
            const int16_t *src = r0 >= 0 || r1 < 0 ? q->mv : q->mv1;                          /* the list-0 motion if there is one, else list 1 */
            const int r = r0 >= 0 ? r0 : r1;
            cr[i] = (int8_t)r;
            cu[i] = r < 0 ? -1 : (int32_t)p->dpb[r0 >= 0 ? p->list0[r0 < p->n_list0 ? r0 : 0] : p->list1[r1 < p->n_list1 ? r1 : 0]].uid;
            for (int k = 0; k < 4; k++) {
                const int blk = b0 + (k >> 1) * 4 + (k & 1);
                cm[(mbi * 16 + blk) * 2] = r < 0 ? 0 : src[(mbi * 16 + blk) * 2]; cm[(mbi * 16 + blk) * 2 + 1] = r < 0 ? 0 : src[(mbi * 16 + blk) * 2 + 1];
            }
        }
    }
    /* next picture goes into a slot that holds no reference */
    int next = -1;
    for (int i = 0; i < p->slots; i++) if (!p->dpb[i].used) { next = i; break; }
    if (next < 0) {                               /* a stream that keeps more pictures than num_ref_frames: drop the oldest short-term one */
        int oldest = -1, oldest_num = 0;
        for (int i = 0; i < p->slots; i++) {
            if (p->dpb[i].is_long) continue;
            int num = p->dpb[i].frame_num > p->sh0.frame_num ? p->dpb[i].frame_num - max_fn : p->dpb[i].frame_num;
            if (oldest < 0 || num < oldest_num) { oldest = i; oldest_num = num; }
        }
        next = oldest >= 0 ? oldest : p->cur_slot;
        p->dpb[next].used = 0;
    }
    p->cur_slot = next;
}

/* ---------------------------------------------------------------- neighbours ------------ */
static inline int mb_avail(const p264parse *p, int mbx, int mby)
{
    if (mbx < 0 || mby < 0 || mbx >= p->mb_w || mby >= p->mb_h) return 0;
    int i = mby * p->mb_w + mbx;
    return i < p->mbi && p->slice_of[i] == (uint16_t)p->slice_no;
}

typedef struct { int ref, mvx, mvy; } nbmv_t;      /* ref: -2 unavailable, -1 intra */

/* motion data of the 4x4 block at picture position (x4,y4), as a predictor for the current MB */
static nbmv_t nb_motion_l(const p264parse *p, int x4, int y4, int list)
{
    nbmv_t r = { -2, 0, 0 };
    /* where the block lies relative to the current macroblock decides everything: inside it - decoded so far or not; in the
     * left / top / top-right / top-left neighbour - that macroblock's availability (begin_mb); anywhere else - not decoded yet */
    const int dx = x4 - p->mbx * 4, dy = y4 - p->mby * 4, sub = (y4 & 3) * 4 + (x4 & 3);
    int i;
    if (dy >= 0) {
        if (dx >= 0) { if (dx >= 4 || dy >= 4 || !(((list ? p->mv_done1 : p->mv_done) >> sub) & 1)) return r; i = p->mbi; }
        else { if (dy >= 4 || !(p->cur_avail & P264_AVAIL_LEFT)) return r; i = p->mbi - 1; }
    } else {
        if (dx < 0)      { if (!(p->cur_avail & P264_AVAIL_TOPLEFT)) return r;  i = p->mbi - p->mb_w - 1; }
        else if (dx < 4) { if (!(p->cur_avail & P264_AVAIL_TOP)) return r;      i = p->mbi - p->mb_w; }
        else             { if (dx >= 8 || !(p->cur_avail & P264_AVAIL_TOPRIGHT)) return r; i = p->mbi - p->mb_w + 1; }
    }
    const picbuf_t *q = &p->buf[p->cur];
    const int8_t *ref = list ? q->ref1 : q->ref; const int16_t *mv = list ? q->mv1 : q->mv;
    r.ref = ref[i * 4 + ((y4 & 2) | ((x4 >> 1) & 1))];     /* -1: intra, or (B pictures) this list is not used there */
    r.mvx = mv[(i * 16 + sub) * 2]; r.mvy = mv[(i * 16 + sub) * 2 + 1];
    return r;
}

/* H.264 8.4.1.3 (core/macroblock.c:87-175).  (bx,by,bw) in 4x4 units inside the MB;
 * dir: 0 none, 1 = 16x8 upper, 2 = 16x8 lower, 3 = 8x16 left, 4 = 8x16 right. */
/* the block `sub` of macroblock i (which exists and is decoded) as a predictor */
static inline nbmv_t nb_of_mb(const p264parse *p, int i, int sub, int list)
{
    const picbuf_t *q = &p->buf[p->cur];
    const int8_t *ref = list ? q->ref1 : q->ref; const int16_t *mv = list ? q->mv1 : q->mv;
    nbmv_t r;
    r.ref = ref[i * 4 + (((sub >> 2) & 2) | ((sub >> 1) & 1))];
    r.mvx = mv[(i * 16 + sub) * 2]; r.mvy = mv[(i * 16 + sub) * 2 + 1];
    return r;
}
static void predict_mv_l(const p264parse *p, int bx, int by, int bw, int ref, int dir, int *px, int *py, int list)
{
    int x0 = p->mbx * 4 + bx, y0 = p->mby * 4 + by;
    nbmv_t a, b, c;
    if (bw == 4 && by == 0) {
        /* the whole macroblock or its upper half (P_L0_16x16, P_SKIP, B 16x16, spatial direct, 16x8 upper - most calls): the neighbours are block 3 of the
         * macroblock to the left, block 12 of the one above, block 12 of the one above to the right or else block 15 of the
         * one above to the left; nothing but the availability flags to ask */
        const nbmv_t none = { -2, 0, 0 };
        const unsigned av = p->cur_avail;
        a = (av & P264_AVAIL_LEFT) ? nb_of_mb(p, p->mbi - 1, 3, list) : none;
        b = (av & P264_AVAIL_TOP) ? nb_of_mb(p, p->mbi - p->mb_w, 12, list) : none;
        c = (av & P264_AVAIL_TOPRIGHT) ? nb_of_mb(p, p->mbi - p->mb_w + 1, 12, list)
          : (av & P264_AVAIL_TOPLEFT) ? nb_of_mb(p, p->mbi - p->mb_w - 1, 15, list) : none;
    } else {
        a = nb_motion_l(p, x0 - 1, y0, list); b = nb_motion_l(p, x0, y0 - 1, list); c = nb_motion_l(p, x0 + bw, y0 - 1, list);
        if (c.ref == -2) c = nb_motion_l(p, x0 - 1, y0 - 1, list);
    }
    if (dir == 1 && b.ref == ref) { *px = b.mvx; *py = b.mvy; return; }
    if (dir == 2 && a.ref == ref) { *px = a.mvx; *py = a.mvy; return; }
    if (dir == 3 && a.ref == ref) { *px = a.mvx; *py = a.mvy; return; }
    if (dir == 4 && c.ref == ref) { *px = c.mvx; *py = c.mvy; return; }
    int hits = (a.ref == ref) + (b.ref == ref) + (c.ref == ref);
    if (hits == 1) {
        const nbmv_t *s = a.ref == ref ? &a : b.ref == ref ? &b : &c;
        *px = s->mvx; *py = s->mvy; return;
    }
    if (hits == 0 && b.ref == -2 && c.ref == -2 && a.ref != -2) { *px = a.mvx; *py = a.mvy; return; }
    *px = median3(a.mvx, b.mvx, c.mvx); *py = median3(a.mvy, b.mvy, c.mvy);
}
static void predict_mv(const p264parse *p, int bx, int by, int bw, int ref, int dir, int *px, int *py) { predict_mv_l(p, bx, by, bw, ref, dir, px, py, 0); }

static void set_motion_l(p264parse *p, int bx, int by, int bw, int bh, int mvx, int mvy, int list)
{
    picbuf_t *q = &p->buf[p->cur];
    int16_t *mv = list ? q->mv1 : q->mv;
    if (bw == 4 && bh == 4) {                                   /* the whole macroblock: sixteen equal vectors, eight 8-byte stores */
        const uint32_t one = (uint32_t)(uint16_t)mvx | (uint32_t)(uint16_t)mvy << 16;
        const uint64_t two = (uint64_t)one | (uint64_t)one << 32;
        int16_t *d = mv + (size_t)p->mbi * 32;
        for (int k = 0; k < 8; k++) memcpy(d + 4 * k, &two, 8);
        if (list) p->mv_done1 = 0xffffu; else p->mv_done = 0xffffu;
        return;
    }
    for (int y = by; y < by + bh; y++)
        for (int x = bx; x < bx + bw; x++) {
            mv[(p->mbi * 16 + y * 4 + x) * 2] = (int16_t)mvx;
            mv[(p->mbi * 16 + y * 4 + x) * 2 + 1] = (int16_t)mvy;
            if (list) p->mv_done1 |= 1u << (y * 4 + x); else p->mv_done |= 1u << (y * 4 + x);
        }
}
static void set_motion(p264parse *p, int bx, int by, int bw, int bh, int mvx, int mvy) { set_motion_l(p, bx, by, bw, bh, mvx, mvy, 0); }

/* Intra4x4PredMode predictor (H.264 8.3.1.1; core/macroblock.c:40-51) */
static int predict_i4mode(const p264parse *p, int blk)
{
    const picbuf_t *q = &p->buf[p->cur];
    int x = blk_x[blk], y = blk_y[blk], ma, mb;
    if (x > 0) ma = q->i4[p->mbi * 16 + blk_of_xy[y][x-1]];
    else if (p->cur_avail & P264_AVAIL_LEFT)
        ma = q->mb[p->mbi - 1].mb_type == P264_MB_I4x4 ? q->i4[(p->mbi - 1) * 16 + blk_of_xy[y][3]] : 2;
    else ma = -1;
    if (y > 0) mb = q->i4[p->mbi * 16 + blk_of_xy[y-1][x]];
    else if (p->cur_avail & P264_AVAIL_TOP)
        mb = q->mb[p->mbi - p->mb_w].mb_type == P264_MB_I4x4 ? q->i4[(p->mbi - p->mb_w) * 16 + blk_of_xy[3][x]] : 2;
    else mb = -1;
    int m = ma < mb ? ma : mb;
    return m < 0 ? 2 : m;
}

/* ---------------------------------------------------------------- macroblock layer ------ */
typedef struct {
    int16_t dc_luma[16], dc_chroma[16], blk[24][16];
    uint32_t mask;
} mbcoef_t;

static int coef_reserve(picbuf_t *q, size_t more)
{
    if (q->coef_n + more <= q->coef_cap) return 0;
    size_t cap = q->coef_cap * 2 + more;
    int16_t *n = (int16_t *)q->alloc(cap * 16 * sizeof(int16_t));
    if (!n) return -1;
    memcpy(n, q->coef, q->coef_n * 16 * sizeof(int16_t));
    if (q->coef_own) q->release(q->coef);               /* (else: a section of q->block, which stays) */
    q->coef = n; q->coef_cap = cap; q->coef_own = 1;
    return 0;
}

#include "parser_cabac.h"

/* residual( ) - decoder/macroblock.c:410-486 */
/* total_coeff predictor nC (H.264 9.2.1; core/macroblock.c:53-65): the mean of the counts of the blocks to the left and above.
 * The coefficient counts around and inside the macroblock on an 8-wide grid (left neighbour of a block one to the left, upper one
 * eight back; 0x80 = no such neighbour): luma block (x, y) at 8 (1 + y) + 1 + x, Cb (x, y) at 8 (6 + y) + 1 + x, Cr at
 * 8 (6 + y) + 5 + x.  nC of 9.2.1 is then two loads and a rounded mean that an absent neighbour falls out of by itself
 * (predict_nc: two table look-ups and four branches per block). */
static const uint8_t nc_pos[24] = { 9, 10, 17, 18, 11, 12, 19, 20, 25, 26, 33, 34, 27, 28, 35, 36,  49, 50, 57, 58,  53, 54, 61, 62 };
static inline int nc_of(const uint8_t *nc, int at)
{
    int r = nc[at - 1] + nc[at - 8];
    if (r < 0x80) r = (r + 1) >> 1;
    return r & 0x7f;
}
static int parse_residual(p264parse *p, bitrd_t *b, p264hip_mb_t *m, mbcoef_t *cf)
{
    uint8_t *nnz = p->nnz + (size_t)p->mbi * 24;
    int cbp_l = m->cbp & 15, cbp_c = m->cbp >> 4, tc;
    uint8_t nc[64];
    {
        const uint8_t *left = nnz - 24, *top = nnz - 24 * (size_t)p->mb_w;
        if (p->cur_avail & P264_AVAIL_LEFT) {
            nc[8] = left[5]; nc[16] = left[7]; nc[24] = left[13]; nc[32] = left[15];
            nc[48] = left[17]; nc[56] = left[19]; nc[52] = left[21]; nc[60] = left[23];
        } else nc[8] = nc[16] = nc[24] = nc[32] = nc[48] = nc[56] = nc[52] = nc[60] = 0x80;
        if (p->cur_avail & P264_AVAIL_TOP) {
            nc[1] = top[10]; nc[2] = top[11]; nc[3] = top[14]; nc[4] = top[15];
            nc[41] = top[18]; nc[42] = top[19]; nc[45] = top[22]; nc[46] = top[23];
        } else nc[1] = nc[2] = nc[3] = nc[4] = nc[41] = nc[42] = nc[45] = nc[46] = 0x80;
    }
    if (m->mb_type == P264_MB_I16x16) {
        if ((tc = cavlc_read_block(b, nc_of(nc, nc_pos[0]), 16, cf->dc_luma)) < 0) return -1;
        if (tc) cf->mask |= P264_COEF_LUMA_DC;
    }
    int maxc = m->mb_type == P264_MB_I16x16 ? 15 : 16;
    for (int i = 0; i < 16; i++) {
        const int at = nc_pos[i];
        nnz[i] = 0; nc[at] = 0;
        if (!(cbp_l & (1 << (i >> 2)))) continue;
        if ((tc = cavlc_read_block(b, nc_of(nc, at), maxc, cf->blk[i])) < 0) return -1;
        nnz[i] = (uint8_t)tc; nc[at] = (uint8_t)tc;
        if (tc) cf->mask |= 1u << i;
    }
    if (cbp_c) {
        memset(cf->dc_chroma, 0, sizeof cf->dc_chroma);
        int t0, t1;
        if ((t0 = cavlc_read_block(b, -1, 4, cf->dc_chroma)) < 0) return -1;
        if ((t1 = cavlc_read_block(b, -1, 4, cf->dc_chroma + 4)) < 0) return -1;
        if (t0 | t1) cf->mask |= P264_COEF_CHROMA_DC;
    }
    for (int i = 16; i < 24; i++) {
        const int at = nc_pos[i];
        nnz[i] = 0; nc[at] = 0;
        if (!(cbp_c & 2)) continue;
        if ((tc = cavlc_read_block(b, nc_of(nc, at), 15, cf->blk[i])) < 0) return -1;
        nnz[i] = (uint8_t)tc; nc[at] = (uint8_t)tc;
        if (tc) cf->mask |= 1u << i;
    }
    return 0;
}

static int rd_residual(p264parse *p, bitrd_t *b, p264hip_mb_t *m, mbcoef_t *cf)
{
    return p->cabac_on ? parse_residual_cabac(p, m, cf) : parse_residual(p, b, m, cf);
}

static int store_coefs(p264parse *p, p264hip_mb_t *m, const mbcoef_t *cf)
{
    picbuf_t *q = &p->buf[p->cur];
    m->coef_mask = cf->mask;
    m->coef_index = (uint32_t)q->coef_n;
    if (!cf->mask) return 0;
    if (coef_reserve(q, 26) < 0) return -1;
    int16_t *dst = q->coef + q->coef_n * 16;
    if (cf->mask & P264_COEF_LUMA_DC)   { memcpy(dst, cf->dc_luma, 32); dst += 16; }
    if (cf->mask & P264_COEF_CHROMA_DC) { memcpy(dst, cf->dc_chroma, 32); dst += 16; }
    for (uint32_t left = cf->mask & 0xffffffu; left; left &= left - 1) { memcpy(dst, cf->blk[__builtin_ctz(left)], 32); dst += 16; }
    q->coef_n = (size_t)(dst - q->coef) / 16;
    return 0;
}

/* the four neighbours (mb_avail, spelled out: they all lie in front of this macroblock, so only the picture's borders and the
 * slice they belong to are left to ask) */
static inline int mb_neighbours(p264parse *p)
{
    int a = 0;
    const int w = p->mb_w, i = p->mbi, x = p->mbx;
    const uint16_t sn = (uint16_t)p->slice_no, *so = p->slice_of;
    if (x > 0 && so[i - 1] == sn) a |= P264_AVAIL_LEFT;
    if (p->mby > 0) {
        if (so[i - w] == sn) a |= P264_AVAIL_TOP;
        if (x + 1 < w && so[i - w + 1] == sn) a |= P264_AVAIL_TOPRIGHT;
        if (x > 0 && so[i - w - 1] == sn) a |= P264_AVAIL_TOPLEFT;
    }
    p->cur_avail = a;
    return a;
}
static void begin_mb(p264parse *p, p264hip_mb_t *m)
{
    memset(m, 0, sizeof *m);
    p->mv_done = 0; p->mv_done1 = 0;
    if (p->cabac_on) { p->cinfo[p->mbi] = 0; memset(p->mvd_abs[0] + p->mbi * 32, 0, 32); memset(p->mvd_abs[1] + p->mbi * 32, 0, 32); }
    const int a = mb_neighbours(p);
    m->avail = (uint8_t)a;
    int e = 0;
    if (p->sh.disable_deblock != 1) {
        e = P264_EDGE_INNER;
        if (p->mbx > 0 && (p->sh.disable_deblock == 0 || (a & P264_AVAIL_LEFT))) e |= P264_EDGE_LEFT;
        if (p->mby > 0 && (p->sh.disable_deblock == 0 || (a & P264_AVAIL_TOP)))  e |= P264_EDGE_TOP;
    }
    m->edges = (uint8_t)e;
}

/* QP bookkeeping of core/macroblock.c:1247-1252 (or the conformant chain in strict mode) */
static void finish_mb_qp(p264parse *p, p264hip_mb_t *m, int has_residual_syntax, int qp)
{
    if (p->strict_qp) {
        if (!has_residual_syntax) qp = p->qp_pred;
        p->qp_pred = qp;
    } else {
        if (m->mb_type != P264_MB_I16x16 && m->cbp == 0) qp = p->last_qp;
        p->last_qp = qp;
    }
    m->qp = (uint8_t)clip3i(qp, 0, 51);
}

/* decoder/macroblock.c:895-934 */
static void decode_pskip(p264parse *p)
{
    picbuf_t *q = &p->buf[p->cur];
    p264hip_mb_t *m = &q->mb[p->mbi];
    begin_mb(p, m);
    m->mb_type = P264_MB_P_SKIP;
    memset(p->nnz + (size_t)p->mbi * 24, 0, 24);
    memset(q->ref + p->mbi * 4, 0, 4);
    memset(q->i4 + p->mbi * 16, 2, 16);
    int mvx = 0, mvy = 0;
    /* (8.4.1.1: the zero vector without a left or an upper neighbour or next to one at rest on reference 0; else the 16x16 prediction) */
    if ((p->cur_avail & (P264_AVAIL_LEFT | P264_AVAIL_TOP)) == (P264_AVAIL_LEFT | P264_AVAIL_TOP)) {
        const nbmv_t a = nb_of_mb(p, p->mbi - 1, 3, 0), b = nb_of_mb(p, p->mbi - p->mb_w, 12, 0);
        if (!((a.ref == 0 && a.mvx == 0 && a.mvy == 0) || (b.ref == 0 && b.mvx == 0 && b.mvy == 0)))
            predict_mv(p, 0, 0, 4, 0, 0, &mvx, &mvy);
    }
    set_motion(p, 0, 0, 4, 4, mvx, mvy);
    m->coef_index = (uint32_t)q->coef_n;
    finish_mb_qp(p, m, 0, p->sh.qp);
    p->last_dqp = 0;
    if (p->cabac_on) p->cinfo[p->mbi] |= CI_SKIP;
}

/* t: mb_type as read (P slices), intra_t >= 0: the macroblock is intra with that I-slice type (I slices; P / B slices after
 * their offset of 5 / 23) */
static int parse_mb_t(p264parse *p, bitrd_t *b, unsigned t, int intra_t)
{
    picbuf_t *q = &p->buf[p->cur];
    p264hip_mb_t *m = &q->mb[p->mbi];
    mbcoef_t cf; cf.mask = 0;
    begin_mb(p, m);
    int8_t *ref = q->ref + p->mbi * 4;
    uint8_t *i4 = q->i4 + p->mbi * 16;

    if (intra_t >= 0) {
        /* ---- intra (decoder/macroblock.c:117-139, 265-301) ---- */
        if (intra_t > 25) { ERR(p, "invalid mb type %d", intra_t); return -1; }
        if (intra_t == 25) { ERR(p, "unsupport i_pcm mb"); return -1; }
        memset(ref, -1, 4);
        memset(q->mv + p->mbi * 32, 0, 64);
        if (q->mv1) { memset(q->ref1 + p->mbi * 4, -1, 4); memset(q->mv1 + p->mbi * 32, 0, 64); }
        if (intra_t == 0) {
            m->mb_type = P264_MB_I4x4;
            for (int i = 0; i < 16; i++) i4[i] = (uint8_t)rd_intra4x4_mode(p, b, predict_i4mode(p, i));
        } else {
            m->mb_type = P264_MB_I16x16;
            m->intra_modes = (uint8_t)((intra_t - 1) & 3);
            m->cbp = (uint8_t)((((intra_t - 1) >> 2) % 3) << 4 | (intra_t > 12 ? 15 : 0));
            memset(i4, 2, 16);
        }
        unsigned cm = rd_chroma_pred_mode(p, b);
        if (cm > 3) { ERR(p, "invalid intra chroma pred mode %u", cm); return -1; }
        m->intra_modes |= (uint8_t)(cm << 4);
    } else {
        /* ---- inter (decoder/macroblock.c:140-167, 304-408) ---- */
        memset(i4, 2, 16);
        int nref = p->sh.num_ref_idx;
        if (t <= 2) {
            m->mb_type = P264_MB_P_L0;
            static const int8_t geo[3][2][4] = {   /* x, y, w, h in 4x4 units */
                { {0,0,4,4}, {0,0,0,0} }, { {0,0,4,2}, {0,2,4,2} }, { {0,0,2,4}, {2,0,2,4} } };
            int nparts = t == 0 ? 1 : 2, r[2] = { 0, 0 };
            for (int k = 0; k < nparts; k++) {
                r[k] = rd_ref_idx(p, b, 0, geo[t][k][0], geo[t][k][1], nref);
                if (r[k] < 0 || r[k] >= nref) { ERR(p, "ref_idx out of range"); return -1; }
                for (int y = geo[t][k][1] >> 1; y < (geo[t][k][1] + geo[t][k][3]) >> 1; y++)      /* (at once: the next partition's context looks at it) */
                    for (int x = geo[t][k][0] >> 1; x < (geo[t][k][0] + geo[t][k][2]) >> 1; x++) ref[y * 2 + x] = (int8_t)r[k];
            }
            for (int k = 0; k < nparts; k++) {
                int dx, dy, px, py;
                if (rd_mvd(p, b, 0, geo[t][k][0], geo[t][k][1], geo[t][k][2], geo[t][k][3], &dx, &dy) < 0) { ERR(p, "mvd out of range"); return -1; }
                int dir = t == 0 ? 0 : t == 1 ? 1 + k : 3 + k;
                predict_mv(p, geo[t][k][0], geo[t][k][1], geo[t][k][2], r[k], dir, &px, &py);
                set_motion(p, geo[t][k][0], geo[t][k][1], geo[t][k][2], geo[t][k][3], px + dx, py + dy);
            }
        } else {
            m->mb_type = P264_MB_P_8x8;
            int sub[4];
            for (int k = 0; k < 4; k++) { sub[k] = rd_sub_mb_type(p, b); if (sub[k] > 3) { ERR(p, "invalid i_sub_partition"); return -1; } }
            memset(ref, 0, 4);
            for (int k = 0; k < 4; k++) {
                int r = 0;
                if (nref > 1 && t == 3) { r = rd_ref_idx(p, b, 0, (k & 1) * 2, (k >> 1) * 2, nref); if (r < 0 || r >= nref) { ERR(p, "ref_idx out of range"); return -1; } }
                ref[k] = (int8_t)r;
            }
            for (int k = 0; k < 4; k++) {
                int ox = (k & 1) * 2, oy = (k >> 1) * 2;
                int sw = (sub[k] == 0 || sub[k] == 1) ? 2 : 1, shh = (sub[k] == 0 || sub[k] == 2) ? 2 : 1;
                for (int sy = 0; sy < 2; sy += shh)
                    for (int sx = 0; sx < 2; sx += sw) {
                        int dx, dy, px, py;
                        if (rd_mvd(p, b, 0, ox + sx, oy + sy, sw, shh, &dx, &dy) < 0) { ERR(p, "mvd out of range"); return -1; }
                        predict_mv(p, ox + sx, oy + sy, sw, ref[k], 0, &px, &py);
                        set_motion(p, ox + sx, oy + sy, sw, shh, px + dx, py + dy);
                    }
            }
        }
    }

    /* ---- coded_block_pattern, mb_qp_delta, residual (decoder/macroblock.c:540-587) ---- */
    if (m->mb_type != P264_MB_I16x16) {
        const int c = rd_cbp(p, b, m->mb_type == P264_MB_I4x4);
        if (c < 0) { ERR(p, "invalid cbp"); return -1; }
        m->cbp = (uint8_t)c;
    }
    int qp = p->sh.qp, has_res = (m->cbp != 0 || m->mb_type == P264_MB_I16x16);
    if (has_res) {
        int dqp = rd_mb_qp_delta(p, b);
        if (dqp < -52 || dqp > 52) { ERR(p, "mb_qp_delta out of range"); return -1; }
        if (p->strict_qp) qp = (p->qp_pred + dqp + 52) % 52;
        else qp = p->sh.qp + dqp;                 /* delta is NOT accumulated: decoder/macroblock.c:568 */
        if (rd_residual(p, b, m, &cf) < 0) { ERR(p, "read residual data failed"); return -1; }
    } else { memset(p->nnz + (size_t)p->mbi * 24, 0, 24); p->last_dqp = 0; }
    if (store_coefs(p, m, &cf) < 0) return -1;
    finish_mb_qp(p, m, has_res, qp);
    if (br_overrun(b)) { ERR(p, "macroblock overruns the slice data"); return -1; }
    return 0;
}

/* ---------------------------------------------------------------- B macroblocks ---------- */
/* The reference has none of this (decoder/macroblock.c:168-171 rejects B macroblock types; its encoder-side helpers
 * core/macroblock.c:254-429 are not reachable from the decoder): H.264 7.3.5, 7.4.5 (tables 7-14, 7-18), 8.4.1.2. */

/* Direct prediction of the current macroblock (B_Skip, B_Direct_16x16, and the direct 8x8 quadrants of B_8x8): reference
 * indices per 8x8 quadrant and vectors per 4x4 block for both lists, into dr[2][4] / dm[2][16][2].  Nothing is stored:
 * the caller copies the quadrants that are direct. */
typedef struct { int8_t ref[2][4]; int16_t mv[2][16][2]; } direct_t;

static int min_positive(int a, int b) { return (a >= 0 && b >= 0) ? (a < b ? a : b) : (a > b ? a : b); }

static void direct_spatial(const p264parse *p, direct_t *d)
{   /* 8.4.1.2.2: the reference indices from the neighbours A, B, C of the MACROBLOCK, the vectors from the ordinary 16x16
     * prediction with them, zero where the co-located block does not move */
    const int x0 = p->mbx * 4, y0 = p->mby * 4;
    int ref[2], mv[2][2] = { { 0, 0 }, { 0, 0 } };
    for (int l = 0; l < 2; l++) {
        nbmv_t a = nb_motion_l(p, x0 - 1, y0, l), b = nb_motion_l(p, x0, y0 - 1, l), c = nb_motion_l(p, x0 + 4, y0 - 1, l);
        if (c.ref == -2) c = nb_motion_l(p, x0 - 1, y0 - 1, l);
        ref[l] = min_positive(a.ref < 0 ? -1 : a.ref, min_positive(b.ref < 0 ? -1 : b.ref, c.ref < 0 ? -1 : c.ref));
    }
    const int zero_pred = ref[0] < 0 && ref[1] < 0;
    if (zero_pred) ref[0] = ref[1] = 0;
    else for (int l = 0; l < 2; l++) if (ref[l] >= 0) predict_mv_l(p, 0, 0, 4, ref[l], 0, &mv[l][0], &mv[l][1], l);
    /* colZeroFlag: RefPicList1[0] is a short-term picture and the co-located block used reference index 0 with a vector
     * inside +-1 (direct_8x8_inference: the corner block of the quadrant speaks for it) */
    const int col_slot = p->list1[0];
    const int col_short = !p->dpb[col_slot].is_long;
    const int8_t *cr = p->col_ref[col_slot] + p->mbi * 4; const int16_t *cm = p->col_mv[col_slot] + p->mbi * 32;
    const int inf = p->sps[p->active_sps].direct_8x8_inference;
    for (int blk = 0; blk < 16; blk++) {
        const int bx = blk & 3, by = blk >> 2, q = (by >> 1) * 2 + (bx >> 1);
        const int cb = inf ? ((by >> 1) * 3) * 4 + (bx >> 1) * 3 : blk;                          /* corner 4x4 of the quadrant: (0|3, 0|3) */
        const int col_zero = col_short && cr[q] == 0 && cm[cb * 2] >= -1 && cm[cb * 2] <= 1 && cm[cb * 2 + 1] >= -1 && cm[cb * 2 + 1] <= 1;
        for (int l = 0; l < 2; l++) {
            d->ref[l][q] = (int8_t)ref[l];
            const int z = zero_pred || ref[l] < 0 || (ref[l] == 0 && col_zero);
            d->mv[l][blk][0] = (int16_t)(z ? 0 : mv[l][0]); d->mv[l][blk][1] = (int16_t)(z ? 0 : mv[l][1]);
        }
    }
}

static void direct_temporal(const p264parse *p, direct_t *d)
{   /* 8.4.1.2.3: list 0 points at the picture the co-located block referred to, list 1 at RefPicList1[0]; the co-located
     * vector split in proportion to the picture distances */
    const int col_slot = p->list1[0];
    const int8_t *cr = p->col_ref[col_slot] + p->mbi * 4; const int32_t *cu = p->col_uid[col_slot] + p->mbi * 4;
    const int16_t *cm = p->col_mv[col_slot] + p->mbi * 32;
    const int inf = p->sps[p->active_sps].direct_8x8_inference;
    for (int q = 0; q < 4; q++) {
        int r0 = 0, scale = 0, use_col = 0;                       /* intra co-located block: both indices 0, zero vectors */
        if (cr[q] >= 0) {
            r0 = -1;
            for (int i = 0; i < p->n_list0 && r0 < 0; i++) if ((int32_t)p->dpb[p->list0[i]].uid == cu[q]) r0 = i;   /* lowest index that names that picture */
            if (r0 < 0) r0 = 0;                                   /* (a stream that dropped it from list 0: not conformant; stay defined) */
            const dpb_frame_t *f0 = &p->dpb[p->list0[r0]], *f1 = &p->dpb[col_slot];
            const int tb = clip3i(p->cur_poc - f0->poc, -128, 127), td = clip3i(f1->poc - f0->poc, -128, 127);
            use_col = 1;
            if (f0->is_long || td == 0) scale = -1;               /* mvL0 = mvCol, mvL1 = 0 */
            else { const int tx = (16384 + (td < 0 ? -td : td) / 2) / td; scale = clip3i((tb * tx + 32) >> 6, -1024, 1023); }
        }
        d->ref[0][q] = (int8_t)r0; d->ref[1][q] = 0;
        for (int k = 0; k < 4; k++) {
            const int bx = (q & 1) * 2 + (k & 1), by = (q >> 1) * 2 + (k >> 1), blk = by * 4 + bx;
            const int cb = inf ? ((q >> 1) * 3) * 4 + (q & 1) * 3 : blk;
            const int cx = use_col ? cm[cb * 2] : 0, cy = use_col ? cm[cb * 2 + 1] : 0;
            int m0x = cx, m0y = cy, m1x = 0, m1y = 0;
            if (use_col && scale != -1) { m0x = (scale * cx + 128) >> 8; m0y = (scale * cy + 128) >> 8; m1x = m0x - cx; m1y = m0y - cy; }
            d->mv[0][blk][0] = (int16_t)m0x; d->mv[0][blk][1] = (int16_t)m0y;
            d->mv[1][blk][0] = (int16_t)m1x; d->mv[1][blk][1] = (int16_t)m1y;
        }
    }
}

static void direct_predict(const p264parse *p, direct_t *d)
{
    if (p->sh.direct_spatial) direct_spatial(p, d); else direct_temporal(p, d);
}
/* copy quadrant q of a direct prediction into the picture arrays (vectors of an unused list are zero, its index -1) */
static void store_direct_quadrant(p264parse *p, const direct_t *d, int q, int list)
{
    picbuf_t *b = &p->buf[p->cur];
    (list ? b->ref1 : b->ref)[p->mbi * 4 + q] = d->ref[list][q];
    for (int k = 0; k < 4; k++) {
        const int bx = (q & 1) * 2 + (k & 1), by = (q >> 1) * 2 + (k >> 1), blk = by * 4 + bx;
        const int used = d->ref[list][q] >= 0;
        set_motion_l(p, bx, by, 1, 1, used ? d->mv[list][blk][0] : 0, used ? d->mv[list][blk][1] : 0, list);
    }
}

static void decode_bskip(p264parse *p)
{
    picbuf_t *q = &p->buf[p->cur];
    p264hip_mb_t *m = &q->mb[p->mbi];
    begin_mb(p, m);
    m->mb_type = P264_MB_B;
    memset(p->nnz + (size_t)p->mbi * 24, 0, 24);
    memset(q->i4 + p->mbi * 16, 2, 16);
    direct_t d;
    direct_predict(p, &d);
    for (int l = 0; l < 2; l++) for (int k = 0; k < 4; k++) store_direct_quadrant(p, &d, k, l);
    m->coef_index = (uint32_t)q->coef_n;
    finish_mb_qp(p, m, 0, p->sh.qp);
    p->last_dqp = 0;
    if (p->cabac_on) p->cinfo[p->mbi] |= CI_SKIP | CI_DIRECT16 | CI_D8(0) | CI_D8(1) | CI_D8(2) | CI_D8(3);
}

/* which lists a partition predicts from: bit 0 list 0, bit 1 list 1 */
enum { PRED_L0 = 1, PRED_L1 = 2, PRED_BI = 3 };
static const uint8_t b_pair[9][2] = {           /* mb_type 4..21, table 7-14: (type - 4) >> 1 -> prediction of the two partitions */
    { PRED_L0, PRED_L0 }, { PRED_L1, PRED_L1 }, { PRED_L0, PRED_L1 }, { PRED_L1, PRED_L0 }, { PRED_L0, PRED_BI },
    { PRED_L1, PRED_BI }, { PRED_BI, PRED_L0 }, { PRED_BI, PRED_L1 }, { PRED_BI, PRED_BI } };
static const uint8_t b_sub_pred[13] = { 0, PRED_L0, PRED_L1, PRED_BI, PRED_L0, PRED_L0, PRED_L1, PRED_L1, PRED_BI, PRED_BI, PRED_L0, PRED_L1, PRED_BI };   /* table 7-18; 0 = direct */
static const uint8_t b_sub_w[13] = { 2, 2, 2, 2, 2, 1, 2, 1, 2, 1, 1, 1, 1 }, b_sub_h[13] = { 2, 2, 2, 2, 1, 2, 1, 2, 1, 2, 1, 1, 1 };    /* sub-partition size in 4x4 units */

static int parse_mb_b_t(p264parse *p, bitrd_t *b, unsigned t)
{
    picbuf_t *q = &p->buf[p->cur];
    p264hip_mb_t *m = &q->mb[p->mbi];
    if (t >= 23) {                                            /* intra macroblock in a B slice: the I-slice syntax with the type offset */
        return parse_mb_t(p, b, t, (int)t - 23);
    }
    mbcoef_t cf; cf.mask = 0;
    begin_mb(p, m);
    m->mb_type = P264_MB_B;
    int8_t *ref[2] = { q->ref + p->mbi * 4, q->ref1 + p->mbi * 4 };
    memset(q->i4 + p->mbi * 16, 2, 16);
    memset(ref[0], -1, 4); memset(ref[1], -1, 4);
    memset(q->mv + p->mbi * 32, 0, 64); memset(q->mv1 + p->mbi * 32, 0, 64);
    const int nref[2] = { p->sh.num_ref_idx, p->sh.num_ref_idx_l1 };
    int direct_all = 0;
    if (t == 0) {                                             /* B_Direct_16x16: like B_Skip, with a residual */
        direct_t d;
        direct_predict(p, &d);
        for (int l = 0; l < 2; l++) for (int k = 0; k < 4; k++) store_direct_quadrant(p, &d, k, l);
        direct_all = 1;
        if (p->cabac_on) p->cinfo[p->mbi] |= CI_DIRECT16 | CI_D8(0) | CI_D8(1) | CI_D8(2) | CI_D8(3);
    } else if (t <= 21) {
        /* one or two partitions: geometry and which lists each uses */
        int nparts, geo[2][4], pred[2];
        if (t <= 3) { nparts = 1; geo[0][0] = 0; geo[0][1] = 0; geo[0][2] = 4; geo[0][3] = 4; pred[0] = t == 1 ? PRED_L0 : t == 2 ? PRED_L1 : PRED_BI; pred[1] = 0; }
        else {
            nparts = 2;
            const int tall = (int)t & 1;                      /* odd types: 8x16 */
            for (int k = 0; k < 2; k++) { geo[k][0] = tall ? 2 * k : 0; geo[k][1] = tall ? 0 : 2 * k; geo[k][2] = tall ? 2 : 4; geo[k][3] = tall ? 4 : 2; pred[k] = b_pair[(t - 4) >> 1][k]; }
        }
        int r[2][2] = { { -1, -1 }, { -1, -1 } };
        for (int l = 0; l < 2; l++)                           /* all ref_idx_l0, then all ref_idx_l1 */
            for (int k = 0; k < nparts; k++) {
                if (!(pred[k] & (1 << l))) continue;
                r[l][k] = rd_ref_idx(p, b, l, geo[k][0], geo[k][1], nref[l]);
                if (r[l][k] < 0 || r[l][k] >= nref[l]) { ERR(p, "ref_idx out of range"); return -1; }
                for (int y = geo[k][1] >> 1; y < (geo[k][1] + geo[k][3]) >> 1; y++)          /* (at once: the next partition's context looks at it) */
                    for (int x = geo[k][0] >> 1; x < (geo[k][0] + geo[k][2]) >> 1; x++) ref[l][y * 2 + x] = (int8_t)r[l][k];
            }
        for (int l = 0; l < 2; l++)                           /* all mvd_l0, then all mvd_l1; a partition that does not use the list still
                                                               * becomes "decoded" for it (index -1, zero vector) when its turn comes */
            for (int k = 0; k < nparts; k++) {
                if (!(pred[k] & (1 << l))) { set_motion_l(p, geo[k][0], geo[k][1], geo[k][2], geo[k][3], 0, 0, l); continue; }
                int dx, dy, px, py;
                if (rd_mvd(p, b, l, geo[k][0], geo[k][1], geo[k][2], geo[k][3], &dx, &dy) < 0) { ERR(p, "mvd out of range"); return -1; }
                const int dir = nparts == 1 ? 0 : geo[0][2] == 4 ? 1 + k : 3 + k;
                predict_mv_l(p, geo[k][0], geo[k][1], geo[k][2], r[l][k], dir, &px, &py, l);
                set_motion_l(p, geo[k][0], geo[k][1], geo[k][2], geo[k][3], px + dx, py + dy, l);
            }
    } else {                                                  /* 22: B_8x8 */
        int sub[4], any_direct = 0;
        for (int k = 0; k < 4; k++) {
            sub[k] = rd_sub_mb_type(p, b);
            if (sub[k] > 12) { ERR(p, "invalid B sub_mb_type %d", sub[k]); return -1; }
            any_direct |= sub[k] == 0;
            if (p->cabac_on && sub[k] == 0) p->cinfo[p->mbi] |= CI_D8(k);
        }
        direct_t d;
        if (any_direct) direct_predict(p, &d);                /* from the macroblock's neighbours, before any of its own motion exists */
        int r[2][4];
        for (int l = 0; l < 2; l++)
            for (int k = 0; k < 4; k++) {
                r[l][k] = -1;
                if (!(b_sub_pred[sub[k]] & (1 << l))) continue;
                r[l][k] = rd_ref_idx(p, b, l, (k & 1) * 2, (k >> 1) * 2, nref[l]);
                if (r[l][k] < 0 || r[l][k] >= nref[l]) { ERR(p, "ref_idx out of range"); return -1; }
                ref[l][k] = (int8_t)r[l][k];
            }
        for (int l = 0; l < 2; l++)
            for (int k = 0; k < 4; k++) {
                const int ox = (k & 1) * 2, oy = (k >> 1) * 2;
                if (sub[k] == 0) { store_direct_quadrant(p, &d, k, l); continue; }             /* its turn: the derived motion becomes visible */
                if (!(b_sub_pred[sub[k]] & (1 << l))) { set_motion_l(p, ox, oy, 2, 2, 0, 0, l); continue; }
                const int sw = b_sub_w[sub[k]], shh = b_sub_h[sub[k]];
                for (int sy = 0; sy < 2; sy += shh)
                    for (int sx = 0; sx < 2; sx += sw) {
                        int dx, dy, px, py;
                        if (rd_mvd(p, b, l, ox + sx, oy + sy, sw, shh, &dx, &dy) < 0) { ERR(p, "mvd out of range"); return -1; }
                        predict_mv_l(p, ox + sx, oy + sy, sw, r[l][k], 0, &px, &py, l);
                        set_motion_l(p, ox + sx, oy + sy, sw, shh, px + dx, py + dy, l);
                    }
            }
    }
    (void)direct_all;
    /* ---- coded_block_pattern, mb_qp_delta, residual: as in P macroblocks ---- */
    const int c = rd_cbp(p, b, 0);
    if (c < 0) { ERR(p, "invalid cbp"); return -1; }
    m->cbp = (uint8_t)c;
    int qp = p->sh.qp, has_res = m->cbp != 0;
    if (has_res) {
        int dqp = rd_mb_qp_delta(p, b);
        if (dqp < -52 || dqp > 52) { ERR(p, "mb_qp_delta out of range"); return -1; }
        if (p->strict_qp) qp = (p->qp_pred + dqp + 52) % 52;
        else qp = p->sh.qp + dqp;
        if (rd_residual(p, b, m, &cf) < 0) { ERR(p, "read residual data failed"); return -1; }
    } else { memset(p->nnz + (size_t)p->mbi * 24, 0, 24); p->last_dqp = 0; }
    if (store_coefs(p, m, &cf) < 0) return -1;
    finish_mb_qp(p, m, has_res, qp);
    if (br_overrun(b)) { ERR(p, "macroblock overruns the slice data"); return -1; }
    return 0;
}

/* one non-skipped macroblock of the current slice, whatever its slice type and entropy coder */
static int parse_mb(p264parse *p, bitrd_t *b)
{
    const int ti = p->cabac_on ? cb_mb_type(p) : (int)br_ue(b);
    if (ti < 0) return -1;
    const unsigned t = (unsigned)ti;
    if (p->sh.type == P264_SLICE_B) return parse_mb_b_t(p, b, t);
    int intra_t = -1;
    if (p->sh.type == P264_SLICE_I) intra_t = (int)t;
    else if (t >= 5) intra_t = (int)t - 5;
    return parse_mb_t(p, b, t, intra_t);
}

/* position of the rbsp_stop_one_bit, in bits from the start of the payload */
static long rbsp_stop_bit(const uint8_t *buf, int size)
{
    int n = size;
    while (n > 0 && buf[n-1] == 0) n--;
    if (n == 0) return 0;
    int tz = 0; while (!((buf[n-1] >> tz) & 1)) tz++;
    return (long)n * 8 - 1 - tz;
}

/* ---------------------------------------------------------------- slice ------------------ */
static void publish_picture(p264parse *p)
{
    picbuf_t *q = &p->buf[p->cur];
    p264hip_picture_t *d = &p->desc[p->cur];
    const pps_t *pps = &p->pps[p->active_pps];
    memset(d, 0, sizeof *d);
    d->mb_w = p->mb_w; d->mb_h = p->mb_h;
    d->slice_type = p->sh0.type;
    d->chroma_qp_offset = pps->chroma_qp_offset;
    d->deblock = (!pps->deblock_ctrl || p->pic_deblock) ? 1 : 0;     /* per-macroblock `edges` gate the slices that switch it off */
    d->alpha_c0_offset = p->pic_alpha; d->beta_offset = p->pic_beta;
    d->dst_slot = p->cur_slot;
    d->n_ref = p->n_list0;
    for (int i = 0; i < p->n_list0; i++) d->ref_slot[i] = p->list0[i];
    d->n_coef_blocks = (uint32_t)q->coef_n;
    d->frame_num = (uint32_t)p->sh0.frame_num;
    d->mb = q->mb; d->mv = q->mv; d->ref_idx = q->ref; d->i4modes = q->i4; d->coefs = q->coef;
    if (p->sh0.type == P264_SLICE_B) {
        d->mv_l1 = q->mv1; d->ref_idx_l1 = q->ref1;
        d->n_ref_l1 = p->n_list1;
        for (int i = 0; i < p->n_list1; i++) d->ref_slot_l1[i] = p->list1[i];
        d->weighted_bipred = p->weighted_bipred;
        memcpy(d->bipred_weight, p->bipred_weight, sizeof d->bipred_weight);
    }
}

/* decoder/decoder.c:502-593,598-664 */
static int decode_slice(p264parse *p, int nal_type, int nal_ref_idc, const uint8_t *payload, int size,
                        const p264hip_picture_t **pic)
{
    bitrd_t b; br_init(&b, payload, (size_t)size);
    slice_t sh;
    int rc = parse_slice_header(p, &b, nal_type, nal_ref_idc, &sh);
    if (rc < 0) { ERR(p, "slice header decode failed"); return -1; }
    if (rc > 0) return 0;
    const pps_t *pps = &p->pps[sh.pps_id];
    /* A new context (buffers, frame store) only when the picture geometry or the frame-store size changes; switching
     * between parameter sets of the same geometry just activates them (H.264 7.4.1.2.1: the frame store lives on.  The
     * reference loops forever in its context switch here, decoder/decoder.c:380-396, so there is nothing to match). */
    {
        const sps_t *sps = &p->sps[pps->sps_id];
        int slots = sps->num_ref_frames + 1;
        if (slots < 2) slots = 2;
        if (p->active_sps < 0 || !p->buf[0].mb || sps->mb_w != p->mb_w || sps->mb_h != p->mb_h || slots != p->slots) {
            if (init_context(p, pps->sps_id, sh.pps_id) < 0) { ERR(p, "out of memory"); return -1; }
        } else { p->active_sps = pps->sps_id; p->active_pps = sh.pps_id; }
    }

    if (sh.first_mb == 0 || !p->pic_open) {
        /* first slice of a new picture */
        if (sh.first_mb != 0) { ERR(p, "slice starts at MB %d but no picture is open", sh.first_mb); return -1; }
        p->pic_open = 1; p->next_mb = 0; p->slice_no = 0;
        p->pic_is_idr = nal_type == NAL_SLICE_IDR; p->pic_ref_idc = nal_ref_idc;
        if (p->pic_is_idr) {                      /* p264_slice_idr, decoder/decoder.c:43-64 */
            for (int i = 0; i < p->slots; i++) p->dpb[i].used = 0;
            p->cur_slot = 0;
        }
        p->sh0 = sh;
        p->pic_deblock = 0; p->pic_alpha = p->pic_beta = 0;
        p->buf[p->cur].coef_n = 0;
        memset(p->slice_of, 0xff, (size_t)p->n_mb * sizeof(uint16_t));
        p->n_list0 = 0; p->n_list1 = 0;
        p->cur_poc = picture_order_count(p, &sh, p->pic_is_idr, nal_ref_idc);
        p->cur_uid = ++p->next_uid;
        if (p->buf[p->cur].ref1) memset(p->buf[p->cur].ref1, -1, (size_t)p->n_mb * 4);   /* nothing predicts from list 1 until a B macroblock says so */
    } else {
        if (sh.first_mb != p->next_mb) { ERR(p, "slice starts at MB %d, expected %d", sh.first_mb, p->next_mb); return -1; }
        p->slice_no++;
    }
    if (sh.disable_deblock != 1) {                /* the filter parameters are per picture on the device */
        if (!p->pic_deblock) { p->pic_deblock = 1; p->pic_alpha = sh.alpha_off; p->pic_beta = sh.beta_off; }
        else if (sh.alpha_off != p->pic_alpha || sh.beta_off != p->pic_beta) { ERR(p, "per-slice deblocking offsets unsupported"); return -1; }
    }
    p->sh = sh;
    if (sh.type == P264_SLICE_P || sh.type == P264_SLICE_B) {
        int prev[P264HIP_MAX_REFS], n_prev = p->n_list0, prev1[P264HIP_MAX_REFS], n_prev1 = p->n_list1;
        memcpy(prev, p->list0, sizeof prev); memcpy(prev1, p->list1, sizeof prev1);
        if ((p->n_list0 = build_list(p, &sh, 0, p->list0)) < 0) { p->n_list0 = 0; return -1; }
        /* reference indices are resolved through ONE list 0 (and one list 1) per picture on the device */
        if (n_prev && (n_prev != p->n_list0 || memcmp(prev, p->list0, sizeof(int) * (size_t)n_prev))) { ERR(p, "slices of one picture with different reference lists unsupported"); return -1; }
        if (sh.type == P264_SLICE_B) {
            if (!p->has_col) { ERR(p, "B slice without list-1 storage (Baseline parameter set)"); return -1; }
            if ((p->n_list1 = build_list(p, &sh, 1, p->list1)) < 0) { p->n_list1 = 0; return -1; }
            if (n_prev1 && (n_prev1 != p->n_list1 || memcmp(prev1, p->list1, sizeof(int) * (size_t)n_prev1))) { ERR(p, "slices of one picture with different reference lists unsupported"); return -1; }
            if (p->sh0.type != P264_SLICE_B && p->slice_no > 0 && p->sh0.type == P264_SLICE_P) { ERR(p, "P and B slices in one picture unsupported"); return -1; }
            p->sh0.type = P264_SLICE_B;           /* a picture with any B slice is reconstructed as B */
            p->weighted_bipred = pps->weighted_bipred == 2;
            implicit_weights(p);
        } else {
            if (p->sh0.type == P264_SLICE_B) { ERR(p, "P and B slices in one picture unsupported"); return -1; }
            p->sh0.type = P264_SLICE_P;           /* a picture with any P slice is reconstructed as P */
        }
    }
    p->qp_pred = sh.qp;
    /* The reference's QP bookkeeping (delta added to the slice QP, last QP carried over residual-free macroblocks and across
     * pictures: SURVEY A-Q2) is kept for what the reference decodes - Baseline CAVLC I / P slices.  It decodes neither CABAC nor
     * B slices nor any other profile: there is no behaviour to match there, those streams get the standard's chain. */
    p->strict_qp = (p->opts & P264PARSE_OPT_STRICT) || pps->cabac || p->sps[pps->sps_id].profile_idc != 66 || sh.type == P264_SLICE_B;

    p->cabac_on = pps->cabac;
    p->last_dqp = 0;
    if (p->cabac_on) {
        /* slice_data( ) with CABAC (7.3.4): alignment bits, the engine started on the next byte, contexts from the slice QP;
         * per macroblock mb_skip_flag (P / B), the macroblock, end_of_slice_flag */
        if (!p->cinfo) { ERR(p, "CABAC slice without its context storage (Baseline parameter set)"); return -1; }
        const size_t at = (size_t)((br_consumed(&b) + 7) >> 3);
        if (at >= (size_t)size) { ERR(p, "CABAC slice without data"); return -1; }
        p264cabac_init_contexts(&p->cb, sh.type == P264_SLICE_I, sh.cabac_init_idc, sh.qp);
        p264cabac_start(&p->cb, payload + at, (size_t)size - at);
        for (;;) {
            if (p->next_mb >= p->n_mb) { ERR(p, "slice data runs past the picture"); p->pic_open = 0; return -1; }
            p->mbi = p->next_mb; p->mbx = p->mbi % p->mb_w; p->mby = p->mbi / p->mb_w;
            /* (the neighbour flags of the macroblock are needed before its first bin: begin_mb computes them again, identically) */
            mb_neighbours(p);
            if (sh.type != P264_SLICE_I && cb_mb_skip_flag(p)) { if (sh.type == P264_SLICE_B) decode_bskip(p); else decode_pskip(p); }
            else if (parse_mb(p, &b) < 0) { ERR(p, "macroblock read failed [%d,%d]", p->mbx, p->mby); p->pic_open = 0; return -1; }
            if (p264cabac_bits_left(&p->cb) < -64) { ERR(p, "CABAC data overrun"); p->pic_open = 0; return -1; }
            p->slice_of[p->mbi] = (uint16_t)p->slice_no;
            p->next_mb++;
            if (p264cabac_terminate(&p->cb)) break;              /* end_of_slice_flag */
        }
    } else {
    long stop = rbsp_stop_bit(payload, size);
    p->skip_run = -1;
    while (p->next_mb < p->n_mb) {
        p->mbi = p->next_mb; p->mbx = p->mbi % p->mb_w; p->mby = p->mbi / p->mb_w;
        if (p->skip_run <= 0 && (long)br_consumed(&b) >= stop) break;   /* !more_rbsp_data(): the slice ends here */
        if (sh.type != P264_SLICE_I && p->skip_run < 0) {
            p->skip_run = (int)br_ue(&b);
            if (p->skip_run > p->n_mb - p->next_mb) { ERR(p, "mb_skip_run %d runs past the picture", p->skip_run); p->pic_open = 0; return -1; }
        }
        if (p->skip_run > 0) {
            if (sh.type == P264_SLICE_B) decode_bskip(p); else decode_pskip(p);
            p->skip_run--;
        } else {
            if ((long)br_consumed(&b) >= stop) break;
            if (parse_mb(p, &b) < 0) { ERR(p, "macroblock read failed [%d,%d]", p->mbx, p->mby); p->pic_open = 0; return -1; }
            p->skip_run = -1;
        }
        p->slice_of[p->mbi] = (uint16_t)p->slice_no;
        p->next_mb++;
    }
    }
    if (p->next_mb < p->n_mb) return 0;                       /* wait for the next slice of this picture */

    publish_picture(p);
    *pic = &p->desc[p->cur];
    finish_picture_marking(p);
    p->cur ^= 1;
    p->pic_open = 0;
    return 1;
}

/* ---------------------------------------------------------------- public ---------------- */
p264parse *p264parse_open(int options)
{
    if (p264amd_cpu_refuse("p264parse_open")) return NULL;     /* (cpu_check.c: the host objects are built for x86-64-v3) */
    if (cavlc_global_init() != 0) { fprintf(stderr, "p264amd: CAVLC tables are not prefix-free\n"); return NULL; }
    p264parse *p = (p264parse *)calloc(1, sizeof *p);
    if (!p) return NULL;
    p->opts = options;
    p->active_sps = p->active_pps = -1;
    return p;
}

void p264parse_set_allocator(p264parse *p, void *(*alloc)(size_t bytes), void (*release)(void *ptr))
{
    if (!p || p->n_mb) return;                              /* only before the first context is built */
    p->alloc = alloc; p->release = release;
}

void p264parse_close(p264parse *p)
{
    if (!p) return;
    free_context(p, 1);
    free(p);
}

int p264parse_nal(p264parse *p, int nal_type, int nal_ref_idc, const uint8_t *payload, int size,
                  const p264hip_picture_t **pic)
{
    if (pic) *pic = NULL;
    if (!p || !payload || size < 0 || !pic) return -1;
    bitrd_t b;
    switch (nal_type) {
    case NAL_SPS: br_init(&b, payload, (size_t)size);
        if (parse_sps(p, &b) < 0) { ERR(p, "sps read failed"); return -1; }
        return 0;
    case NAL_PPS: br_init(&b, payload, (size_t)size);
        if (parse_pps(p, &b) < 0) { ERR(p, "pps read failed"); return -1; }
        return 0;
    case NAL_SLICE_IDR:
    case NAL_SLICE:
        return decode_slice(p, nal_type, nal_ref_idc, payload, size, pic);
    case NAL_SLICE_DPA: case NAL_SLICE_DPB: case NAL_SLICE_DPC:
        ERR(p, "partitioned stream unsupported"); return -1;
    default:
        return 0;                                            /* SEI, AUD, ...: ignored (decoder.c:797-799) */
    }
}

int p264parse_mb_width(const p264parse *p)   { return p ? p->mb_w : 0; }
int p264parse_mb_height(const p264parse *p)  { return p ? p->mb_h : 0; }
int p264parse_slots(const p264parse *p)      { return p ? p->slots : 0; }
int p264parse_generation(const p264parse *p) { return p ? p->generation : 0; }

/* first index >= from with buf[i..i+2] == 00 00 01, or an index with i + 3 > size; the scan hops between zero bytes */
static int64_t annexb_find(const uint8_t *buf, int64_t size, int64_t from)
{
    int64_t i = from;
    while (i + 3 <= size) {
        const uint8_t *z = (const uint8_t *)memchr(buf + i, 0, (size_t)(size - 2 - i));
        if (!z) return size;
        i = z - buf;
        if (buf[i + 1] == 0 && buf[i + 2] == 1) return i;
        i++;
    }
    return size;
}

int p264_annexb_next(const uint8_t *buf, int64_t size, int64_t *pos, int64_t *nal_off, int64_t *nal_len)
{
    int64_t i = annexb_find(buf, size, *pos);
    if (i + 3 > size) { *pos = size; return 0; }
    int64_t start = i + 3, j = annexb_find(buf, size, start);
    int64_t end = j + 3 <= size ? j : size;
    *pos = end;
    while (end > start && buf[end-1] == 0) end--;           /* zeros in front of a start code belong to it */
    *nal_off = start; *nal_len = end - start;
    return 1;
}

/* ---------------------------------------------------------------- known-answer surface ---- */
/* B-picture derivations on hand-made state (tests/test_direct_kat.py drives them with the vectors recorded from the reference's
 * encoder-side functions, tests/golden/kat_direct.npz): the same static functions the macroblock layer calls, on a parser whose
 * picture is 3 x 2 macroblocks with the current one at (1, 1). */
static p264parse *kat_state(void)
{
    p264parse *p = (p264parse *)calloc(1, sizeof *p);
    if (!p) return NULL;
    p->mb_w = 3; p->mb_h = 2; p->n_mb = 6; p->slots = P264HIP_MAX_REFS + 1;
    picbuf_t *q = &p->buf[0];
    q->mv = (int16_t *)calloc(6 * 32, sizeof(int16_t)); q->mv1 = (int16_t *)calloc(6 * 32, sizeof(int16_t));
    q->ref = (int8_t *)malloc(6 * 4); q->ref1 = (int8_t *)malloc(6 * 4);
    for (int i = 0; i <= P264HIP_MAX_REFS; i++) {
        p->col_mv[i] = (int16_t *)calloc(6 * 32, sizeof(int16_t)); p->col_ref[i] = (int8_t *)malloc(6 * 4); p->col_uid[i] = (int32_t *)malloc(6 * 4 * sizeof(int32_t));
    }
    p->has_col = 1; p->active_sps = 0;
    p->mbx = 1; p->mby = 1; p->mbi = 4;
    return p;
}
static void kat_free(p264parse *p)
{
    free(p->buf[0].mv); free(p->buf[0].mv1); free(p->buf[0].ref); free(p->buf[0].ref1);
    for (int i = 0; i <= P264HIP_MAX_REFS; i++) { free(p->col_mv[i]); free(p->col_ref[i]); free(p->col_uid[i]); }
    free(p);
}
/* lists of a B picture from picture order counts: slot i holds list-0 entry i, slot n0 + k list-1 entry k unless that picture
 * (same order count) already sits in list 0 */
static void kat_lists(p264parse *p, int n0, const int *poc0, int n1, const int *poc1, int cur_poc)
{
    int n = 0;
    for (int i = 0; i < n0; i++) { p->dpb[n].used = 1; p->dpb[n].poc = poc0[i]; p->dpb[n].uid = (uint32_t)(poc0[i] + 4096); p->list0[i] = n++; }
    for (int k = 0; k < n1; k++) {
        int s = -1;
        for (int i = 0; i < n0; i++) if (poc0[i] == poc1[k]) { s = p->list0[i]; break; }
        if (s < 0) { p->dpb[n].used = 1; p->dpb[n].poc = poc1[k]; p->dpb[n].uid = (uint32_t)(poc1[k] + 4096); s = n++; }
        p->list1[k] = s;
    }
    p->n_list0 = n0; p->n_list1 = n1; p->cur_poc = cur_poc;
}

int p264parse_kat_bipred(int n0, const int *poc0, int n1, const int *poc1, int cur_poc, int16_t *weights)
{
    if (n0 < 0 || n1 < 0 || n0 > 8 || n1 > 8) return -1;
    p264parse *p = kat_state();
    if (!p) return -1;
    kat_lists(p, n0, poc0, n1, poc1, cur_poc);
    p->weighted_bipred = 1;
    implicit_weights(p);
    memcpy(weights, p->bipred_weight, sizeof p->bipred_weight);
    kat_free(p);
    return 0;
}

int p264parse_kat_direct(int spatial, const int8_t *nb_ref, const int16_t *nb_mv, int col_intra, const int8_t *col_ref, const int16_t *col_mv,
                         int n0, const int *poc0, int poc1_0, int cur_poc, int n_col_list, const int *col_list_poc,
                         int8_t *out_ref, int16_t *out_mv)
{
    if (n0 < 1 || n0 > 8 || n_col_list < 0 || n_col_list > 8) return -1;
    p264parse *p = kat_state();
    if (!p) return -1;
    kat_lists(p, n0, poc0, 1, &poc1_0, cur_poc);
    p->sh.direct_spatial = spatial;
    picbuf_t *q = &p->buf[0];
    memset(q->ref, -1, 24); memset(q->ref1, -1, 24);
    /* neighbours A, B, C, D: the 4x4 block left of / above / above right of / above left of the macroblock's first block */
    static const int nb_mb[4] = { 3, 1, 2, 0 }, nb_quad[4] = { 1, 2, 2, 3 }, nb_sub[4] = { 3, 12, 12, 15 };
    static const int nb_flag[4] = { P264_AVAIL_LEFT, P264_AVAIL_TOP, P264_AVAIL_TOPRIGHT, P264_AVAIL_TOPLEFT };
    p->cur_avail = 0;
    for (int n = 0; n < 4; n++) {
        if (nb_ref[n] != -2) p->cur_avail |= nb_flag[n];        /* (availability is a property of the macroblock: both lists agree) */
        for (int l = 0; l < 2; l++) {
            (l ? q->ref1 : q->ref)[nb_mb[n] * 4 + nb_quad[n]] = nb_ref[l * 4 + n] < 0 ? -1 : nb_ref[l * 4 + n];
            int16_t *mv = (l ? q->mv1 : q->mv) + (nb_mb[n] * 16 + nb_sub[n]) * 2;
            mv[0] = nb_mv[(l * 4 + n) * 2]; mv[1] = nb_mv[(l * 4 + n) * 2 + 1];
        }
    }
    /* the co-located macroblock as finish_picture_marking would have left it: the list-0 motion where there is one, else list 1 */
    const int cs = p->list1[0];
    for (int q8 = 0; q8 < 4; q8++) {
        const int r0 = col_intra ? -1 : col_ref[q8], r1 = col_intra ? -1 : col_ref[4 + q8], r = r0 >= 0 ? r0 : r1;
        p->col_ref[cs][p->mbi * 4 + q8] = (int8_t)r;
        p->col_uid[cs][p->mbi * 4 + q8] = r < 0 || r >= n_col_list ? -1 : col_list_poc[r] + 4096;   /* (list 1 of the co-located picture: not modelled, see the test) */
        for (int k = 0; k < 4; k++) {
            const int blk = (q8 >> 1) * 8 + (q8 & 1) * 2 + (k >> 1) * 4 + (k & 1);
            const int16_t *src = col_mv + ((r0 >= 0 ? 0 : 16) + blk) * 2;
            p->col_mv[cs][(p->mbi * 16 + blk) * 2] = r < 0 ? 0 : src[0]; p->col_mv[cs][(p->mbi * 16 + blk) * 2 + 1] = r < 0 ? 0 : src[1];
        }
    }
    direct_t d;
    direct_predict(p, &d);
    memcpy(out_ref, d.ref, sizeof d.ref); memcpy(out_mv, d.mv, sizeof d.mv);
    kat_free(p);
    return 0;
}
