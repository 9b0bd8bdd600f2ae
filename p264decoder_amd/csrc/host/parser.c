/* parser.c - host-side H.264 bitstream layer (see include/p264parse.h).
 *
 * CPU work by design: entropy decoding is bit-serial.  Everything it learns about a picture
 * is written straight into the structure-of-arrays buffers of p264hip_picture_t, which the
 * HIP layer uploads as they are.
 *
 * Supported subset = what the reference decodes: CAVLC, I and P slices,
 * frame MBs, one reference list.  Beyond it we follow ITU-T H.264 where that is cheap
 * (several slices per picture, multiple reference frames, list-0 reordering of short-term
 * pictures, sub-8x8 partitions); everything else is rejected with -1 and a line on stderr,
 * the reference's error convention (decoder/decoder.c:558-577,780-795).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "p264parse.h"
#include "bits.h"
#include "vlc.h"
#include "cavlc_tables.h"

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
    int num_ref_frames, gaps_allowed, mb_w, mb_h, frame_mbs_only;
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
    int no_output_of_prior, long_term_flag, adaptive_marking;
    int n_mmco; struct { int op, a, b; } mmco[34];   /* memory_management_control_operation 1..6 and its operands */
} slice_t;

typedef struct { int used, frame_num, pic_num, is_long, long_idx; } dpb_frame_t;   /* long_idx = LongTermFrameIdx (= LongTermPicNum for frames) */

typedef struct {
    p264hip_mb_t *mb; int16_t *mv; int8_t *ref; uint8_t *i4; int16_t *coef;
    size_t coef_cap, coef_n;
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

    dpb_frame_t dpb[P264HIP_MAX_REFS + 1];
    int cur_slot;
    int last_qp;                              /* never reset, like h->mb.i_last_qp (core/macroblock.c:1248-1252) */
    int qp_pred;                              /* strict mode only */

    /* current MB */
    int mbx, mby, mbi;
    unsigned mv_done;                         /* bit (y*4+x): that 4x4 of the current MB has its MV */
    int      cur_avail;                       /* P264_AVAIL_* of the current MB (set by begin_mb) */
    int skip_run;
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
    br_u1(b);                                       /* direct_8x8_inference */
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
    if (q->release) { q->release(q->mb); q->release(q->mv); q->release(q->ref); q->release(q->i4); q->release(q->coef); }
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
        q->mb  = (p264hip_mb_t *)q->alloc(n * sizeof(p264hip_mb_t));
        q->mv  = (int16_t *)q->alloc(n * 32 * sizeof(int16_t));
        q->ref = (int8_t *)q->alloc(n * 4);
        q->i4  = (uint8_t *)q->alloc(n * 16);
        q->coef_cap = n * 4 + 64;
        q->coef = (int16_t *)q->alloc(q->coef_cap * 16 * sizeof(int16_t));
        if (!q->mb || !q->mv || !q->ref || !q->i4 || !q->coef) return -1;
        memset(q->mb, 0, n * sizeof(p264hip_mb_t)); memset(q->mv, 0, n * 32 * sizeof(int16_t));
        memset(q->ref, 0, n * 4); memset(q->i4, 0, n * 16);
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
    if (sh->type != P264_SLICE_P && sh->type != P264_SLICE_I) { ERR(p, "only I/P slices supported (type %d)", sh->type); return -1; }
    if (pps->cabac) { ERR(p, "CABAC unsupported (decoder/macroblock.c:594-597)"); return -1; }
    if (!sps->frame_mbs_only) { ERR(p, "field/MBAFF coding unsupported"); return -1; }

    sh->frame_num = (int)br_u(b, sps->log2_max_frame_num);
    if (nal_type == NAL_SLICE_IDR) sh->idr_pic_id = (int)br_ue(b);
    if (sps->poc_type == 0) {
        br_u(b, sps->log2_max_poc_lsb);
        if (pps->pic_order_present) br_se(b);
    } else if (sps->poc_type == 1 && !sps->delta_pic_order_always_zero) {
        br_se(b);
        if (pps->pic_order_present) br_se(b);
    }
    if (pps->redundant_pic_cnt && br_ue(b) != 0) return 1;       /* redundant picture: ignore the slice */
    sh->num_ref_idx = 0;
    if (sh->type == P264_SLICE_P) {
        sh->num_ref_idx = pps->num_ref_idx_l0;
        if (br_u1(b)) sh->num_ref_idx = (int)br_ue(b) + 1;
        if (sh->num_ref_idx < 1 || sh->num_ref_idx > P264HIP_MAX_REFS) { ERR(p, "num_ref_idx_l0_active %d too large", sh->num_ref_idx); return -1; }
        if (br_u1(b)) {                                           /* ref_pic_list_reordering_flag_l0 */
            for (;;) {
                unsigned idc = br_ue(b);
                if (idc == 3) break;
                if (idc > 3 || sh->n_reorder >= 33 || br_overrun(b)) { ERR(p, "wrong reordering of pic nums idc"); return -1; }
                sh->reorder[sh->n_reorder].idc = (int)idc;
                sh->reorder[sh->n_reorder].arg = (int)br_ue(b);
                sh->n_reorder++;
            }
        }
        if (pps->weighted_pred) { ERR(p, "weighted prediction unsupported (decoder/decoder.c:259-262)"); return -1; }
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
    sh->qp = pps->pic_init_qp + br_se(b);
    if (pps->deblock_ctrl) {
        sh->disable_deblock = (int)br_ue(b);
        if (sh->disable_deblock != 1) { sh->alpha_off = br_se(b); sh->beta_off = br_se(b); }
    }
    if (br_overrun(b)) { ERR(p, "slice header overruns the NAL"); return -1; }
    return 0;
}

/* ---------------------------------------------------------------- frame store ----------- */
/* List 0 of a P picture (H.264 8.2.4.2.1, 8.2.4.3): short-term pictures by descending PicNum (decoder/lists.c:72-143),
 * then long-term pictures by ascending LongTermPicNum, then the slice's reordering commands (short-term: idc 0 / 1,
 * long-term: idc 2; the reference ignores them, decoder/lists.c:146-149). */
static int build_list0(p264parse *p, const slice_t *sh)
{
    const sps_t *sps = &p->sps[p->active_sps];
    int max_fn = 1 << sps->log2_max_frame_num;
    int idx[P264HIP_MAX_REFS + 1], n = 0, n_short;
    for (int i = 0; i < p->slots; i++) {
        if (!p->dpb[i].used || p->dpb[i].is_long || i == p->cur_slot) continue;
        p->dpb[i].pic_num = p->dpb[i].frame_num > sh->frame_num ? p->dpb[i].frame_num - max_fn : p->dpb[i].frame_num;
        int j = n++;
        while (j > 0 && p->dpb[idx[j-1]].pic_num < p->dpb[i].pic_num) { idx[j] = idx[j-1]; j--; }
        idx[j] = i;
    }
    n_short = n;
    for (int i = 0; i < p->slots; i++) {
        if (!p->dpb[i].used || !p->dpb[i].is_long || i == p->cur_slot) continue;
        int j = n++;
        while (j > n_short && p->dpb[idx[j-1]].long_idx > p->dpb[i].long_idx) { idx[j] = idx[j-1]; j--; }
        idx[j] = i;
    }
    if (n == 0) { ERR(p, "P slice without a reference picture"); return -1; }
    int len = sh->num_ref_idx;
    int list[P264HIP_MAX_REFS + 1];
    for (int i = 0; i < len; i++) list[i] = idx[i < n ? i : n - 1];
    int pred = sh->frame_num, at = 0;
    for (int k = 0; k < sh->n_reorder && at < len; k++) {
        int idc = sh->reorder[k].idc, slot = -1;
        if (idc == 2) {                                   /* long_term_pic_num */
            for (int i = n_short; i < n; i++) if (p->dpb[idx[i]].long_idx == sh->reorder[k].arg) slot = idx[i];
        } else {
            int d = sh->reorder[k].arg + 1;
            pred = idc == 0 ? pred - d : pred + d;
            if (pred < 0) pred += max_fn;
            if (pred >= max_fn) pred -= max_fn;
            int want = pred > sh->frame_num ? pred - max_fn : pred;
            for (int i = 0; i < n_short; i++) if (p->dpb[idx[i]].pic_num == want) slot = idx[i];
        }
        if (slot < 0) { ERR(p, "reordering names a picture that is not in the frame store"); return -1; }
        for (int i = len; i > at; i--) list[i] = list[i-1];
        list[at++] = slot;
        int w = at;
        for (int r = at; r <= len; r++) if (list[r] != slot) list[w++] = list[r];
    }
    p->n_list0 = len;
    for (int i = 0; i < len; i++) p->list0[i] = list[i];
    return 0;
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
static nbmv_t nb_motion(const p264parse *p, int x4, int y4)
{
    nbmv_t r = { -2, 0, 0 };
    if (x4 < 0 || y4 < 0) return r;
    int mx = x4 >> 2, my = y4 >> 2;
    if (mx >= p->mb_w || my >= p->mb_h) return r;
    int i = my * p->mb_w + mx, sub = (y4 & 3) * 4 + (x4 & 3);
    if (i == p->mbi) { if (!((p->mv_done >> sub) & 1)) return r; }
    else if (!(i < p->mbi && p->slice_of[i] == (uint16_t)p->slice_no)) return r;
    const picbuf_t *q = &p->buf[p->cur];
    r.ref = q->ref[i * 4 + ((y4 & 2) | ((x4 >> 1) & 1))];
    r.mvx = q->mv[(i * 16 + sub) * 2]; r.mvy = q->mv[(i * 16 + sub) * 2 + 1];
    return r;
}

/* H.264 8.4.1.3 (core/macroblock.c:87-175).  (bx,by,bw) in 4x4 units inside the MB;
 * dir: 0 none, 1 = 16x8 upper, 2 = 16x8 lower, 3 = 8x16 left, 4 = 8x16 right. */
static void predict_mv(const p264parse *p, int bx, int by, int bw, int ref, int dir, int *px, int *py)
{
    int x0 = p->mbx * 4 + bx, y0 = p->mby * 4 + by;
    nbmv_t a = nb_motion(p, x0 - 1, y0), b = nb_motion(p, x0, y0 - 1), c = nb_motion(p, x0 + bw, y0 - 1);
    if (c.ref == -2) c = nb_motion(p, x0 - 1, y0 - 1);
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

static void set_motion(p264parse *p, int bx, int by, int bw, int bh, int mvx, int mvy)
{
    picbuf_t *q = &p->buf[p->cur];
    for (int y = by; y < by + bh; y++)
        for (int x = bx; x < bx + bw; x++) {
            q->mv[(p->mbi * 16 + y * 4 + x) * 2] = (int16_t)mvx;
            q->mv[(p->mbi * 16 + y * 4 + x) * 2 + 1] = (int16_t)mvy;
            p->mv_done |= 1u << (y * 4 + x);
        }
}

/* total_coeff predictor nC (H.264 9.2.1; core/macroblock.c:53-65).  blk: 0-15 luma, 16-23 chroma */
static int predict_nc(const p264parse *p, int blk)
{
    const uint8_t *cur = p->nnz + (size_t)p->mbi * 24;
    int na = -1, nb = -1;
    const int left_ok = p->cur_avail & P264_AVAIL_LEFT, top_ok = p->cur_avail & P264_AVAIL_TOP;    /* (begin_mb) */
    if (blk < 16) {
        int x = blk_x[blk], y = blk_y[blk];
        if (x > 0) na = cur[blk_of_xy[y][x-1]]; else if (left_ok) na = (cur - 24)[blk_of_xy[y][3]];
        if (y > 0) nb = cur[blk_of_xy[y-1][x]]; else if (top_ok) nb = (cur - 24 * p->mb_w)[blk_of_xy[3][x]];
    } else {
        int base = blk < 20 ? 16 : 20, c = blk - base, x = c & 1, y = c >> 1;
        if (x > 0) na = cur[blk - 1]; else if (left_ok) na = (cur - 24)[base + y * 2 + 1];
        if (y > 0) nb = cur[blk - 2]; else if (top_ok) nb = (cur - 24 * p->mb_w)[base + 2 + x];
    }
    if (na >= 0 && nb >= 0) return (na + nb + 1) >> 1;
    return na >= 0 ? na : nb >= 0 ? nb : 0;
}

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
    q->release(q->coef);
    q->coef = n; q->coef_cap = cap;
    return 0;
}

/* residual( ) - decoder/macroblock.c:410-486 */
static int parse_residual(p264parse *p, bitrd_t *b, p264hip_mb_t *m, mbcoef_t *cf)
{
    uint8_t *nnz = p->nnz + (size_t)p->mbi * 24;
    int cbp_l = m->cbp & 15, cbp_c = m->cbp >> 4, tc;
    if (m->mb_type == P264_MB_I16x16) {
        memset(cf->dc_luma, 0, sizeof cf->dc_luma);
        if ((tc = cavlc_read_block(b, predict_nc(p, 0), 16, cf->dc_luma)) < 0) return -1;
        if (tc) cf->mask |= P264_COEF_LUMA_DC;
    }
    int maxc = m->mb_type == P264_MB_I16x16 ? 15 : 16;
    for (int i = 0; i < 16; i++) {
        nnz[i] = 0;
        if (!(cbp_l & (1 << (i >> 2)))) continue;
        memset(cf->blk[i], 0, sizeof cf->blk[i]);
        if ((tc = cavlc_read_block(b, predict_nc(p, i), maxc, cf->blk[i])) < 0) return -1;
        nnz[i] = (uint8_t)tc;
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
        nnz[i] = 0;
        if (!(cbp_c & 2)) continue;
        memset(cf->blk[i], 0, sizeof cf->blk[i]);
        if ((tc = cavlc_read_block(b, predict_nc(p, i), 15, cf->blk[i])) < 0) return -1;
        nnz[i] = (uint8_t)tc;
        if (tc) cf->mask |= 1u << i;
    }
    return 0;
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

static void begin_mb(p264parse *p, p264hip_mb_t *m)
{
    memset(m, 0, sizeof *m);
    p->mv_done = 0;
    int a = 0;
    if (mb_avail(p, p->mbx - 1, p->mby))     a |= P264_AVAIL_LEFT;
    if (mb_avail(p, p->mbx, p->mby - 1))     a |= P264_AVAIL_TOP;
    if (mb_avail(p, p->mbx + 1, p->mby - 1)) a |= P264_AVAIL_TOPRIGHT;
    if (mb_avail(p, p->mbx - 1, p->mby - 1)) a |= P264_AVAIL_TOPLEFT;
    m->avail = (uint8_t)a;
    p->cur_avail = a;
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
    if (p->opts & P264PARSE_OPT_STRICT) {
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
    int x0 = p->mbx * 4, y0 = p->mby * 4;
    nbmv_t a = nb_motion(p, x0 - 1, y0), b = nb_motion(p, x0, y0 - 1);
    if (!(a.ref == -2 || b.ref == -2 || (a.ref == 0 && a.mvx == 0 && a.mvy == 0) || (b.ref == 0 && b.mvx == 0 && b.mvy == 0)))
        predict_mv(p, 0, 0, 4, 0, 0, &mvx, &mvy);
    set_motion(p, 0, 0, 4, 4, mvx, mvy);
    m->coef_index = (uint32_t)q->coef_n;
    finish_mb_qp(p, m, 0, p->sh.qp);
}

static int parse_mb(p264parse *p, bitrd_t *b)
{
    picbuf_t *q = &p->buf[p->cur];
    p264hip_mb_t *m = &q->mb[p->mbi];
    mbcoef_t cf; cf.mask = 0;
    begin_mb(p, m);
    unsigned t = br_ue(b);
    int intra_t = -1;
    if (p->sh.type == P264_SLICE_I) intra_t = (int)t;
    else if (t >= 5) intra_t = (int)t - 5;
    int8_t *ref = q->ref + p->mbi * 4;
    uint8_t *i4 = q->i4 + p->mbi * 16;

    if (intra_t >= 0) {
        /* ---- intra (decoder/macroblock.c:117-139, 265-301) ---- */
        if (intra_t > 25) { ERR(p, "invalid mb type %d", intra_t); return -1; }
        if (intra_t == 25) { ERR(p, "unsupport i_pcm mb"); return -1; }
        memset(ref, -1, 4);
        memset(q->mv + p->mbi * 32, 0, 64);
        if (intra_t == 0) {
            m->mb_type = P264_MB_I4x4;
            for (int i = 0; i < 16; i++) {
                int pred = predict_i4mode(p, i);
                if (br_u1(b)) i4[i] = (uint8_t)pred;
                else { int rem = (int)br_u(b, 3); i4[i] = (uint8_t)(rem >= pred ? rem + 1 : rem); }
            }
        } else {
            m->mb_type = P264_MB_I16x16;
            m->intra_modes = (uint8_t)((intra_t - 1) & 3);
            m->cbp = (uint8_t)((((intra_t - 1) >> 2) % 3) << 4 | (intra_t > 12 ? 15 : 0));
            memset(i4, 2, 16);
        }
        unsigned cm = br_ue(b);
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
            for (int k = 0; k < nparts; k++) if (nref > 1) { r[k] = (int)br_te(b, nref - 1); if (r[k] >= nref) { ERR(p, "ref_idx out of range"); return -1; } }
            for (int k = 0; k < nparts; k++)
                for (int y = geo[t][k][1] >> 1; y < (geo[t][k][1] + geo[t][k][3]) >> 1; y++)
                    for (int x = geo[t][k][0] >> 1; x < (geo[t][k][0] + geo[t][k][2]) >> 1; x++) ref[y * 2 + x] = (int8_t)r[k];
            for (int k = 0; k < nparts; k++) {
                int dx = br_se(b), dy = br_se(b), px, py;
                int dir = t == 0 ? 0 : t == 1 ? 1 + k : 3 + k;
                predict_mv(p, geo[t][k][0], geo[t][k][1], geo[t][k][2], r[k], dir, &px, &py);
                set_motion(p, geo[t][k][0], geo[t][k][1], geo[t][k][2], geo[t][k][3], px + dx, py + dy);
            }
        } else {
            m->mb_type = P264_MB_P_8x8;
            int sub[4];
            for (int k = 0; k < 4; k++) { sub[k] = (int)br_ue(b); if (sub[k] > 3) { ERR(p, "invalid i_sub_partition"); return -1; } }
            for (int k = 0; k < 4; k++) {
                int r = 0;
                if (nref > 1 && t == 3) { r = (int)br_te(b, nref - 1); if (r >= nref) { ERR(p, "ref_idx out of range"); return -1; } }
                ref[k] = (int8_t)r;
            }
            for (int k = 0; k < 4; k++) {
                int ox = (k & 1) * 2, oy = (k >> 1) * 2;
                int sw = (sub[k] == 0 || sub[k] == 1) ? 2 : 1, shh = (sub[k] == 0 || sub[k] == 2) ? 2 : 1;
                for (int sy = 0; sy < 2; sy += shh)
                    for (int sx = 0; sx < 2; sx += sw) {
                        int dx = br_se(b), dy = br_se(b), px, py;
                        predict_mv(p, ox + sx, oy + sy, sw, ref[k], 0, &px, &py);
                        set_motion(p, ox + sx, oy + sy, sw, shh, px + dx, py + dy);
                    }
            }
        }
    }

    /* ---- coded_block_pattern, mb_qp_delta, residual (decoder/macroblock.c:540-587) ---- */
    if (m->mb_type != P264_MB_I16x16) {
        unsigned c = br_ue(b);
        if (c >= 48) { ERR(p, "invalid cbp"); return -1; }
        m->cbp = m->mb_type == P264_MB_I4x4 ? cbp_intra_of_code[c] : cbp_inter_of_code[c];
    }
    int qp = p->sh.qp, has_res = (m->cbp != 0 || m->mb_type == P264_MB_I16x16);
    if (has_res) {
        int dqp = br_se(b);
        if (p->opts & P264PARSE_OPT_STRICT) qp = (p->qp_pred + dqp + 52) % 52;
        else qp = p->sh.qp + dqp;                 /* delta is NOT accumulated: decoder/macroblock.c:568 */
        if (parse_residual(p, b, m, &cf) < 0) { ERR(p, "read residual data failed"); return -1; }
    } else memset(p->nnz + (size_t)p->mbi * 24, 0, 24);
    if (store_coefs(p, m, &cf) < 0) return -1;
    finish_mb_qp(p, m, has_res, qp);
    if (br_overrun(b)) { ERR(p, "macroblock overruns the slice data"); return -1; }
    return 0;
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
        p->n_list0 = 0;
    } else {
        if (sh.first_mb != p->next_mb) { ERR(p, "slice starts at MB %d, expected %d", sh.first_mb, p->next_mb); return -1; }
        p->slice_no++;
    }
    if (sh.disable_deblock != 1) {                /* the filter parameters are per picture on the device */
        if (!p->pic_deblock) { p->pic_deblock = 1; p->pic_alpha = sh.alpha_off; p->pic_beta = sh.beta_off; }
        else if (sh.alpha_off != p->pic_alpha || sh.beta_off != p->pic_beta) { ERR(p, "per-slice deblocking offsets unsupported"); return -1; }
    }
    p->sh = sh;
    if (sh.type == P264_SLICE_P) {
        int prev[P264HIP_MAX_REFS], n_prev = p->n_list0;
        memcpy(prev, p->list0, sizeof prev);
        if (build_list0(p, &sh) < 0) return -1;
        /* reference indices are resolved through ONE list per picture on the device */
        if (n_prev && (n_prev != p->n_list0 || memcmp(prev, p->list0, sizeof(int) * (size_t)n_prev))) { ERR(p, "slices of one picture with different reference lists unsupported"); return -1; }
        p->sh0.type = P264_SLICE_P;               /* a picture with any P slice is reconstructed as P */
    }
    p->qp_pred = sh.qp;

    long stop = rbsp_stop_bit(payload, size);
    p->skip_run = -1;
    while (p->next_mb < p->n_mb) {
        p->mbi = p->next_mb; p->mbx = p->mbi % p->mb_w; p->mby = p->mbi / p->mb_w;
        if (p->skip_run <= 0 && (long)b.consumed >= stop) break;   /* !more_rbsp_data(): the slice ends here */
        if (sh.type == P264_SLICE_P && p->skip_run < 0) {
            p->skip_run = (int)br_ue(&b);
            if (p->skip_run > p->n_mb - p->next_mb) { ERR(p, "mb_skip_run %d runs past the picture", p->skip_run); p->pic_open = 0; return -1; }
        }
        if (p->skip_run > 0) {
            decode_pskip(p);
            p->skip_run--;
        } else {
            if ((long)b.consumed >= stop) break;
            if (parse_mb(p, &b) < 0) { ERR(p, "macroblock read failed [%d,%d]", p->mbx, p->mby); p->pic_open = 0; return -1; }
            p->skip_run = -1;
        }
        p->slice_of[p->mbi] = (uint16_t)p->slice_no;
        p->next_mb++;
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
