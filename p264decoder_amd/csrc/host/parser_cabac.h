/* parser_cabac.h - the CABAC macroblock layer (included by parser.c): binarisation and context selection of H.264 9.3.2 /
 * 9.3.3.1 for every syntax element of I, P and B frame macroblocks, on the arithmetic decoding engine of cabac.h.
 *
 * The reference has an engine (core/cabac.c:842-902) and no macroblock layer on top of it: its CABAC macroblock read is a
 * stub (decoder/macroblock.c:594-597).  So this is written from the standard; what pins it: the engine by the reference's
 * encoder (tests/test_cabac_kat.py), the layer by the stream writer, which carries its own arithmetic ENCODER and its own
 * context selection (tools/synth264_cabac.h) - tests/test_cabac_streams.py: the same syntax written with CAVLC and with CABAC
 * must parse to the same pictures.  Whole-stream parity with the reference is unpinned (it cannot decode such streams).
 *
 * The semantic half of a macroblock (vector prediction, direct prediction, coefficient storage, QP bookkeeping) is the
 * CAVLC code: parser.c reads every syntax element through the rd_* functions below, which pick the entropy decoder.
 */

/* what later macroblocks' context selection needs to know about a macroblock (p->cinfo[]) */
#define CI_SKIP      0x0001u        /* P_Skip / B_Skip */
#define CI_DIRECT16  0x0002u        /* B_Skip or B_Direct_16x16 */
#define CI_DC_Y      0x0004u        /* coded_block_flag of the Intra16x16 DC block, the Cb DC block, the Cr DC block */
#define CI_DC_CB     0x0008u
#define CI_DC_CR     0x0010u
#define CI_D8(q)     (0x0100u << (q))   /* 8x8 quadrant q is direct-predicted */

static inline int cb_left(const p264parse *p) { return p->cur_avail & P264_AVAIL_LEFT; }
static inline int cb_top(const p264parse *p)  { return p->cur_avail & P264_AVAIL_TOP; }

/* ---- macroblock types ---------------------------------------------------------------------------------------------------- */
static int cb_mb_skip_flag(p264parse *p)
{
    int ctx = p->sh.type == P264_SLICE_B ? 24 : 11;
    if (cb_left(p) && !(p->cinfo[p->mbi - 1] & CI_SKIP)) ctx++;
    if (cb_top(p) && !(p->cinfo[p->mbi - p->mb_w] & CI_SKIP)) ctx++;
    return p264cabac_decision(&p->cb, ctx);
}
/* mb_type of an intra macroblock in I-slice numbering (0 I_NxN, 1..24 Intra16x16, 25 I_PCM): in I slices the first bin has
 * neighbour contexts 3..5 and the rest 6..10; as the suffix of a P / B type, contexts base..base+3 without neighbours */
static int cb_intra_mb_type(p264parse *p, int base, int i_slice)
{
    const picbuf_t *q = &p->buf[p->cur];
    int s = base;
    if (i_slice) {
        int ctx = 3;
        if (cb_left(p) && q->mb[p->mbi - 1].mb_type != P264_MB_I4x4) ctx++;
        if (cb_top(p) && q->mb[p->mbi - p->mb_w].mb_type != P264_MB_I4x4) ctx++;
        if (!p264cabac_decision(&p->cb, ctx)) return 0;
        s = 3 + 2;
    } else if (!p264cabac_decision(&p->cb, s)) return 0;
    if (p264cabac_terminate(&p->cb)) return 25;
    int t = 1;
    t += 12 * p264cabac_decision(&p->cb, s + 1);                                     /* luma coded */
    if (p264cabac_decision(&p->cb, s + 2)) t += 4 + 4 * p264cabac_decision(&p->cb, s + 2 + i_slice);   /* chroma: 0, 1, 2 */
    t += 2 * p264cabac_decision(&p->cb, s + 3 + i_slice);                           /* prediction mode */
    t += p264cabac_decision(&p->cb, s + 3 + 2 * i_slice);
    return t;
}
/* mb_type as the CAVLC ue(v) would give it: I slices 0..25, P slices 0..4 and 5 + intra, B slices 0..22 and 23 + intra */
static int cb_mb_type(p264parse *p)
{
    p264cabac_t *c = &p->cb;
    if (p->sh.type == P264_SLICE_I) return cb_intra_mb_type(p, 3, 1);
    if (p->sh.type == P264_SLICE_P) {
        if (p264cabac_decision(c, 14)) return 5 + cb_intra_mb_type(p, 17, 0);
        if (!p264cabac_decision(c, 15)) return 3 * p264cabac_decision(c, 16);       /* 16x16, or 8x8 */
        return 2 - p264cabac_decision(c, 17);                                       /* 8x16, or 16x8 */
    }
    int ctx = 27;
    if (cb_left(p) && !(p->cinfo[p->mbi - 1] & CI_DIRECT16)) ctx++;
    if (cb_top(p) && !(p->cinfo[p->mbi - p->mb_w] & CI_DIRECT16)) ctx++;
    if (!p264cabac_decision(c, ctx)) return 0;                                      /* B_Direct_16x16 */
    if (!p264cabac_decision(c, 27 + 3)) return 1 + p264cabac_decision(c, 27 + 5);   /* B_L0_16x16, B_L1_16x16 */
    int bits = p264cabac_decision(c, 27 + 4) << 3;
    bits |= p264cabac_decision(c, 27 + 5) << 2;
    bits |= p264cabac_decision(c, 27 + 5) << 1;
    bits |= p264cabac_decision(c, 27 + 5);
    if (bits < 8) return bits + 3;                                                  /* B_Bi_16x16 .. B_L1_L0_16x8 */
    if (bits == 13) return 23 + cb_intra_mb_type(p, 32, 0);
    if (bits == 14) return 11;                                                      /* B_L1_L0_8x16 */
    if (bits == 15) return 22;                                                      /* B_8x8 */
    bits = bits << 1 | p264cabac_decision(c, 27 + 5);
    return bits - 4;                                                                /* B_L0_Bi_16x8 .. B_Bi_Bi_8x16 */
}
static int cb_sub_mb_type(p264parse *p)
{
    p264cabac_t *c = &p->cb;
    if (p->sh.type == P264_SLICE_P) {
        if (p264cabac_decision(c, 21)) return 0;                                    /* 8x8 */
        if (!p264cabac_decision(c, 22)) return 1;                                   /* 8x4 */
        return p264cabac_decision(c, 23) ? 2 : 3;                                   /* 4x8, 4x4 */
    }
    if (!p264cabac_decision(c, 36)) return 0;                                       /* direct */
    if (!p264cabac_decision(c, 37)) return 1 + p264cabac_decision(c, 39);           /* 8x8 from list 0 / list 1 */
    int t = 3;
    if (p264cabac_decision(c, 38)) {
        if (p264cabac_decision(c, 39)) return 11 + p264cabac_decision(c, 39);       /* 4x4 from list 1 / both */
        t += 4;
    }
    t += 2 * p264cabac_decision(c, 39);
    t += p264cabac_decision(c, 39);
    return t;
}

/* ---- prediction --------------------------------------------------------------------------------------------------------- */
/* is the 4x4 block at picture position (x4, y4) - left of / above the partition being read - available, and which
 * macroblock and block is it?  (blocks of the current macroblock left of / above a partition are always decoded already) */
static int cb_block(const p264parse *p, int x4, int y4, int *mbi, int *blk)
{
    if (x4 < 0 || y4 < 0) return 0;
    const int mx = x4 >> 2, my = y4 >> 2;
    if (mx >= p->mb_w || my >= p->mb_h) return 0;
    const int i = my * p->mb_w + mx;
    if (i != p->mbi && !(i < p->mbi && p->slice_of[i] == (uint16_t)p->slice_no)) return 0;
    *mbi = i; *blk = (y4 & 3) * 4 + (x4 & 3);
    return 1;
}
static int cb_ref_idx(p264parse *p, int list, int bx, int by)
{
    const picbuf_t *q = &p->buf[p->cur];
    const int8_t *ref = list ? q->ref1 : q->ref;
    const int x0 = p->mbx * 4 + bx, y0 = p->mby * 4 + by;
    int ctx = 0, mbi, blk;
    for (int n = 0; n < 2; n++) {
        if (!cb_block(p, n ? x0 : x0 - 1, n ? y0 - 1 : y0, &mbi, &blk)) continue;
        const int q8 = ((blk >> 2) & 2) | ((blk >> 1) & 1);                          /* the block's 8x8 quadrant */
        if (P264_MB_IS_INTRA(q->mb[mbi].mb_type) || (p->cinfo[mbi] & CI_D8(q8))) continue;
        if (ref[mbi * 4 + q8] > 0) ctx += n ? 2 : 1;
    }
    int v = 0;
    while (p264cabac_decision(&p->cb, 54 + ctx)) {
        ctx = (ctx >> 2) + 4;
        if (++v >= 32) return -1;
    }
    return v;
}
static int cb_mvd_comp(p264parse *p, int base, int sum)
{
    p264cabac_t *c = &p->cb;
    if (!p264cabac_decision(c, base + (sum < 3 ? 0 : sum > 32 ? 2 : 1))) return 0;
    int v = 1, ctx = base + 3;
    while (v < 9 && p264cabac_decision(c, ctx)) { if (v < 4) ctx++; v++; }
    if (v >= 9) {                                                                  /* UEG3 suffix */
        int k = 3;
        while (p264cabac_bypass(c)) { v += 1 << k; if (++k > 24) return 1 << 30; }
        while (k--) v += p264cabac_bypass(c) << k;
    }
    return p264cabac_bypass(c) ? -v : v;
}
static int cb_mvd(p264parse *p, int list, int bx, int by, int bw, int bh, int *dx, int *dy)
{
    uint8_t *ma = p->mvd_abs[list];
    const int x0 = p->mbx * 4 + bx, y0 = p->mby * 4 + by;
    int sum[2] = { 0, 0 }, mbi, blk;
    if (cb_block(p, x0 - 1, y0, &mbi, &blk)) { sum[0] += ma[(mbi * 16 + blk) * 2]; sum[1] += ma[(mbi * 16 + blk) * 2 + 1]; }
    if (cb_block(p, x0, y0 - 1, &mbi, &blk)) { sum[0] += ma[(mbi * 16 + blk) * 2]; sum[1] += ma[(mbi * 16 + blk) * 2 + 1]; }
    *dx = cb_mvd_comp(p, 40, sum[0]);
    *dy = cb_mvd_comp(p, 47, sum[1]);
    if (*dx == 1 << 30 || *dy == 1 << 30 || *dx < -(1 << 20) || *dx > (1 << 20) || *dy < -(1 << 20) || *dy > (1 << 20)) return -1;
    const int ax = *dx < 0 ? -*dx : *dx, ay = *dy < 0 ? -*dy : *dy;
    for (int y = by; y < by + bh; y++)
        for (int x = bx; x < bx + bw; x++) {
            ma[(p->mbi * 16 + y * 4 + x) * 2] = (uint8_t)(ax > 255 ? 255 : ax);
            ma[(p->mbi * 16 + y * 4 + x) * 2 + 1] = (uint8_t)(ay > 255 ? 255 : ay);
        }
    return 0;
}
static int cb_intra4x4_mode(p264parse *p, int pred)
{
    p264cabac_t *c = &p->cb;
    if (p264cabac_decision(c, 68)) return pred;
    int m = p264cabac_decision(c, 69);
    m += 2 * p264cabac_decision(c, 69);
    m += 4 * p264cabac_decision(c, 69);
    return m >= pred ? m + 1 : m;
}
static int cb_chroma_pred_mode(p264parse *p)
{
    const picbuf_t *q = &p->buf[p->cur];
    p264cabac_t *c = &p->cb;
    int ctx = 64;
    if (cb_left(p) && P264_MB_IS_INTRA(q->mb[p->mbi - 1].mb_type) && (q->mb[p->mbi - 1].intra_modes >> 4)) ctx++;
    if (cb_top(p) && P264_MB_IS_INTRA(q->mb[p->mbi - p->mb_w].mb_type) && (q->mb[p->mbi - p->mb_w].intra_modes >> 4)) ctx++;
    if (!p264cabac_decision(c, ctx)) return 0;
    if (!p264cabac_decision(c, 64 + 3)) return 1;
    return p264cabac_decision(c, 64 + 3) ? 3 : 2;
}
static int cb_cbp(p264parse *p)
{
    const picbuf_t *q = &p->buf[p->cur];
    p264cabac_t *c = &p->cb;
    /* neighbours' patterns; an unavailable neighbour counts as "coded" for luma (condition 0) and "not coded" for chroma */
    const int L = cb_left(p) != 0, T = cb_top(p) != 0;
    const int cl = L ? q->mb[p->mbi - 1].cbp : 0x0f, ct = T ? q->mb[p->mbi - p->mb_w].cbp : 0x0f;
    int cbp = 0;
    for (int b8 = 0; b8 < 4; b8++) {
        const int a = (b8 & 1) ? (cbp >> (b8 - 1)) & 1 : (cl >> (b8 + 1)) & 1;         /* left 8x8: inside this macroblock, or the left one's right column */
        const int b = (b8 & 2) ? (cbp >> (b8 - 2)) & 1 : (ct >> (b8 + 2)) & 1;         /* upper 8x8 */
        cbp |= p264cabac_decision(c, 73 + !a + 2 * !b) << b8;
    }
    const int ccl = L ? q->mb[p->mbi - 1].cbp >> 4 : 0, cct = T ? q->mb[p->mbi - p->mb_w].cbp >> 4 : 0;
    if (p264cabac_decision(c, 77 + (ccl != 0) + 2 * (cct != 0)))
        cbp |= (1 + p264cabac_decision(c, 77 + 4 + (ccl == 2) + 2 * (cct == 2))) << 4;
    return cbp;
}
static int cb_mb_qp_delta(p264parse *p)
{
    p264cabac_t *c = &p->cb;
    int ctx = p->last_dqp != 0, v = 0;
    while (p264cabac_decision(c, 60 + ctx)) { ctx = 2 + (ctx >> 1); if (++v > 104) return 1 << 30; }
    return (v & 1) ? (v + 1) >> 1 : -((v + 1) >> 1);
}

/* ---- residual ----------------------------------------------------------------------------------------------------------- */
/* ctxBlockCat: 0 Intra16x16 DC, 1 Intra16x16 AC, 2 luma 4x4, 3 chroma DC, 4 chroma AC.  Levels come out in scan order
 * (out[0..n-1]), like cavlc_read_block.  nza / nzb: coded_block_flag of the left / upper block of the same kind. */
static int cb_residual_block(p264parse *p, int cat, int nza, int nzb, int16_t *out)
{
    static const uint8_t n_of[5] = { 16, 15, 16, 4, 15 }, sig_off[5] = { 0, 15, 29, 44, 47 }, abs_off[5] = { 0, 10, 20, 30, 39 };
    p264cabac_t *c = &p->cb;
    if (!p264cabac_decision(c, 85 + 4 * cat + nza + 2 * nzb)) return 0;
    const int n = n_of[cat];
    int pos[16], cnt = 0, i;
    for (i = 0; i < n - 1; i++) {
        const int k = cat == 3 ? (i < 2 ? i : 2) : i;
        if (!p264cabac_decision(c, 105 + sig_off[cat] + k)) continue;
        pos[cnt++] = i;
        if (p264cabac_decision(c, 166 + sig_off[cat] + k)) break;
    }
    if (i == n - 1) pos[cnt++] = n - 1;                     /* no "last" seen: the final coefficient is significant */
    int eq1 = 0, gt1 = 0;
    for (int k = cnt - 1; k >= 0; k--) {
        int ctx = 227 + abs_off[cat] + (gt1 ? 0 : (eq1 < 3 ? 1 + eq1 : 4));
        int a = 1;
        if (p264cabac_decision(c, ctx)) {
            ctx = 227 + abs_off[cat] + 5 + (gt1 < 4 - (cat == 3) ? gt1 : 4 - (cat == 3));
            a = 2;
            while (a < 15 && p264cabac_decision(c, ctx)) a++;
            if (a >= 15) {                                  /* Exp-Golomb (k = 0) escape */
                int j = 0;
                while (p264cabac_bypass(c)) { a += 1 << j; if (++j > 16) return -1; }
                while (j--) a += p264cabac_bypass(c) << j;
            }
            gt1++;
        } else eq1++;
        if (a > 32767) return -1;
        out[pos[k]] = (int16_t)(p264cabac_bypass(c) ? -a : a);
    }
    return cnt;
}
/* coded_block_flag of the block left of / above block blk (0..15 luma, 16..23 chroma AC) */
static void cb_nz_neighbours(const p264parse *p, int blk, int intra, int *nza, int *nzb)
{
    const uint8_t *cur = p->nnz + (size_t)p->mbi * 24;
    int a, b;
    if (blk < 16) {
        const int x = blk_x[blk], y = blk_y[blk];
        a = x > 0 ? cur[blk_of_xy[y][x-1]] : cb_left(p) ? (cur - 24)[blk_of_xy[y][3]] : -1;
        b = y > 0 ? cur[blk_of_xy[y-1][x]] : cb_top(p) ? (cur - 24 * p->mb_w)[blk_of_xy[3][x]] : -1;
    } else {
        const int base = blk < 20 ? 16 : 20, k = blk - base, x = k & 1, y = k >> 1;
        a = x > 0 ? cur[blk - 1] : cb_left(p) ? (cur - 24)[base + y * 2 + 1] : -1;
        b = y > 0 ? cur[blk - 2] : cb_top(p) ? (cur - 24 * p->mb_w)[base + 2 + x] : -1;
    }
    *nza = a < 0 ? intra : a > 0;                           /* no neighbour: 1 for intra macroblocks, 0 for inter ones */
    *nzb = b < 0 ? intra : b > 0;
}
static void cb_dc_neighbours(const p264parse *p, unsigned bit, int intra, int *nza, int *nzb)
{
    *nza = cb_left(p) ? (p->cinfo[p->mbi - 1] & bit) != 0 : intra;
    *nzb = cb_top(p) ? (p->cinfo[p->mbi - p->mb_w] & bit) != 0 : intra;
}
/* residual( ) with CABAC: the same blocks in the same order as parse_residual */
static int parse_residual_cabac(p264parse *p, p264hip_mb_t *m, mbcoef_t *cf)
{
    uint8_t *nnz = p->nnz + (size_t)p->mbi * 24;
    const int cbp_l = m->cbp & 15, cbp_c = m->cbp >> 4, intra = P264_MB_IS_INTRA(m->mb_type), i16 = m->mb_type == P264_MB_I16x16;
    int tc, a, b;
    if (i16) {
        memset(cf->dc_luma, 0, sizeof cf->dc_luma);
        cb_dc_neighbours(p, CI_DC_Y, intra, &a, &b);
        if ((tc = cb_residual_block(p, 0, a, b, cf->dc_luma)) < 0) return -1;
        if (tc) { cf->mask |= P264_COEF_LUMA_DC; p->cinfo[p->mbi] |= CI_DC_Y; }
    }
    for (int i = 0; i < 16; i++) {
        nnz[i] = 0;
        if (!(cbp_l & (1 << (i >> 2)))) continue;
        memset(cf->blk[i], 0, sizeof cf->blk[i]);
        cb_nz_neighbours(p, i, intra, &a, &b);
        if ((tc = cb_residual_block(p, i16 ? 1 : 2, a, b, cf->blk[i])) < 0) return -1;
        nnz[i] = (uint8_t)tc;
        if (tc) cf->mask |= 1u << i;
    }
    if (cbp_c) {
        memset(cf->dc_chroma, 0, sizeof cf->dc_chroma);
        int t0, t1;
        cb_dc_neighbours(p, CI_DC_CB, intra, &a, &b);
        if ((t0 = cb_residual_block(p, 3, a, b, cf->dc_chroma)) < 0) return -1;
        if (t0) p->cinfo[p->mbi] |= CI_DC_CB;
        cb_dc_neighbours(p, CI_DC_CR, intra, &a, &b);
        if ((t1 = cb_residual_block(p, 3, a, b, cf->dc_chroma + 4)) < 0) return -1;
        if (t1) p->cinfo[p->mbi] |= CI_DC_CR;
        if (t0 | t1) cf->mask |= P264_COEF_CHROMA_DC;
    }
    for (int i = 16; i < 24; i++) {
        nnz[i] = 0;
        if (!(cbp_c & 2)) continue;
        memset(cf->blk[i], 0, sizeof cf->blk[i]);
        cb_nz_neighbours(p, i, intra, &a, &b);
        if ((tc = cb_residual_block(p, 4, a, b, cf->blk[i])) < 0) return -1;
        nnz[i] = (uint8_t)tc;
        if (tc) cf->mask |= 1u << i;
    }
    return 0;
}

/* ---- the syntax elements as the macroblock parsers ask for them: CAVLC or CABAC ---------------------------------------- */
static int rd_ref_idx(p264parse *p, bitrd_t *b, int list, int bx, int by, int n_active)
{
    if (n_active <= 1) return 0;
    return p->cabac_on ? cb_ref_idx(p, list, bx, by) : (int)br_te(b, n_active - 1);
}
static int rd_mvd(p264parse *p, bitrd_t *b, int list, int bx, int by, int bw, int bh, int *dx, int *dy)
{
    if (p->cabac_on) return cb_mvd(p, list, bx, by, bw, bh, dx, dy);
    *dx = br_se(b); *dy = br_se(b);
    return 0;
}
static int rd_sub_mb_type(p264parse *p, bitrd_t *b) { return p->cabac_on ? cb_sub_mb_type(p) : (int)br_ue(b); }
static int rd_intra4x4_mode(p264parse *p, bitrd_t *b, int pred)
{
    if (p->cabac_on) return cb_intra4x4_mode(p, pred);
    if (br_u1(b)) return pred;
    const int rem = (int)br_u(b, 3);
    return rem >= pred ? rem + 1 : rem;
}
static unsigned rd_chroma_pred_mode(p264parse *p, bitrd_t *b) { return p->cabac_on ? (unsigned)cb_chroma_pred_mode(p) : br_ue(b); }
/* coded_block_pattern, already mapped (the CAVLC code number goes through the intra / inter table) */
static int rd_cbp(p264parse *p, bitrd_t *b, int intra4x4)
{
    if (p->cabac_on) return cb_cbp(p);
    const unsigned c = br_ue(b);
    if (c >= 48) return -1;
    return intra4x4 ? cbp_intra_of_code[c] : cbp_inter_of_code[c];
}
static int rd_mb_qp_delta(p264parse *p, bitrd_t *b)
{
    const int v = p->cabac_on ? cb_mb_qp_delta(p) : br_se(b);
    p->last_dqp = v;
    return v;
}
static int rd_residual(p264parse *p, bitrd_t *b, p264hip_mb_t *m, mbcoef_t *cf);
