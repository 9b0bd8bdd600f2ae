/* synth264_b.h - the B-picture half of the stream writer (included by synth264.c; --bframes N).
 *
 * Writes CAVLC B slices of H.264 7.3.3 / 7.3.5 with random syntax and keeps its OWN model of what a decoder has to derive
 * from them (so that --dump-mv can record the intended result): the two reference lists in picture-order-count order
 * (8.2.4.2.3), list-wise motion vector prediction, spatial and temporal direct prediction (8.4.1.2) with the co-located
 * motion of the reference pictures, implicit bi-prediction weights (8.4.2.3.1).  Written from the standard, independently of
 * the decoder's parser (csrc/host/parser.c): the parser test compares the two.
 * B pictures are non-reference pictures here; the reference pictures around them are the P / I pictures of the ordinary writer.
 */

/* ---- the picture's lists -------------------------------------------------------------------------------------------- */
static void b_build_lists(void)
{
    for (int l = 0; l < 2; l++) {
        int n = 0;
        /* first the pictures on the list's own side of the current one, nearest first, then the others, nearest first */
        for (int side = 0; side < 2; side++) {
            const int want_before = (l == 0) == (side == 0);
            for (;;) {
                int best = -1;
                for (int i = 0; i < 8; i++) {
                    if (!wdpb[i].used || wdpb[i].is_long) continue;
                    if ((wdpb[i].poc < cur_poc) != want_before) continue;
                    int taken = 0;
                    for (int k = 0; k < n; k++) if (blist[l][k] == i) taken = 1;
                    if (taken) continue;
                    const int di = wdpb[i].poc > cur_poc ? wdpb[i].poc - cur_poc : cur_poc - wdpb[i].poc;
                    if (best < 0 || di < (wdpb[best].poc > cur_poc ? wdpb[best].poc - cur_poc : cur_poc - wdpb[best].poc)) best = i;
                }
                if (best < 0) break;
                blist[l][n++] = best;
            }
        }
        blist_n[l] = n;
    }
    if (blist_n[1] > 1 && blist_n[0] == blist_n[1] && !memcmp(blist[0], blist[1], sizeof(int) * (size_t)blist_n[0])) {
        const int t = blist[1][0]; blist[1][0] = blist[1][1]; blist[1][1] = t;
    }
}

static int b_clip(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
/* DistScaleFactor of 8.4.1.2.3 for (picture in list 0, picture in list 1); 9999 = "do not scale" */
static int b_dist_scale(const wdpb_t *p0, const wdpb_t *p1)
{
    const int tb = b_clip(cur_poc - p0->poc, -128, 127), td = b_clip(p1->poc - p0->poc, -128, 127);
    if (td == 0 || p0->is_long) return 9999;
    const int tx = (16384 + (td < 0 ? -td : td) / 2) / td;
    return b_clip((tb * tx + 32) >> 6, -1024, 1023);
}
static int b_weight(int r0, int r1)
{
    if (!opt_implicit) return 32;
    const wdpb_t *p0 = &wdpb[blist[0][r0]], *p1 = &wdpb[blist[1][r1]];
    if (p0->is_long || p1->is_long) return 32;
    const int dsf = b_dist_scale(p0, p1);
    if (dsf == 9999) return 32;
    const int w1 = dsf >> 2;
    return (w1 < -64 || w1 > 128) ? 32 : 64 - w1;
}

/* ---- direct prediction ------------------------------------------------------------------------------------------------ */
typedef struct { int ref[2][16]; int mx[2][16], my[2][16]; } bdirect_t;       /* per 4x4 block */

static int b_minpos(int a, int b) { if (a < 0) return b; if (b < 0) return a; return a < b ? a : b; }

static void b_direct(int mbx, int mby, bdirect_t *d)
{
    const wdpb_t *col = &wdpb[blist[1][0]];
    if (!opt_temporal) {
        int ref[2], px[2] = { 0, 0 }, py[2] = { 0, 0 };
        for (int l = 0; l < 2; l++) {
            nb_t a = nb_motion_l(mbx * 4 - 1, mby * 4, l), b = nb_motion_l(mbx * 4, mby * 4 - 1, l), c = nb_motion_l(mbx * 4 + 4, mby * 4 - 1, l);
            if (c.ref == -2) c = nb_motion_l(mbx * 4 - 1, mby * 4 - 1, l);
            ref[l] = b_minpos(a.ref, b_minpos(b.ref, c.ref));
            if (ref[l] < 0) ref[l] = -1;
        }
        const int none = ref[0] < 0 && ref[1] < 0;
        if (none) { ref[0] = 0; ref[1] = 0; }
        else for (int l = 0; l < 2; l++) if (ref[l] >= 0) predict_mv_l(mbx, mby, 0, 0, 4, 0, ref[l], &px[l], &py[l], l);
        for (int k = 0; k < 16; k++) {
            /* the co-located block that speaks for block k: itself, or the corner block of its quadrant (direct_8x8_inference) */
            const int x = k & 3, y = k >> 2, ck = opt_d8inf ? (y < 2 ? 0 : 3) * 4 + (x < 2 ? 0 : 3) : k;
            const int still = !col->is_long && col->cref[cur * 16 + ck] == 0 &&
                              col->cmv[(cur * 16 + ck) * 2] >= -1 && col->cmv[(cur * 16 + ck) * 2] <= 1 &&
                              col->cmv[(cur * 16 + ck) * 2 + 1] >= -1 && col->cmv[(cur * 16 + ck) * 2 + 1] <= 1;
            for (int l = 0; l < 2; l++) {
                d->ref[l][k] = ref[l];
                const int zero = none || ref[l] < 0 || (ref[l] == 0 && still);
                d->mx[l][k] = zero ? 0 : px[l]; d->my[l][k] = zero ? 0 : py[l];
            }
        }
        return;
    }
    for (int k = 0; k < 16; k++) {
        const int x = k & 3, y = k >> 2, ck = opt_d8inf ? (y < 2 ? 0 : 3) * 4 + (x < 2 ? 0 : 3) : k;
        const int cr = col->cref[cur * 16 + ck];
        int r0 = 0, m0x = 0, m0y = 0, m1x = 0, m1y = 0;
        if (cr >= 0) {
            const int want = col->cpic[cur * 16 + ck];
            r0 = -1;
            for (int i = 0; i < n_active && r0 < 0; i++) if (wdpb[blist[0][i < blist_n[0] ? i : blist_n[0] - 1]].pic == want) r0 = i;
            if (r0 < 0) r0 = 0;
            const int cx = col->cmv[(cur * 16 + ck) * 2], cy = col->cmv[(cur * 16 + ck) * 2 + 1];
            const int dsf = b_dist_scale(&wdpb[blist[0][r0 < blist_n[0] ? r0 : blist_n[0] - 1]], col);
            if (dsf == 9999) { m0x = cx; m0y = cy; }
            else { m0x = (dsf * cx + 128) >> 8; m0y = (dsf * cy + 128) >> 8; m1x = m0x - cx; m1y = m0y - cy; }
        }
        d->ref[0][k] = r0; d->ref[1][k] = 0;
        d->mx[0][k] = m0x; d->my[0][k] = m0y; d->mx[1][k] = m1x; d->my[1][k] = m1y;
    }
}
/* blocks (bx..bx+bw, by..by+bh) of the current macroblock take their direct motion for list l */
static void b_apply_direct(const bdirect_t *d, int bx, int by, int bw, int bh, int l)
{
    for (int y = by; y < by + bh; y++)
        for (int x = bx; x < bx + bw; x++) {
            const int k = y * 4 + x, on = d->ref[l][k] >= 0;
            set_mv_l(x, y, 1, 1, on ? d->mx[l][k] : 0, on ? d->my[l][k] : 0, on ? d->ref[l][k] : -1, l);
        }
}

/* ---- macroblocks ------------------------------------------------------------------------------------------------------- */
static void b_skip(int mbx, int mby)
{
    bdirect_t d;
    mb_type[cur] = T_P;
    memset(i4m + cur * 16, 2, 16);
    memset(nnz + (size_t)cur * 24, 0, 24);
    b_direct(mbx, mby, &d);
    b_apply_direct(&d, 0, 0, 4, 4, 0);
    b_apply_direct(&d, 0, 0, 4, 4, 1);
}

static void b_pick_mv(int mbx, int mby, int bx, int by, int bw, int bh, int px, int py, int *mx, int *my)
{
    if (pct(25) && mv_ok(mbx, mby, bx, by, bw, bh, px, py)) { *mx = px; *my = py; }
    else random_mv(mbx, mby, bx, by, bw, bh, mx, my);
}

static void put_b_mb(bw_t *b, int mbx, int mby)
{
    const int nact[2] = { n_active, n_active1 };
    mb_type[cur] = T_P;
    memset(i4m + cur * 16, 2, 16);
    const int pick = rnd(100);
    if (pick < 12) {                                                    /* B_Direct_16x16 */
        bdirect_t d;
        sx_mb_type(b, 0);
        b_direct(mbx, mby, &d);
        b_apply_direct(&d, 0, 0, 4, 4, 0);
        b_apply_direct(&d, 0, 0, 4, 4, 1);
    } else if (pick < 72) {                                             /* one or two partitions */
        /* table 7-14: 1..3 = 16x16 from list 0 / list 1 / both; 4..21 = pairs of 16x8 (even) or 8x16 (odd) partitions */
        static const int pairs[9][2] = { {1,1}, {2,2}, {1,2}, {2,1}, {1,3}, {2,3}, {3,1}, {3,2}, {3,3} };   /* 1 list 0, 2 list 1, 3 both */
        const int t = pick < 45 ? 1 + rnd(3) : 4 + rnd(18);
        int np, g[2][4], uses[2];
        if (t <= 3) { np = 1; g[0][0] = 0; g[0][1] = 0; g[0][2] = 4; g[0][3] = 4; uses[0] = t; uses[1] = 0; }
        else {
            np = 2;
            for (int k = 0; k < 2; k++) {
                if (t & 1) { g[k][0] = 2 * k; g[k][1] = 0; g[k][2] = 2; g[k][3] = 4; }
                else       { g[k][0] = 0; g[k][1] = 2 * k; g[k][2] = 4; g[k][3] = 2; }
                uses[k] = pairs[(t - 4) / 2][k];
            }
        }
        sx_mb_type(b, t);
        int r[2][2];
        for (int l = 0; l < 2; l++)
            for (int k = 0; k < np; k++) {
                r[l][k] = -1;
                if (!(uses[k] & (1 << l))) continue;
                r[l][k] = nact[l] > 1 ? (pct(55) ? 0 : rnd(nact[l])) : 0;
                sx_ref_idx(b, l, g[k][0], g[k][1], g[k][2], g[k][3], nact[l], r[l][k]);
            }
        for (int l = 0; l < 2; l++)
            for (int k = 0; k < np; k++) {
                if (r[l][k] < 0) { set_mv_l(g[k][0], g[k][1], g[k][2], g[k][3], 0, 0, -1, l); continue; }
                int px, py, mx, my;
                const int dir = np == 1 ? 0 : (t & 1) ? 3 + k : 1 + k;
                predict_mv_l(mbx, mby, g[k][0], g[k][1], g[k][2], dir, r[l][k], &px, &py, l);
                b_pick_mv(mbx, mby, g[k][0], g[k][1], g[k][2], g[k][3], px, py, &mx, &my);
                sx_mvd(b, l, g[k][0], g[k][1], g[k][2], g[k][3], mx - px, my - py);
                set_mv_l(g[k][0], g[k][1], g[k][2], g[k][3], mx, my, r[l][k], l);
            }
    } else {                                                            /* B_8x8 */
        /* table 7-18: 0 direct; 1..3 8x8 list 0 / 1 / both; 4,5 8x4 4x8 list 0; 6,7 list 1; 8,9 both; 10..12 4x4 list 0 / 1 / both */
        static const int s_uses[13] = { 0, 1, 2, 3, 1, 1, 2, 2, 3, 3, 1, 2, 3 };
        static const int s_w[13] = { 2, 2, 2, 2, 2, 1, 2, 1, 2, 1, 1, 1, 1 }, s_h[13] = { 2, 2, 2, 2, 1, 2, 1, 2, 1, 2, 1, 1, 1 };
        int sub[4], any_direct = 0;
        sx_mb_type(b, 22);
        for (int k = 0; k < 4; k++) { sub[k] = pct(25) ? 0 : 1 + rnd(opt_sub8x8 ? 12 : 3); sx_sub_mb_type(b, k, sub[k]); any_direct |= !sub[k]; }
        bdirect_t d;
        if (any_direct) b_direct(mbx, mby, &d);
        int r[2][4];
        for (int l = 0; l < 2; l++)
            for (int k = 0; k < 4; k++) {
                r[l][k] = -1;
                if (!(s_uses[sub[k]] & (1 << l))) continue;
                r[l][k] = nact[l] > 1 ? (pct(55) ? 0 : rnd(nact[l])) : 0;
                sx_ref_idx(b, l, (k & 1) * 2, (k >> 1) * 2, 2, 2, nact[l], r[l][k]);
            }
        for (int l = 0; l < 2; l++)
            for (int k = 0; k < 4; k++) {
                const int ox = (k & 1) * 2, oy = (k >> 1) * 2;
                if (!sub[k]) { b_apply_direct(&d, ox, oy, 2, 2, l); continue; }
                if (r[l][k] < 0) { set_mv_l(ox, oy, 2, 2, 0, 0, -1, l); continue; }
                for (int sy = 0; sy < 2; sy += s_h[sub[k]])
                    for (int sx = 0; sx < 2; sx += s_w[sub[k]]) {
                        int px, py, mx, my;
                        predict_mv_l(mbx, mby, ox + sx, oy + sy, s_w[sub[k]], 0, r[l][k], &px, &py, l);
                        b_pick_mv(mbx, mby, ox + sx, oy + sy, s_w[sub[k]], s_h[sub[k]], px, py, &mx, &my);
                        sx_mvd(b, l, ox + sx, oy + sy, s_w[sub[k]], s_h[sub[k]], mx - px, my - py);
                        set_mv_l(ox + sx, oy + sy, s_w[sub[k]], s_h[sub[k]], mx, my, r[l][k], l);
                    }
            }
    }
    resid_t rs; int dummy;
    int cbp = rand_residual(&rs, 0, &dummy);
    sx_cbp(b, cbp, 0);
    if (cbp) { sx_dqp(b, rand_qp_delta()); put_residual(b, mbx, mby, &rs, 0, cbp); }
    else { memset(nnz + (size_t)cur * 24, 0, 24); w_last_dqp = 0; }
}

/* the motion of a finished reference picture, kept with its frame-store entry (for the direct prediction of B pictures) */
static void b_save_col(wdpb_t *e, int is_intra_picture)
{
    if (!e->cmv) { e->cmv = calloc((size_t)NMB * 32, 2); e->cref = malloc((size_t)NMB * 16); e->cpic = malloc((size_t)NMB * 16 * sizeof(int)); }
    for (int i = 0; i < NMB * 16; i++) {
        const int intra = is_intra_picture || mb_type[i >> 4] <= T_I16;
        const int r = intra ? -1 : refs[i];
        e->cref[i] = (int8_t)r;
        e->cpic[i] = r < 0 ? -1 : wlist[r < wlist_n ? r : wlist_n - 1];
        e->cmv[i * 2] = r < 0 ? 0 : mvs[i * 2]; e->cmv[i * 2 + 1] = r < 0 ? 0 : mvs[i * 2 + 1];
    }
}
