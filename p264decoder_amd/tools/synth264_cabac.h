/* synth264_cabac.h - the entropy-coding half of the stream writer (included by synth264.c): every macroblock-level syntax
 * element goes through an sx_* function, which writes it with CAVLC (the codes of H.264 9.1 / 9.2, as the writer always
 * did) or - with --cabac - with CABAC: the binarisations of 9.3.2, the context selection of 9.3.3.1 over the writer's OWN
 * record of the neighbouring macroblocks, and the arithmetic ENCODER of 9.3.4.  Written from the standard, separately from
 * the decoder's parser_cabac.h; tests/test_cabac_streams.py makes the parser read both forms of the same syntax.
 * (Tables: cabac_tables.h - standard data shared with the decoder, like cavlc_tables.h.)
 */
#include "cabac_tables.h"

static int opt_cabac = 0;
static int slice_kind;                       /* 0 I, 1 P, 2 B: which mb_type / sub_mb_type / skip tables apply */

/* ---- arithmetic encoder (9.3.4.2 - 9.3.4.5) --------------------------------------------------------------------------- */
static struct { uint32_t low, range; int outstanding, first; uint8_t pstate[460], mps[460]; } ce;

static void ce_init(int is_i, int idc, int qp)
{
    const int t = is_i ? 0 : 1 + idc;
    for (int i = 0; i < 460; i++) {
        int pre = ((cabac_mn[i][t][0] * qp) >> 4) + cabac_mn[i][t][1];
        if (pre < 1) pre = 1;
        if (pre > 126) pre = 126;
        if (pre <= 63) { ce.pstate[i] = (uint8_t)(63 - pre); ce.mps[i] = 0; } else { ce.pstate[i] = (uint8_t)(pre - 64); ce.mps[i] = 1; }
    }
    ce.low = 0; ce.range = 510; ce.outstanding = 0; ce.first = 1;
}
static void ce_putbit(bw_t *b, int v)
{
    if (ce.first) ce.first = 0; else bw_put(b, 1, (uint32_t)v);
    for (; ce.outstanding > 0; ce.outstanding--) bw_put(b, 1, (uint32_t)!v);
}
static void ce_renorm(bw_t *b)
{
    while (ce.range < 256) {
        if (ce.low < 256) ce_putbit(b, 0);
        else if (ce.low >= 512) { ce.low -= 512; ce_putbit(b, 1); }
        else { ce.low -= 256; ce.outstanding++; }
        ce.range <<= 1; ce.low <<= 1;
    }
}
static void ce_bin(bw_t *b, int ctx, int bin)
{
    const uint32_t lps = cabac_range_lps[ce.pstate[ctx]][(ce.range >> 6) & 3];
    ce.range -= lps;
    if (bin != ce.mps[ctx]) {
        ce.low += ce.range; ce.range = lps;
        if (ce.pstate[ctx] == 0) ce.mps[ctx] ^= 1;
        ce.pstate[ctx] = cabac_trans_lps[ce.pstate[ctx]];
    } else if (ce.pstate[ctx] < 62) ce.pstate[ctx]++;
    ce_renorm(b);
}
static void ce_bypass(bw_t *b, int bin)
{
    ce.low <<= 1;
    if (bin) ce.low += ce.range;
    if (ce.low >= 1024) { ce_putbit(b, 1); ce.low -= 1024; }
    else if (ce.low < 512) ce_putbit(b, 0);
    else { ce.low -= 512; ce.outstanding++; }
}
static void ce_terminate(bw_t *b, int bin)
{
    ce.range -= 2;
    if (bin) {
        ce.low += ce.range;
        ce.range = 2;
        ce_renorm(b);
        ce_putbit(b, (int)((ce.low >> 9) & 1));
        bw_put(b, 2, ((ce.low >> 7) & 3) | 1);              /* the last 1 is the rbsp stop bit */
    } else ce_renorm(b);
}

/* ---- the writer's record of what context selection looks at ------------------------------------------------------------ */
static uint8_t *w_skip, *w_direct16, *w_d8, *w_cbp, *w_cmode, *w_dc, *w_mvd[2];   /* per macroblock; w_mvd: [mb][16][2] */
static int w_last_dqp;
static void w_alloc(void)
{
    w_skip = calloc((size_t)NMB, 1); w_direct16 = calloc((size_t)NMB, 1); w_d8 = calloc((size_t)NMB, 1); w_cbp = calloc((size_t)NMB, 1);
    w_cmode = calloc((size_t)NMB, 1); w_dc = calloc((size_t)NMB, 1); w_mvd[0] = calloc((size_t)NMB, 32); w_mvd[1] = calloc((size_t)NMB, 32);
}
static void w_begin_mb(void)
{
    /* no list is used by the macroblock until one of its partitions says so (reference-index contexts look at partitions of
     * the current macroblock before their vectors are written) */
    memset(refs + cur * 16, -1, 16);
    if (refs1) memset(refs1 + cur * 16, -1, 16);
    w_skip[cur] = 0; w_direct16[cur] = 0; w_d8[cur] = 0; w_cbp[cur] = 0; w_cmode[cur] = 0; w_dc[cur] = 0;
    memset(w_mvd[0] + cur * 32, 0, 32); memset(w_mvd[1] + cur * 32, 0, 32);
}
static int w_A(void) { return avail(cur % W - 1, cur / W); }                 /* the macroblock to the left / above is usable */
static int w_B(void) { return avail(cur % W, cur / W - 1); }
static int w_is_intra(int mb) { return mb_type[mb] <= T_I16; }

/* ---- macroblock types ------------------------------------------------------------------------------------------------------ */
static void sx_mb_skip(bw_t *b, int skipped)
{   /* CABAC only (CAVLC counts runs in the slice loop) */
    const int base = slice_kind == 2 ? 24 : 11;
    ce_bin(b, base + (w_A() && !w_skip[cur - 1]) + (w_B() && !w_skip[cur - W]), skipped);
    if (skipped) { w_skip[cur] = 1; w_last_dqp = 0; if (slice_kind == 2) { w_direct16[cur] = 1; w_d8[cur] = 15; } }
}
static void ce_intra_type(bw_t *b, int ti, int base)
{   /* ti: I-slice numbering 0 .. 24 */
    const int in_i = slice_kind == 0;
    int s;
    if (in_i) {
        const int ctx = 3 + (w_A() && mb_type[cur - 1] != T_I4) + (w_B() && mb_type[cur - W] != T_I4);
        ce_bin(b, ctx, ti != 0);
        if (!ti) return;
        s = 5;
    } else {
        ce_bin(b, base, ti != 0);
        if (!ti) return;
        s = base;
    }
    ce_terminate(b, 0);                                      /* not I_PCM */
    const int v = ti - 1, luma = v / 12, chroma = (v % 12) / 4, pm = v & 3;
    ce_bin(b, s + 1, luma);
    ce_bin(b, s + 2, chroma != 0);
    if (chroma) ce_bin(b, s + 2 + in_i, chroma == 2);
    ce_bin(b, s + 3 + in_i, pm >> 1);
    ce_bin(b, s + 3 + 2 * in_i, pm & 1);
}
static void sx_mb_type(bw_t *b, int t)
{
    if (!opt_cabac) { bw_ue(b, (uint32_t)t); return; }
    if (slice_kind == 0) { ce_intra_type(b, t, 3); return; }
    if (slice_kind == 1) {
        if (t >= 5) { ce_bin(b, 14, 1); ce_intra_type(b, t - 5, 17); return; }
        ce_bin(b, 14, 0);
        if (t == 0 || t == 3) { ce_bin(b, 15, 0); ce_bin(b, 16, t == 3); }
        else { ce_bin(b, 15, 1); ce_bin(b, 17, t == 1); }
        return;
    }
    const int ctx = 27 + (w_A() && !w_direct16[cur - 1]) + (w_B() && !w_direct16[cur - W]);
    if (t == 0) { ce_bin(b, ctx, 0); w_direct16[cur] = 1; w_d8[cur] = 15; return; }
    ce_bin(b, ctx, 1);
    if (t <= 2) { ce_bin(b, 30, 0); ce_bin(b, 32, t - 1); return; }
    ce_bin(b, 30, 1);
    int code, extra = -1;                                    /* four bins, for types 12..21 a fifth */
    if (t <= 10) code = t - 3;
    else if (t == 11) code = 14;
    else if (t == 22) code = 15;
    else if (t >= 23) code = 13;
    else { code = (t + 4) >> 1; extra = (t + 4) & 1; }
    ce_bin(b, 31, (code >> 3) & 1); ce_bin(b, 32, (code >> 2) & 1); ce_bin(b, 32, (code >> 1) & 1); ce_bin(b, 32, code & 1);
    if (extra >= 0) ce_bin(b, 32, extra);
    if (t >= 23) ce_intra_type(b, t - 23, 32);
}
static void sx_sub_mb_type(bw_t *b, int k8, int t)
{
    if (slice_kind == 2 && t == 0) w_d8[cur] |= (uint8_t)(1 << k8);
    if (!opt_cabac) { bw_ue(b, (uint32_t)t); return; }
    if (slice_kind == 1) {
        if (t == 0) { ce_bin(b, 21, 1); return; }
        ce_bin(b, 21, 0);
        if (t == 1) { ce_bin(b, 22, 0); return; }
        ce_bin(b, 22, 1); ce_bin(b, 23, t == 2);
        return;
    }
    if (t == 0) { ce_bin(b, 36, 0); return; }
    ce_bin(b, 36, 1);
    if (t <= 2) { ce_bin(b, 37, 0); ce_bin(b, 39, t - 1); return; }
    ce_bin(b, 37, 1);
    if (t >= 11) { ce_bin(b, 38, 1); ce_bin(b, 39, 1); ce_bin(b, 39, t - 11); return; }
    const int v = t >= 7 ? t - 7 : t - 3;                    /* 3..6 -> 0..3 behind a 0; 7..10 -> 0..3 behind 1 0 (1 1 = the 4x4 types above) */
    ce_bin(b, 38, t >= 7);
    if (t >= 7) ce_bin(b, 39, 0);
    ce_bin(b, 39, v >> 1); ce_bin(b, 39, v & 1);
}

/* ---- prediction ---------------------------------------------------------------------------------------------------------- */
/* the 4x4 block at picture position (x4, y4) as a neighbour of the partition being written: its macroblock and block index, or
 * 0 when it cannot be used (outside the picture / slice, not yet coded) */
static int w_nb(int x4, int y4, int *mb, int *blk)
{
    if (x4 < 0 || y4 < 0 || (x4 >> 2) >= W || (y4 >> 2) >= H) return 0;
    const int i = (y4 >> 2) * W + (x4 >> 2);
    if (i > cur || i < slice_first) return 0;
    *mb = i; *blk = (y4 & 3) * 4 + (x4 & 3);
    return 1;
}
static void sx_ref_idx(bw_t *b, int l, int bx, int by, int bw_, int bh, int n_act, int v)
{
    int8_t *rf = l ? refs1 : refs;
    if (!opt_cabac) { put_te(b, n_act - 1, v); return; }
    for (int y = by; y < by + bh; y++) for (int x = bx; x < bx + bw_; x++) rf[cur * 16 + y * 4 + x] = (int8_t)v;   /* visible to the next partition's context */
    if (n_act <= 1) return;
    const int x0 = (cur % W) * 4 + bx, y0 = (cur / W) * 4 + by;
    int inc = 0, mb, blk;
    if (w_nb(x0 - 1, y0, &mb, &blk) && !w_is_intra(mb) && !((w_d8[mb] >> (((blk >> 3) << 1) | ((blk >> 1) & 1))) & 1) && rf[mb * 16 + blk] > 0) inc += 1;
    if (w_nb(x0, y0 - 1, &mb, &blk) && !w_is_intra(mb) && !((w_d8[mb] >> (((blk >> 3) << 1) | ((blk >> 1) & 1))) & 1) && rf[mb * 16 + blk] > 0) inc += 2;
    int ctx = 54 + inc;
    for (int i = 0; i < v; i++) { ce_bin(b, ctx, 1); ctx = 54 + (i == 0 ? 4 : 5); }
    ce_bin(b, ctx, 0);
}
static void ce_mvd(bw_t *b, int base, int sum, int v)
{
    const int a = v < 0 ? -v : v;
    ce_bin(b, base + (sum < 3 ? 0 : sum > 32 ? 2 : 1), a != 0);
    if (!a) return;
    /* unary part up to 9 with contexts base+3, +4, +5, +6, +6 ..., then 3rd order Exp-Golomb in bypass bins, then the sign */
    for (int i = 1; i < (a < 9 ? a : 9); i++) ce_bin(b, base + (i < 4 ? 2 + i : 6), 1);
    if (a < 9) ce_bin(b, base + (a < 4 ? 2 + a : 6), 0);
    else {
        int rest = a - 9, k = 3;
        while (rest >= (1 << k)) { ce_bypass(b, 1); rest -= 1 << k; k++; }
        ce_bypass(b, 0);
        while (k--) ce_bypass(b, (rest >> k) & 1);
    }
    ce_bypass(b, v < 0);
}
static void sx_mvd(bw_t *b, int l, int bx, int by, int bw_, int bh, int dx, int dy)
{
    if (!opt_cabac) { bw_se(b, dx); bw_se(b, dy); return; }
    const int x0 = (cur % W) * 4 + bx, y0 = (cur / W) * 4 + by;
    int sx = 0, sy = 0, mb, blk;
    if (w_nb(x0 - 1, y0, &mb, &blk)) { sx += w_mvd[l][(mb * 16 + blk) * 2]; sy += w_mvd[l][(mb * 16 + blk) * 2 + 1]; }
    if (w_nb(x0, y0 - 1, &mb, &blk)) { sx += w_mvd[l][(mb * 16 + blk) * 2]; sy += w_mvd[l][(mb * 16 + blk) * 2 + 1]; }
    ce_mvd(b, 40, sx, dx);
    ce_mvd(b, 47, sy, dy);
    const int ax = dx < 0 ? -dx : dx, ay = dy < 0 ? -dy : dy;
    for (int y = by; y < by + bh; y++)
        for (int x = bx; x < bx + bw_; x++) {
            w_mvd[l][(cur * 16 + y * 4 + x) * 2] = (uint8_t)(ax > 255 ? 255 : ax);
            w_mvd[l][(cur * 16 + y * 4 + x) * 2 + 1] = (uint8_t)(ay > 255 ? 255 : ay);
        }
}
static void sx_i4mode(bw_t *b, int mode, int pred)
{
    if (!opt_cabac) {
        if (mode == pred) bw_put(b, 1, 1);
        else { bw_put(b, 1, 0); bw_put(b, 3, (uint32_t)(mode < pred ? mode : mode - 1)); }
        return;
    }
    ce_bin(b, 68, mode == pred);
    if (mode == pred) return;
    const int rem = mode < pred ? mode : mode - 1;
    ce_bin(b, 69, rem & 1); ce_bin(b, 69, (rem >> 1) & 1); ce_bin(b, 69, rem >> 2);
}
static void sx_chroma_mode(bw_t *b, int v)
{
    w_cmode[cur] = (uint8_t)v;
    if (!opt_cabac) { bw_ue(b, (uint32_t)v); return; }
    const int ctx = 64 + (w_A() && w_is_intra(cur - 1) && w_cmode[cur - 1]) + (w_B() && w_is_intra(cur - W) && w_cmode[cur - W]);
    ce_bin(b, ctx, v != 0);
    if (!v) return;
    ce_bin(b, 67, v != 1);
    if (v != 1) ce_bin(b, 67, v != 2);
}
static void sx_cbp(bw_t *b, int cbp, int intra4x4)
{
    w_cbp[cur] = (uint8_t)cbp;
    if (!opt_cabac) {
        int code = -1;
        for (int k = 0; k < 48; k++) if ((intra4x4 ? cbp_intra_of_code[k] : cbp_inter_of_code[k]) == cbp) code = k;
        bw_ue(b, (uint32_t)code);
        return;
    }
    /* luma: one bin per 8x8; the condition of a neighbouring 8x8 is "its bit is 0" (a neighbour that is missing has none) */
    const int hasA = w_A(), hasB = w_B();
    for (int q = 0; q < 4; q++) {
        int ca, cb;
        if (q & 1) ca = !((cbp >> (q - 1)) & 1); else ca = hasA && !((w_cbp[cur - 1] >> (q + 1)) & 1);
        if (q & 2) cb = !((cbp >> (q - 2)) & 1); else cb = hasB && !((w_cbp[cur - W] >> (q + 2)) & 1);
        ce_bin(b, 73 + ca + 2 * cb, (cbp >> q) & 1);
    }
    const int cc = cbp >> 4, la = hasA ? w_cbp[cur - 1] >> 4 : 0, lb = hasB ? w_cbp[cur - W] >> 4 : 0;
    ce_bin(b, 77 + (la != 0) + 2 * (lb != 0), cc != 0);
    if (cc) ce_bin(b, 81 + (la == 2) + 2 * (lb == 2), cc == 2);
}
static void sx_dqp(bw_t *b, int v)
{
    if (!opt_cabac) { bw_se(b, v); return; }
    const int n = v > 0 ? 2 * v - 1 : -2 * v;               /* 0, 1, -1, 2, -2 ... -> 0, 1, 2, 3, 4 ... in unary */
    int ctx = 60 + (w_last_dqp != 0);
    for (int i = 0; i < n; i++) { ce_bin(b, ctx, 1); ctx = i == 0 ? 62 : 63; }
    ce_bin(b, ctx, 0);
    w_last_dqp = v;
}

/* ---- residual blocks ----------------------------------------------------------------------------------------------------- */
/* cat: 0 Intra16x16 DC, 1 Intra16x16 AC, 2 luma 4x4, 3 chroma DC (plane = blk), 4 chroma AC; blk: the block (0..23) */
static void ce_block(bw_t *b, int cat, int blk, const int16_t *lv, int n)
{
    static const int sig0[5] = { 105, 120, 134, 149, 152 }, last0[5] = { 166, 181, 195, 210, 213 }, abs0[5] = { 227, 237, 247, 257, 266 };
    const int intra = w_is_intra(cur), hasA = w_A(), hasB = w_B();
    int fa, fb, coded = 0;
    for (int i = 0; i < n; i++) coded |= lv[i] != 0;
    if (cat == 0 || cat == 3) {
        const int bit = cat == 0 ? 1 : 2 << blk;
        fa = hasA ? (w_dc[cur - 1] & bit) != 0 : intra;
        fb = hasB ? (w_dc[cur - W] & bit) != 0 : intra;
        if (coded) w_dc[cur] |= (uint8_t)bit;
    } else {
        const uint8_t *c = nnz + (size_t)cur * 24;
        int a, bb;
        if (blk < 16) {
            const int x = blk_x[blk], y = blk_y[blk];
            a = x ? c[blk_of_xy[y][x-1]] : hasA ? (c - 24)[blk_of_xy[y][3]] : -1;
            bb = y ? c[blk_of_xy[y-1][x]] : hasB ? (c - 24 * W)[blk_of_xy[3][x]] : -1;
        } else {
            const int base = blk < 20 ? 16 : 20, k = blk - base;
            a = (k & 1) ? c[blk - 1] : hasA ? (c - 24)[base + (k >> 1) * 2 + 1] : -1;
            bb = (k >> 1) ? c[blk - 2] : hasB ? (c - 24 * W)[base + 2 + (k & 1)] : -1;
        }
        fa = a < 0 ? intra : a != 0;
        fb = bb < 0 ? intra : bb != 0;
    }
    ce_bin(b, 85 + 4 * cat + fa + 2 * fb, coded);
    if (!coded) return;
    int last = n - 1;
    while (!lv[last]) last--;
    for (int i = 0; i < n - 1 && i <= last; i++) {
        const int k = cat == 3 && i > 2 ? 2 : i;
        ce_bin(b, sig0[cat] + k, lv[i] != 0);
        if (lv[i]) ce_bin(b, last0[cat] + k, i == last);
    }
    int ones = 0, big = 0;
    for (int i = last; i >= 0; i--) {
        if (!lv[i]) continue;
        const int a = lv[i] < 0 ? -lv[i] : lv[i];
        ce_bin(b, abs0[cat] + (big ? 0 : ones < 3 ? 1 + ones : 4), a > 1);
        if (a > 1) {
            const int cap = cat == 3 ? 3 : 4, ctx = abs0[cat] + 5 + (big < cap ? big : cap);
            for (int k = 2; k < (a < 15 ? a : 15); k++) ce_bin(b, ctx, 1);
            if (a < 15) ce_bin(b, ctx, 0);
            else {
                int rest = a - 15, k = 0;
                while (rest >= (1 << k)) { ce_bypass(b, 1); rest -= 1 << k; k++; }
                ce_bypass(b, 0);
                while (k--) ce_bypass(b, (rest >> k) & 1);
            }
            big++;
        } else ones++;
        ce_bypass(b, lv[i] < 0);
    }
}
