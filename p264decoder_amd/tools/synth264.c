/* synth264.c - synthetic H.264 Annex-B stream writer (random syntax, no rate control).
 *
 * There is no encoder in the build image and the reference ships a single 352x288 clip, so the
 * 720p / 1080p workloads of BASELINE.json (configs 2-3) are written by this tool.  It emits
 * only syntax from the subset the reference decodes correctly ("safe
 * subset"): Baseline CAVLC, one slice per picture, I and P slices, I4x4 / I16x16 / P_L0
 * 16x16,16x8,8x16 / P_8x8 with 8x8 sub-blocks / P_SKIP, one reference frame, constant QP,
 * mb_qp_delta = 0, deblocking on with zero offsets, motion vectors that keep every referenced
 * sample within 24 samples of the picture.  Like an encoder it mirrors the decoder's
 * neighbour state (MV / intra-mode / nC predictors) so that what it writes is decodable.
 * Output is a pure function of the arguments (splitmix64), so tests regenerate streams and
 * only their hashes are committed.
 *
 *   synth264 out.264 --mbw 120 --mbh 68 --frames 60 --gop 30 --seed 3 [--intra-only]
 *            [--qp 26] [--coded 12] [--maxlevel 32] [--mvmax 64] [--cqo 0] [--nodeblock]
 *            [--refs 2]      two reference frames, reference index per partition (outside the reference's safe subset, A-Q5)
 *            [--slices N]    N slices per picture (equal runs of macroblocks; the reference handles one, decoder/decoder.c:516-523)
 *            [--dump-mv f]   per picture: the intended vectors int16[mb][16][2] and reference indices int8[mb][16]
 *            [--qp-delta N]  random mb_qp_delta in [-N, N] on every macroblock that carries one (the reference adds it to the
 *                            SLICE QP instead of accumulating it, decoder/macroblock.c:568 = SURVEY A-Q2; so does this writer:
 *                            the QP of the macroblock is slice QP + delta, kept inside 0..51)
 *            [--deblock-offsets A B]  slice_alpha_c0_offset_div2 / slice_beta_offset_div2 (the reference uses them unshifted, A-Q3)
 *            [--sub8x8]      P_8x8 with all four sub_mb_types (8x8, 8x4, 4x8, 4x4; mis-decoded by the reference, A-Q4)
 *            [--reorder]     with --refs 2: some P slices swap the two entries of list 0 (ref_pic_list_reordering; ignored by
 *                            the reference, decoder/lists.c:146-149)
 *            [--dump-mv f]   additionally records, per picture, one byte: 1 when list 0 was reordered
 *            [--mmco]        with --refs 2..4: adaptive reference picture marking (memory_management_control_operation 1, 2, 3, 6:
 *                            short-term pictures dropped early, turned into long-term ones, long-term ones dropped, the current
 *                            picture stored as long-term) and long-term IDR pictures; --dump-mv then also records list 0 of every
 *                            picture as the writer means it: one byte n, then n 16-bit picture numbers (decode order)
 *            [--bframes N]   Main profile with N non-reference B pictures between consecutive reference pictures (needs --refs >= 2):
 *                            every B macroblock / sub-macroblock type, B_Skip and direct prediction - spatial, or temporal with
 *                            [--temporal]; [--d8inf] direct_8x8_inference; [--implicit] weighted_bipred_idc 2.  The reference
 *                            decodes none of this (decoder/macroblock.c:168-171): see synth264_b.h.  --dump-mv then records per
 *                            picture both lists' vectors and indices, both lists as picture numbers and the implicit weights
 *            [--cabac]       Main profile with entropy_coding_mode_flag 1: the same syntax through the CABAC binarisations and the
 *                            arithmetic encoder (synth264_cabac.h) instead of the CAVLC codes; the random choices do not depend on
 *                            the entropy coder, so `args` and `args --cabac` describe the same pictures
 *            [--pps-alt]     two identical PPS (ids 0 and 1), pictures alternate between them: every picture re-activates a
 *                            parameter set (context re-initialisation in the decoder, decoder/decoder.c:304-343)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "cavlc_tables.h"

/* ---------------------------------------------------------------- PRNG ------------------ */
static uint64_t g_rng;
static uint64_t rnd64(void)
{
    uint64_t z = (g_rng += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
static int rnd(int n) { return (int)(rnd64() % (uint64_t)n); }            /* [0,n) */
static int pct(int p) { return rnd(100) < p; }

/* ---------------------------------------------------------------- bit writer ------------ */
typedef struct { uint8_t *buf; size_t cap, len; uint32_t acc; int nacc; } bw_t;
static void bw_byte(bw_t *b, uint8_t v)
{
    if (b->len + 1 > b->cap) { b->cap = b->cap * 2 + 1024; b->buf = realloc(b->buf, b->cap); }
    b->buf[b->len++] = v;
}
static void bw_put(bw_t *b, int n, uint32_t v)
{
    for (int i = n - 1; i >= 0; i--) {
        b->acc = (b->acc << 1) | ((v >> i) & 1);
        if (++b->nacc == 8) { bw_byte(b, (uint8_t)b->acc); b->acc = 0; b->nacc = 0; }
    }
}
static void bw_ue(bw_t *b, uint32_t v)
{
    uint32_t x = v + 1; int n = 0;
    while ((x >> n) > 1) n++;
    bw_put(b, n, 0); bw_put(b, n + 1, x);
}
static void bw_se(bw_t *b, int v) { bw_ue(b, v > 0 ? (uint32_t)(2 * v - 1) : (uint32_t)(-2 * v)); }
static void bw_trailing(bw_t *b) { bw_put(b, 1, 1); while (b->nacc) bw_put(b, 1, 0); }

/* write one NAL: start code, header, payload with emulation prevention */
static void write_nal(FILE *f, int ref_idc, int type, const bw_t *b)
{
    static const uint8_t sc[4] = { 0, 0, 0, 1 };
    fwrite(sc, 1, 4, f);
    fputc((ref_idc << 5) | type, f);
    int zeros = 0;
    for (size_t i = 0; i < b->len; i++) {
        uint8_t v = b->buf[i];
        if (zeros >= 2 && v <= 3) { fputc(3, f); zeros = 0; }
        fputc(v, f);
        zeros = v == 0 ? zeros + 1 : 0;
    }
}

/* ---------------------------------------------------------------- stream state ---------- */
enum { T_I4 = 0, T_I16 = 1, T_P = 3, T_P8 = 4, T_SKIP = 5 };
static int W, H, NMB;                       /* in macroblocks */
static int opt_qp = 26, opt_coded = 12, opt_maxlevel = 32, opt_mvmax = 64, opt_cqo = 0, opt_deblock = 1, opt_refs = 1;
static int n_active = 1;                    /* num_ref_idx_l0_active of the current slice */
static int8_t *refs;                        /* [mb][16] reference index per 4x4 block (-1 intra) */
static FILE *dump_mv;                       /* --dump-mv: intended vectors and reference indices, for checking a parser */
static uint8_t *mb_type;                    /* per MB of the current picture */
static int16_t *mvs;                        /* [mb][16][2] */
static uint8_t *nnz;                        /* [mb][24] */
static int8_t  *i4m;                        /* [mb][16], 2 for non-I4x4 */
static int cur;                             /* current MB index */

static int opt_pps_alt = 0, cur_pps = 0;
static int opt_mmco = 0, opt_mmco5 = 0, had_mmco5 = 0;   /* --mmco5: also memory_management_control_operation 5; had_mmco5: the picture just written carried one */
/* the writer's own model of the decoded picture buffer (H.264 8.2.4, 8.2.5) */
typedef struct { int used, pic, frame_num, is_long, long_idx;
                 int poc; int16_t *cmv; int8_t *cref; int *cpic; } wdpb_t;   /* --bframes: picture order count; the picture's motion for direct prediction:
                                                                              * per 4x4 block its vector, reference index and the picture that index named (-1 intra) */
static wdpb_t wdpb[8];
static int wlist[8], wlist_n;                /* list 0 of the current picture: picture numbers in decode order */
static int opt_qpdelta = 0, opt_alpha = 0, opt_beta = 0, opt_sub8x8 = 0, opt_reorder = 0, slice_reordered;
static int opt_idc = 0;                     /* --deblock-idc 2: no filtering across slice boundaries */
/* --bframes N: N non-reference B pictures between consecutive reference pictures (Main profile, picture order count type 0) */
static int opt_bframes = 0, opt_temporal = 0, opt_implicit = 0, opt_d8inf = 0;
static int cur_poc, n_active1 = 1, blist[2][8], blist_n[2], cur_entry = -1;      /* B picture: its order count and its two lists as indices into wdpb */
static int opt_slices = 1, slice_first;     /* --slices: equal runs of macroblocks; slice_first = first MB of the current slice */
/* a neighbour is usable for prediction when it was coded earlier IN THE SAME SLICE (H.264 6.4.x) */
static int avail(int mbx, int mby) { return mbx >= 0 && mby >= 0 && mbx < W && mby < H && mby * W + mbx < cur && mby * W + mbx >= slice_first; }

typedef struct { int ref, x, y; } nb_t;     /* ref -2 unavailable, -1 intra (or, B pictures: the list is not used there), >= 0 reference index */
static unsigned mv_done, mv_done1;          /* 4x4 blocks of the current macroblock whose list-0 / list-1 motion is decided */
static int16_t *mvs1; static int8_t *refs1; /* list 1 of the current picture (--bframes) */
static nb_t nb_motion_l(int x4, int y4, int l)
{
    nb_t r = { -2, 0, 0 };
    if (x4 < 0 || y4 < 0 || (x4 >> 2) >= W || (y4 >> 2) >= H) return r;
    int i = (y4 >> 2) * W + (x4 >> 2), sub = (y4 & 3) * 4 + (x4 & 3);
    if (i == cur) { if (!(((l ? mv_done1 : mv_done) >> sub) & 1)) return r; }
    else if (i > cur || i < slice_first) return r;
    if (i != cur && mb_type[i] <= T_I16) { r.ref = -1; return r; }
    const int8_t *rf = l ? refs1 : refs; const int16_t *mv = l ? mvs1 : mvs;
    r.ref = rf[i * 16 + sub]; r.x = mv[(i * 16 + sub) * 2]; r.y = mv[(i * 16 + sub) * 2 + 1];
    return r;
}
static nb_t nb_motion(int x4, int y4) { return nb_motion_l(x4, y4, 0); }
static int med3(int a, int b, int c) { int lo = a < b ? a : b, hi = a < b ? b : a; return c < lo ? lo : c > hi ? hi : c; }
/* H.264 8.4.1.3: dir 1/2 = 16x8 upper/lower, 3/4 = 8x16 left/right; ref = reference index of the partition */
static void predict_mv_l(int mbx, int mby, int bx, int by, int bw, int dir, int ref, int *px, int *py, int l)
{
    int x0 = mbx * 4 + bx, y0 = mby * 4 + by;
    nb_t a = nb_motion_l(x0 - 1, y0, l), b = nb_motion_l(x0, y0 - 1, l), c = nb_motion_l(x0 + bw, y0 - 1, l);
    if (c.ref == -2) c = nb_motion_l(x0 - 1, y0 - 1, l);
    if (dir == 1 && b.ref == ref) { *px = b.x; *py = b.y; return; }
    if (dir == 2 && a.ref == ref) { *px = a.x; *py = a.y; return; }
    if (dir == 3 && a.ref == ref) { *px = a.x; *py = a.y; return; }
    if (dir == 4 && c.ref == ref) { *px = c.x; *py = c.y; return; }
    int hits = (a.ref == ref) + (b.ref == ref) + (c.ref == ref);
    if (hits == 1) { nb_t *s = a.ref == ref ? &a : b.ref == ref ? &b : &c; *px = s->x; *py = s->y; return; }
    if (hits == 0 && b.ref == -2 && c.ref == -2 && a.ref != -2) { *px = a.x; *py = a.y; return; }
    /* neighbours that are intra or unavailable count as zero vectors in the median */
    if (a.ref < 0) a.x = a.y = 0;
    if (b.ref < 0) b.x = b.y = 0;
    if (c.ref < 0) c.x = c.y = 0;
    *px = med3(a.x, b.x, c.x); *py = med3(a.y, b.y, c.y);
}
static void predict_mv(int mbx, int mby, int bx, int by, int bw, int dir, int ref, int *px, int *py) { predict_mv_l(mbx, mby, bx, by, bw, dir, ref, px, py, 0); }
static void set_mv_l(int bx, int by, int bw, int bh, int mx, int my, int ref, int l)
{
    int8_t *rf = l ? refs1 : refs; int16_t *mv = l ? mvs1 : mvs;
    for (int y = by; y < by + bh; y++)
        for (int x = bx; x < bx + bw; x++) {
            rf[cur * 16 + y * 4 + x] = (int8_t)ref;
            mv[(cur * 16 + y * 4 + x) * 2] = (int16_t)mx; mv[(cur * 16 + y * 4 + x) * 2 + 1] = (int16_t)my;
            if (l) mv_done1 |= 1u << (y * 4 + x); else mv_done |= 1u << (y * 4 + x);
        }
}
static void set_mv(int bx, int by, int bw, int bh, int mx, int my, int ref) { set_mv_l(bx, by, bw, bh, mx, my, ref, 0); }
/* keep every referenced sample within ~19 samples of the picture (A-Q9 allows 24) */
static int mv_ok(int mbx, int mby, int bx, int by, int bw, int bh, int mx, int my)
{
    int x = mbx * 16 + bx * 4 + (mx >> 2), y = mby * 16 + by * 4 + (my >> 2);
    return x >= -16 && y >= -16 && x + bw * 4 <= W * 16 + 16 && y + bh * 4 <= H * 16 + 16;
}
static void random_mv(int mbx, int mby, int bx, int by, int bw, int bh, int *mx, int *my)
{
    for (int tries = 0; tries < 64; tries++) {
        int x = rnd(2 * opt_mvmax + 1) - opt_mvmax, y = rnd(2 * opt_mvmax + 1) - opt_mvmax;
        if (mv_ok(mbx, mby, bx, by, bw, bh, x, y)) { *mx = x; *my = y; return; }
    }
    *mx = 0; *my = 0;
}

static int predict_nc(int mbx, int mby, int blk)
{
    const uint8_t *c = nnz + (size_t)cur * 24;
    int na = -1, nb = -1, L = avail(mbx - 1, mby), T = avail(mbx, mby - 1);
    if (blk < 16) {
        int x = blk_x[blk], y = blk_y[blk];
        if (x > 0) na = c[blk_of_xy[y][x-1]]; else if (L) na = (c - 24)[blk_of_xy[y][3]];
        if (y > 0) nb = c[blk_of_xy[y-1][x]]; else if (T) nb = (c - 24 * W)[blk_of_xy[3][x]];
    } else {
        int base = blk < 20 ? 16 : 20, k = blk - base, x = k & 1, y = k >> 1;
        if (x > 0) na = c[blk - 1]; else if (L) na = (c - 24)[base + y * 2 + 1];
        if (y > 0) nb = c[blk - 2]; else if (T) nb = (c - 24 * W)[base + 2 + x];
    }
    if (na >= 0 && nb >= 0) return (na + nb + 1) >> 1;
    return na >= 0 ? na : nb >= 0 ? nb : 0;
}

/* te(v): one inverted bit when the range is 0..1, ue(v) beyond */
static void put_te(bw_t *b, int max, int v) { if (max == 1) bw_put(b, 1, (uint32_t)!v); else if (max > 1) bw_ue(b, (uint32_t)v); }
#include "synth264_cabac.h"

/* ---------------------------------------------------------------- CAVLC residual writer - */
static int rand_level(void)
{
    int m = 1;
    while (m < opt_maxlevel && pct(45)) m += 1 + (pct(20) ? rnd(6) : 0);
    if (m > opt_maxlevel) m = opt_maxlevel;
    return rnd(2) ? m : -m;
}
/* fill `n` scan positions with tc random non-zero levels */
static int rand_block(int16_t *lv, int n)
{
    memset(lv, 0, sizeof(int16_t) * 16);
    int tc = 1;
    while (tc < n && pct(55)) tc++;
    if (pct(4)) tc = n;                                   /* exercise total_coeff == max */
    for (int k = 0; k < tc; k++) {
        int p;
        do p = pct(60) ? rnd((n + 1) / 2) : rnd(n); while (lv[p]);
        lv[p] = (int16_t)(pct(35) ? (rnd(2) ? 1 : -1) : rand_level());
    }
    return tc;
}

static void put_block(bw_t *b, const int16_t *lv, int n, int nC)
{
    int idx[16], tc = 0;
    for (int i = n - 1; i >= 0; i--) if (lv[i]) idx[tc++] = i;          /* high frequency first */
    int t1 = 0;
    while (t1 < tc && t1 < 3 && (lv[idx[t1]] == 1 || lv[idx[t1]] == -1)) t1++;
    if (nC < 0) bw_put(b, ctdc_len[t1][tc], ctdc_code[t1][tc]);
    else if (nC >= 8) bw_put(b, 6, tc == 0 ? 3u : (uint32_t)(((tc - 1) << 2) | t1));
    else { int c = nC < 2 ? 0 : nC < 4 ? 1 : 2; bw_put(b, ct_len[c][t1][tc], ct_code[c][t1][tc]); }
    if (!tc) return;
    for (int i = 0; i < t1; i++) bw_put(b, 1, lv[idx[i]] < 0);
    int sl = (tc > 10 && t1 < 3) ? 1 : 0;
    for (int i = t1; i < tc; i++) {
        int v = lv[idx[i]], code = v > 0 ? 2 * v - 2 : -2 * v - 1;
        if (i == t1 && t1 < 3) code -= 2;
        if (sl == 0) {
            if (code < 14) { bw_put(b, code, 0); bw_put(b, 1, 1); }
            else if (code < 30) { bw_put(b, 14, 0); bw_put(b, 1, 1); bw_put(b, 4, (uint32_t)(code - 14)); }
            else { bw_put(b, 15, 0); bw_put(b, 1, 1); bw_put(b, 12, (uint32_t)(code - 30)); }
        } else {
            if (code < (15 << sl)) { bw_put(b, code >> sl, 0); bw_put(b, 1, 1); bw_put(b, sl, (uint32_t)(code & ((1 << sl) - 1))); }
            else { bw_put(b, 15, 0); bw_put(b, 1, 1); bw_put(b, 12, (uint32_t)(code - (15 << sl))); }
        }
        if (sl == 0) sl = 1;
        int a = v < 0 ? -v : v;
        if (a > (3 << (sl - 1)) && sl < 6) sl++;
    }
    int total_zeros = idx[0] + 1 - tc;
    if (tc < n) {
        if (n == 4) bw_put(b, tzdc_len[tc-1][total_zeros], tzdc_code[tc-1][total_zeros]);
        else bw_put(b, tz_len[tc-1][total_zeros], tz_code[tc-1][total_zeros]);
    }
    int zl = total_zeros;
    for (int i = 0; i < tc - 1 && zl > 0; i++) {
        int run = idx[i] - idx[i+1] - 1, t = (zl > 7 ? 7 : zl) - 1;
        bw_put(b, rb_len[t][run], rb_code[t][run]);
        zl -= run;
    }
}

/* ---------------------------------------------------------------- macroblock writers ---- */
static int rand_qp_delta(void)
{
    if (!opt_qpdelta) return 0;
    int lo = -opt_qpdelta, hi = opt_qpdelta;
    if (opt_qp + lo < 0) lo = -opt_qp;
    if (opt_qp + hi > 51) hi = 51 - opt_qp;
    return pct(30) ? 0 : lo + rnd(hi - lo + 1);
}
typedef struct { int16_t dc_luma[16], dc_c[2][16], blk[24][16]; int has[24]; } resid_t;

/* draw residual content for the MB; returns cbp (luma bits 0-3, chroma bits 4-5) */
static int rand_residual(resid_t *r, int is_i16, int *i16_ac)
{
    memset(r, 0, sizeof *r);
    int cbp = 0, n = is_i16 ? 15 : 16;
    int dense = pct(opt_coded * 2);                    /* some MBs carry most of the coefficients */
    if (is_i16) {
        if (pct(60)) rand_block(r->dc_luma, 16);
        *i16_ac = pct(35);
        if (*i16_ac) { cbp = 15; for (int i = 0; i < 16; i++) if (pct(dense ? 60 : 25)) { rand_block(r->blk[i], n); r->has[i] = 1; } }
    } else {
        for (int q = 0; q < 4; q++) {
            if (!pct(dense ? 70 : opt_coded)) continue;
            int any = 0;
            for (int j = 0; j < 4; j++) if (pct(dense ? 60 : 40)) { rand_block(r->blk[q*4+j], n); r->has[q*4+j] = 1; any = 1; }
            if (pct(90) || any) cbp |= 1 << q;        /* occasionally a coded 8x8 with four empty blocks */
            if (!(cbp & (1 << q))) for (int j = 0; j < 4; j++) { memset(r->blk[q*4+j], 0, 32); r->has[q*4+j] = 0; }
        }
    }
    int cc = pct(dense ? 60 : opt_coded) ? 1 + pct(50) : 0;
    if (cc) {
        for (int p = 0; p < 2; p++) if (pct(70)) rand_block(r->dc_c[p], 4);
        if (cc == 2) for (int i = 16; i < 24; i++) if (pct(50)) { rand_block(r->blk[i], 15); r->has[i] = 1; }
    }
    return cbp | (cc << 4);
}

static void put_residual(bw_t *b, int mbx, int mby, const resid_t *r, int is_i16, int cbp)
{
    uint8_t *nz = nnz + (size_t)cur * 24;
    if (is_i16) { if (opt_cabac) ce_block(b, 0, 0, r->dc_luma, 16); else put_block(b, r->dc_luma, 16, predict_nc(mbx, mby, 0)); }
    for (int i = 0; i < 16; i++) {
        nz[i] = 0;
        if (!(cbp & (1 << (i >> 2)))) continue;
        if (opt_cabac) ce_block(b, is_i16 ? 1 : 2, i, r->blk[i], is_i16 ? 15 : 16);
        else put_block(b, r->blk[i], is_i16 ? 15 : 16, predict_nc(mbx, mby, i));
        int tc = 0; for (int k = 0; k < 16; k++) tc += r->blk[i][k] != 0;
        nz[i] = (uint8_t)tc;
    }
    if (cbp >> 4) {
        if (opt_cabac) { ce_block(b, 3, 0, r->dc_c[0], 4); ce_block(b, 3, 1, r->dc_c[1], 4); }
        else { put_block(b, r->dc_c[0], 4, -1); put_block(b, r->dc_c[1], 4, -1); }
    }
    for (int i = 16; i < 24; i++) {
        nz[i] = 0;
        if (!((cbp >> 4) & 2)) continue;
        if (opt_cabac) ce_block(b, 4, i, r->blk[i], 15); else
        put_block(b, r->blk[i], 15, predict_nc(mbx, mby, i));
        int tc = 0; for (int k = 0; k < 16; k++) tc += r->blk[i][k] != 0;
        nz[i] = (uint8_t)tc;
    }
}

static int pred_i4mode(int mbx, int mby, int blk)
{
    int x = blk_x[blk], y = blk_y[blk], ma, mb;
    if (x > 0) ma = i4m[cur * 16 + blk_of_xy[y][x-1]];
    else if (avail(mbx - 1, mby)) ma = mb_type[cur - 1] == T_I4 ? i4m[(cur - 1) * 16 + blk_of_xy[y][3]] : 2;
    else ma = -1;
    if (y > 0) mb = i4m[cur * 16 + blk_of_xy[y-1][x]];
    else if (avail(mbx, mby - 1)) mb = mb_type[cur - W] == T_I4 ? i4m[(cur - W) * 16 + blk_of_xy[3][x]] : 2;
    else mb = -1;
    int m = ma < mb ? ma : mb;
    return m < 0 ? 2 : m;
}

/* intra MB (I slice: offset 0; P slice: mb_type + 5) */
static void put_intra(bw_t *b, int mbx, int mby, int type_offset)
{
    int L = avail(mbx - 1, mby), T = avail(mbx, mby - 1), TL = avail(mbx - 1, mby - 1);
    int is16 = pct(50);
    resid_t r; int i16_ac = 0;
    int cbp = rand_residual(&r, is16, &i16_ac);
    memset(mvs + cur * 32, 0, 64);
    memset(refs + cur * 16, -1, 16);
    if (is16) {
        int legal[4], nl = 0;
        if (T) legal[nl++] = 0;
        if (L) legal[nl++] = 1;
        legal[nl++] = 2;
        if (L && T && TL) legal[nl++] = 3;
        int mode = legal[rnd(nl)];
        mb_type[cur] = T_I16;
        memset(i4m + cur * 16, 2, 16);
        sx_mb_type(b, type_offset + 1 + mode + 4 * (cbp >> 4) + (i16_ac ? 12 : 0));
        w_cbp[cur] = (uint8_t)cbp;
    } else {
        mb_type[cur] = T_I4;
        sx_mb_type(b, type_offset);
        for (int i = 0; i < 16; i++) {
            int bx = blk_x[i], by = blk_y[i];
            int l = bx > 0 || L, t = by > 0 || T;
            int tl = (bx > 0 && by > 0) ? 1 : bx > 0 ? T : by > 0 ? L : TL;
            int legal[9], nl = 0;
            if (t) legal[nl++] = 0;
            if (l) legal[nl++] = 1;
            legal[nl++] = 2;
            if (t) { legal[nl++] = 3; legal[nl++] = 7; }           /* missing top-right is replicated, as in the standard */
            if (l && t && tl) { legal[nl++] = 4; legal[nl++] = 5; legal[nl++] = 6; }
            if (l) legal[nl++] = 8;
            int mode = legal[rnd(nl)], pred = pred_i4mode(mbx, mby, i);
            i4m[cur * 16 + i] = (int8_t)mode;
            sx_i4mode(b, mode, pred);
        }
    }
    {   /* intra_chroma_pred_mode */
        int legal[4], nl = 0;
        legal[nl++] = 0;
        if (L) legal[nl++] = 1;
        if (T) legal[nl++] = 2;
        if (L && T && TL) legal[nl++] = 3;
        sx_chroma_mode(b, legal[rnd(nl)]);
    }
    if (!is16) sx_cbp(b, cbp, 1);
    if (cbp || is16) { sx_dqp(b, rand_qp_delta()); put_residual(b, mbx, mby, &r, is16, cbp); }
    else { memset(nnz + (size_t)cur * 24, 0, 24); w_last_dqp = 0; }
}

/* try to write a non-skipped inter MB; returns 0 if no legal vectors were found */
static void put_inter(bw_t *b, int mbx, int mby)
{
    int pick = rnd(76);           /* 16x16 : 16x8 : 8x16 : 8x8 = 50 : 9 : 9 : 8 */
    int t = pick < 50 ? 0 : pick < 59 ? 1 : pick < 68 ? 2 : 3;
    static const int8_t geo[3][2][4] = { { {0,0,4,4}, {0,0,0,0} }, { {0,0,4,2}, {0,2,4,2} }, { {0,0,2,4}, {2,0,2,4} } };
    mb_type[cur] = t == 3 ? T_P8 : T_P;
    memset(i4m + cur * 16, 2, 16);
    sx_mb_type(b, t);
    /* te(v) with two active references: one bit, inverted */
    int pref[4] = { 0, 0, 0, 0 };
    if (t < 3) {
        int np = t == 0 ? 1 : 2;
        for (int k = 0; k < np; k++) {
            if (n_active > 1) pref[k] = pct(55) ? 0 : rnd(n_active);
            sx_ref_idx(b, 0, geo[t][k][0], geo[t][k][1], geo[t][k][2], geo[t][k][3], n_active, pref[k]);
        }
        for (int k = 0; k < np; k++) {
            int mx, my, px, py, dir = t == 0 ? 0 : t == 1 ? 1 + k : 3 + k;
            predict_mv(mbx, mby, geo[t][k][0], geo[t][k][1], geo[t][k][2], dir, pref[k], &px, &py);
            if (pct(25) && mv_ok(mbx, mby, geo[t][k][0], geo[t][k][1], geo[t][k][2], geo[t][k][3], px, py)) { mx = px; my = py; }
            else random_mv(mbx, mby, geo[t][k][0], geo[t][k][1], geo[t][k][2], geo[t][k][3], &mx, &my);
            sx_mvd(b, 0, geo[t][k][0], geo[t][k][1], geo[t][k][2], geo[t][k][3], mx - px, my - py);
            set_mv(geo[t][k][0], geo[t][k][1], geo[t][k][2], geo[t][k][3], mx, my, pref[k]);
        }
    } else {
        int sub[4];
        for (int k = 0; k < 4; k++) { sub[k] = opt_sub8x8 ? rnd(4) : 0; sx_sub_mb_type(b, k, sub[k]); }   /* sub_mb_type: 8x8 only inside the reference's safe subset (A-Q4) */
        for (int k = 0; k < 4; k++) {
            if (n_active > 1) pref[k] = pct(55) ? 0 : rnd(n_active);
            sx_ref_idx(b, 0, (k & 1) * 2, (k >> 1) * 2, 2, 2, n_active, pref[k]);
        }
        for (int k = 0; k < 4; k++) {
            int ox = (k & 1) * 2, oy = (k >> 1) * 2;
            int sw = (sub[k] == 0 || sub[k] == 1) ? 2 : 1, sh = (sub[k] == 0 || sub[k] == 2) ? 2 : 1;   /* 8x8, 8x4, 4x8, 4x4 */
            for (int sy = 0; sy < 2; sy += sh)
                for (int sx = 0; sx < 2; sx += sw) {
                    int mx, my, px, py;
                    predict_mv(mbx, mby, ox + sx, oy + sy, sw, 0, pref[k], &px, &py);
                    if (sub[k] && pct(30) && mv_ok(mbx, mby, ox + sx, oy + sy, sw, sh, px, py)) { mx = px; my = py; }
                    else random_mv(mbx, mby, ox + sx, oy + sy, sw, sh, &mx, &my);
                    sx_mvd(b, 0, ox + sx, oy + sy, sw, sh, mx - px, my - py);
                    set_mv(ox + sx, oy + sy, sw, sh, mx, my, pref[k]);
                }
        }
    }
    resid_t r; int dummy;
    int cbp = rand_residual(&r, 0, &dummy);
    sx_cbp(b, cbp, 0);
    if (cbp) { sx_dqp(b, rand_qp_delta()); put_residual(b, mbx, mby, &r, 0, cbp); }
    else { memset(nnz + (size_t)cur * 24, 0, 24); w_last_dqp = 0; }
}

/* P_SKIP motion (8.4.1.1); returns 0 when the inferred vector would leave the safe zone */
static int try_skip(int mbx, int mby)
{
    int mx = 0, my = 0;
    nb_t a = nb_motion(mbx * 4 - 1, mby * 4), b = nb_motion(mbx * 4, mby * 4 - 1);
    if (!(a.ref == -2 || b.ref == -2 || (a.ref == 0 && !a.x && !a.y) || (b.ref == 0 && !b.x && !b.y)))
        predict_mv(mbx, mby, 0, 0, 4, 0, 0, &mx, &my);
    if (!mv_ok(mbx, mby, 0, 0, 4, 4, mx, my)) return 0;
    mb_type[cur] = T_SKIP;
    memset(i4m + cur * 16, 2, 16);
    memset(nnz + (size_t)cur * 24, 0, 24);
    set_mv(0, 0, 4, 4, mx, my, 0);
    return 1;
}

#include "synth264_b.h"

/* ---------------------------------------------------------------- pictures -------------- */
/* list 0 of the picture about to be written, from the writer's frame-store model: short-term pictures by descending PicNum,
 * then long-term ones by ascending index (8.2.4.2.1) */
static int model_list0(int frame_num, int max_fn, int *list)
{
    int n = 0, n_short;
    int key[8];
    for (int i = 0; i < 8; i++) {                       /* short-term, PicNum descending */
        if (!wdpb[i].used || wdpb[i].is_long) continue;
        const int k = wdpb[i].frame_num > frame_num ? wdpb[i].frame_num - max_fn : wdpb[i].frame_num;
        int j = n++;
        while (j > 0 && key[j-1] < k) { key[j] = key[j-1]; list[j] = list[j-1]; j--; }
        key[j] = k; list[j] = wdpb[i].pic;
    }
    n_short = n;
    for (int i = 0; i < 8; i++) {                       /* long-term, index ascending */
        if (!wdpb[i].used || !wdpb[i].is_long) continue;
        int j = n++;
        while (j > n_short && key[j-1] > wdpb[i].long_idx) { key[j] = key[j-1]; list[j] = list[j-1]; j--; }
        key[j] = wdpb[i].long_idx; list[j] = wdpb[i].pic;
    }
    return n;
}

static void put_slice(FILE *f, int idr, int is_p, int is_b, int frame_num, int idr_id, int log2_fn, int refs_available, int pic_no)
{
    const int max_fn = 1 << log2_fn;
    int full[8], n_full = 0;
    if (opt_mmco || opt_bframes) { n_full = is_p ? model_list0(frame_num, max_fn, full) : 0; refs_available = n_full; }
    n_active = is_p && opt_refs > 1 && refs_available > 1 ? (refs_available < opt_refs ? refs_available : opt_refs) : 1;
    if (!opt_mmco && n_active > 2) n_active = 2;
    slice_reordered = opt_reorder && n_active > 1 && pct(50);
    wlist_n = 0;
    if ((opt_mmco || opt_bframes) && is_p) { wlist_n = n_active; for (int i = 0; i < n_active; i++) wlist[i] = full[i < n_full ? i : n_full - 1]; }
    if (is_b) {                                     /* the two lists of a B picture, all of them active (at most four entries each) */
        b_build_lists();
        n_active = blist_n[0] > 4 ? 4 : blist_n[0]; n_active1 = blist_n[1] > 4 ? 4 : blist_n[1];
    }
    /* ---- reference picture marking of this picture (decided once, written into every slice header) ---- */
    int n_cmd = 0, cmd[8][3], idr_long = 0, cur_long = 0, cur_long_idx = 0;
    if (opt_mmco) {
        if (idr) { idr_long = pct(30); for (int i = 0; i < 8; i++) wdpb[i].used = 0; cur_long = idr_long; cur_long_idx = 0; }
        else {
            int n_short = 0, n_long = 0, cnt = 0;
            for (int i = 0; i < 8; i++) if (wdpb[i].used) { cnt++; if (wdpb[i].is_long) n_long++; else n_short++; }
            const int want = pct(60) ? 1 + rnd(opt_mmco5 ? 5 : 4) : 0;   /* 1: drop a short-term, 2: short -> long, 3: drop a long-term, 4: current -> long, 5: drop everything */
            int pick = -1;
            if (want == 1 || want == 2) { int k = n_short ? rnd(n_short) : -1; for (int i = 0; i < 8 && k >= 0; i++) if (wdpb[i].used && !wdpb[i].is_long && k-- == 0) pick = i; }
            if (want == 3) { int k = n_long ? rnd(n_long) : -1; for (int i = 0; i < 8 && k >= 0; i++) if (wdpb[i].used && wdpb[i].is_long && k-- == 0) pick = i; }
            if (want == 1 && pick >= 0 && cnt > 1) {
                const int num = wdpb[pick].frame_num > frame_num ? wdpb[pick].frame_num - max_fn : wdpb[pick].frame_num;
                cmd[n_cmd][0] = 1; cmd[n_cmd][1] = frame_num - num - 1; n_cmd++; wdpb[pick].used = 0;
            } else if (want == 2 && pick >= 0) {
                const int num = wdpb[pick].frame_num > frame_num ? wdpb[pick].frame_num - max_fn : wdpb[pick].frame_num, li = rnd(2);
                cmd[n_cmd][0] = 3; cmd[n_cmd][1] = frame_num - num - 1; cmd[n_cmd][2] = li; n_cmd++;
                for (int i = 0; i < 8; i++) if (i != pick && wdpb[i].used && wdpb[i].is_long && wdpb[i].long_idx == li) wdpb[i].used = 0;
                wdpb[pick].is_long = 1; wdpb[pick].long_idx = li;
            } else if (want == 3 && pick >= 0 && cnt > 1) {
                cmd[n_cmd][0] = 2; cmd[n_cmd][1] = wdpb[pick].long_idx; n_cmd++; wdpb[pick].used = 0;
            } else if (want == 5) {
                /* operation 5: every other reference goes, and the picture counts as frame_num 0 from here on (7.4.3, 8.2.1):
                 * the next picture is written with frame_num 1 (main loop) */
                cmd[n_cmd][0] = 5; n_cmd++;
                for (int i = 0; i < 8; i++) wdpb[i].used = 0;
                had_mmco5 = 1;
            } else if (want == 4) {
                const int li = rnd(2);
                cmd[n_cmd][0] = 6; cmd[n_cmd][2] = li; n_cmd++;
                for (int i = 0; i < 8; i++) if (wdpb[i].used && wdpb[i].is_long && wdpb[i].long_idx == li) wdpb[i].used = 0;
                cur_long = 1; cur_long_idx = li;
            }
            /* no sliding window under adaptive marking: make room for this picture ourselves */
            cnt = 0; for (int i = 0; i < 8; i++) cnt += wdpb[i].used;
            if (n_cmd && cnt >= opt_refs) {
                int old = -1, old_num = 0;
                for (int i = 0; i < 8; i++) if (wdpb[i].used && !wdpb[i].is_long) { const int num = wdpb[i].frame_num > frame_num ? wdpb[i].frame_num - max_fn : wdpb[i].frame_num; if (old < 0 || num < old_num) { old = i; old_num = num; } }
                if (old >= 0) { cmd[n_cmd][0] = 1; cmd[n_cmd][1] = frame_num - old_num - 1; n_cmd++; wdpb[old].used = 0; }
                else { for (int i = 0; i < 8; i++) if (wdpb[i].used) { cmd[n_cmd][0] = 2; cmd[n_cmd][1] = wdpb[i].long_idx; n_cmd++; wdpb[i].used = 0; break; } }
            }
            if (!n_cmd) {                                       /* sliding window (8.2.5.3) */
                int old = -1, old_num = 0;
                for (int i = 0; i < 8; i++) if (wdpb[i].used && !wdpb[i].is_long) { const int num = wdpb[i].frame_num > frame_num ? wdpb[i].frame_num - max_fn : wdpb[i].frame_num; if (old < 0 || num < old_num) { old = i; old_num = num; } }
                if (cnt >= opt_refs && old >= 0) wdpb[old].used = 0;
            }
        }
        for (int i = 0; i < 8; i++) if (!wdpb[i].used) { wdpb[i].used = 1; wdpb[i].pic = pic_no; wdpb[i].frame_num = had_mmco5 ? 0 : frame_num; wdpb[i].is_long = cur_long; wdpb[i].long_idx = cur_long_idx; break; }
    }
    if (opt_bframes && !opt_mmco && !is_b) {        /* reference picture under the sliding window; its motion is saved when it is finished */
        if (idr) for (int i = 0; i < 8; i++) wdpb[i].used = 0;
        int cnt = 0, old = -1, old_num = 0;
        for (int i = 0; i < 8; i++) if (wdpb[i].used) { cnt++; const int num = wdpb[i].frame_num > frame_num ? wdpb[i].frame_num - max_fn : wdpb[i].frame_num; if (old < 0 || num < old_num) { old = i; old_num = num; } }
        if (cnt >= opt_refs && old >= 0) wdpb[old].used = 0;
        for (int i = 0; i < 8; i++) if (!wdpb[i].used) { wdpb[i].used = 1; wdpb[i].pic = pic_no; wdpb[i].frame_num = frame_num; wdpb[i].is_long = 0; wdpb[i].poc = cur_poc; cur_entry = i; break; }
    }
    for (int sl = 0; sl < opt_slices; sl++) {
        const int first = (int)((long)NMB * sl / opt_slices), end = (int)((long)NMB * (sl + 1) / opt_slices);
        if (first == end) continue;
        slice_first = first;
        bw_t b = { 0 };
        bw_ue(&b, (uint32_t)first);                 /* first_mb_in_slice */
        bw_ue(&b, is_b ? 6 : is_p ? 5 : 7);         /* slice_type: all slices of the picture alike */
        bw_ue(&b, (uint32_t)cur_pps);               /* pps id */
        bw_put(&b, log2_fn, (uint32_t)frame_num);
        if (idr) bw_ue(&b, (uint32_t)idr_id);
        if (opt_bframes) bw_put(&b, 8, (uint32_t)(cur_poc & 255));    /* pic_order_cnt_lsb (type 0, 8 bits) */
        if (is_b) {
            bw_put(&b, 1, (uint32_t)!opt_temporal);                   /* direct_spatial_mv_pred_flag */
            bw_put(&b, 1, 1); bw_ue(&b, (uint32_t)(n_active - 1)); bw_ue(&b, (uint32_t)(n_active1 - 1));   /* num_ref_idx_active_override */
            bw_put(&b, 1, 0); bw_put(&b, 1, 0);                       /* no reordering of list 0, list 1 */
        }
        if (is_p) {
            if (opt_refs > 1) { bw_put(&b, 1, 1); bw_ue(&b, (uint32_t)(n_active - 1)); }   /* num_ref_idx_active_override */
            else bw_put(&b, 1, 0);
            if (opt_reorder && n_active > 1 && slice_reordered) {
                /* list 0 starts as (frame_num - 1, frame_num - 2); "subtract 2 from the prediction" puts frame_num - 2 first */
                bw_put(&b, 1, 1); bw_ue(&b, 0); bw_ue(&b, 1); bw_ue(&b, 3);
            } else bw_put(&b, 1, 0);                        /* no reordering */
        }
        if (is_b) { }                               /* a non-reference picture: no dec_ref_pic_marking */
        else if (idr) { bw_put(&b, 1, 0); bw_put(&b, 1, (uint32_t)idr_long); }    /* no_output_of_prior_pics, long_term_reference */
        else if (n_cmd) {
            bw_put(&b, 1, 1);                               /* adaptive_ref_pic_marking_mode */
            for (int k = 0; k < n_cmd; k++) {
                bw_ue(&b, (uint32_t)cmd[k][0]);
                if (cmd[k][0] == 1 || cmd[k][0] == 3 || cmd[k][0] == 2) bw_ue(&b, (uint32_t)cmd[k][1]);
                if (cmd[k][0] == 3 || cmd[k][0] == 6) bw_ue(&b, (uint32_t)cmd[k][2]);
            }
            bw_ue(&b, 0);
        } else bw_put(&b, 1, 0);                            /* sliding-window marking */
        if (opt_cabac && (is_p || is_b)) bw_ue(&b, (uint32_t)(pic_no % 3));   /* cabac_init_idc */
        bw_se(&b, 0);                               /* slice_qp_delta */
        bw_ue(&b, (uint32_t)(opt_deblock ? opt_idc : 1)); /* disable_deblocking_filter_idc: 0 also across slice boundaries, 2 not */
        if (opt_deblock) { bw_se(&b, opt_alpha); bw_se(&b, opt_beta); }
        int skip_run = 0;
        slice_kind = is_b ? 2 : is_p ? 1 : 0;
        w_last_dqp = 0;
        if (opt_cabac) {                            /* cabac_alignment_one_bit, then the arithmetic code words */
            while (b.nacc) bw_put(&b, 1, 1);
            ce_init(!is_p && !is_b, pic_no % 3, opt_qp);
        }
        for (cur = first; cur < end; cur++) {
            int mbx = cur % W, mby = cur / W;
            mv_done = 0; mv_done1 = 0;
            w_begin_mb();
            if (opt_bframes) { memset(refs1 + cur * 16, -1, 16); memset(mvs1 + cur * 32, 0, 64); }
            int skipped = 0, k = 0;
            if (is_b) { k = rnd(100); if (k < 18) { b_skip(mbx, mby); skipped = 1; } }
            else if (is_p) { k = rnd(100); if (k < 20 && try_skip(mbx, mby)) skipped = 1; }
            if (is_b || is_p) {
                if (opt_cabac) sx_mb_skip(&b, skipped);
                else if (skipped) skip_run++;
                else { bw_ue(&b, (uint32_t)skip_run); skip_run = 0; }
            }
            if (!skipped) {
                if (is_b) { if (k >= 95) put_intra(&b, mbx, mby, 23); else put_b_mb(&b, mbx, mby); }
                else if (is_p) { if (k >= 96) put_intra(&b, mbx, mby, 5); else put_inter(&b, mbx, mby); }
                else put_intra(&b, mbx, mby, 0);
            }
            if (opt_cabac) ce_terminate(&b, cur == end - 1);      /* end_of_slice_flag */
        }
        if (opt_cabac) { while (b.nacc) bw_put(&b, 1, 0); }       /* (the stop bit came with the last terminate bin) */
        else {
            if (skip_run) bw_ue(&b, (uint32_t)skip_run);
            bw_trailing(&b);
        }
        write_nal(f, is_b ? 0 : 3, idr ? 5 : 1, &b);
        free(b.buf);
    }
    slice_first = 0;
    if (opt_bframes && !is_b && cur_entry >= 0) b_save_col(&wdpb[cur_entry], !is_p);
    if (dump_mv) {
        fwrite(mvs, 2, (size_t)NMB * 32, dump_mv); fwrite(refs, 1, (size_t)NMB * 16, dump_mv);
        if (opt_bframes) {
            /* list 1, then both lists as the writer means them (one byte n, n 16-bit picture numbers each), then the
             * implicit weight of the list-0 prediction for every pair of indices (16-bit each, n0 x n1) */
            fwrite(mvs1, 2, (size_t)NMB * 32, dump_mv); fwrite(refs1, 1, (size_t)NMB * 16, dump_mv);
            const int n0 = is_b ? n_active : is_p ? wlist_n : 0, n1 = is_b ? n_active1 : 0;
            fputc(n0, dump_mv);
            for (int i = 0; i < n0; i++) { const int pn = is_b ? wdpb[blist[0][i]].pic : wlist[i]; fputc(pn & 255, dump_mv); fputc(pn >> 8, dump_mv); }
            fputc(n1, dump_mv);
            for (int i = 0; i < n1; i++) { const int pn = wdpb[blist[1][i]].pic; fputc(pn & 255, dump_mv); fputc(pn >> 8, dump_mv); }
            for (int i = 0; i < n0 && is_b; i++) for (int j = 0; j < n1; j++) { const int w = b_weight(i, j); fputc(w & 255, dump_mv); fputc((w >> 8) & 255, dump_mv); }
        }
        if (opt_reorder) fputc(slice_reordered, dump_mv);
        if (opt_mmco) { fputc(wlist_n, dump_mv); for (int i = 0; i < wlist_n; i++) { fputc(wlist[i] & 255, dump_mv); fputc(wlist[i] >> 8, dump_mv); } }
    }
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: synth264 out.264 [--mbw N --mbh N --frames N --gop N --seed N --intra-only ...]\n"); return 2; }
    int frames = 30, gop = 30, intra_only = 0, crop_bottom = 0;
    uint64_t seed = 1;
    W = 22; H = 18;
    for (int i = 2; i < argc; i++) {
        const char *a = argv[i];
        int v = i + 1 < argc ? atoi(argv[i + 1]) : 0;
        if (!strcmp(a, "--mbw")) { W = v; i++; }
        else if (!strcmp(a, "--mbh")) { H = v; i++; }
        else if (!strcmp(a, "--frames")) { frames = v; i++; }
        else if (!strcmp(a, "--gop")) { gop = v; i++; }
        else if (!strcmp(a, "--seed")) { seed = (uint64_t)v; i++; }
        else if (!strcmp(a, "--qp")) { opt_qp = v; i++; }
        else if (!strcmp(a, "--coded")) { opt_coded = v; i++; }
        else if (!strcmp(a, "--maxlevel")) { opt_maxlevel = v; i++; }
        else if (!strcmp(a, "--mvmax")) { opt_mvmax = v; i++; }
        else if (!strcmp(a, "--cqo")) { opt_cqo = v; i++; }
        else if (!strcmp(a, "--crop-bottom")) { crop_bottom = v; i++; }
        else if (!strcmp(a, "--intra-only")) intra_only = 1;
        else if (!strcmp(a, "--nodeblock")) opt_deblock = 0;
        else if (!strcmp(a, "--refs")) { opt_refs = v; i++; }
        else if (!strcmp(a, "--slices")) { opt_slices = v < 1 ? 1 : v; i++; }
        else if (!strcmp(a, "--deblock-idc")) { opt_idc = v == 2 ? 2 : 0; i++; }
        else if (!strcmp(a, "--dump-mv")) { dump_mv = fopen(argv[i + 1], "wb"); i++; }
        else if (!strcmp(a, "--qp-delta")) { opt_qpdelta = v < 0 ? -v : v; i++; }
        else if (!strcmp(a, "--deblock-offsets")) { opt_alpha = v; opt_beta = i + 2 < argc ? atoi(argv[i + 2]) : 0; i += 2; }
        else if (!strcmp(a, "--sub8x8")) opt_sub8x8 = 1;
        else if (!strcmp(a, "--reorder")) opt_reorder = 1;
        else if (!strcmp(a, "--pps-alt")) opt_pps_alt = 1;
        else if (!strcmp(a, "--bframes")) { opt_bframes = v; i++; }
        else if (!strcmp(a, "--cabac")) opt_cabac = 1;
        else if (!strcmp(a, "--temporal")) opt_temporal = 1;
        else if (!strcmp(a, "--implicit")) opt_implicit = 1;
        else if (!strcmp(a, "--d8inf")) opt_d8inf = 1;
        else if (!strcmp(a, "--mmco")) opt_mmco = 1;
        else if (!strcmp(a, "--mmco5")) opt_mmco = opt_mmco5 = 1;
        else { fprintf(stderr, "unknown option %s\n", a); return 2; }
    }
    if (W < 1 || H < 1 || W > 512 || H > 512 || frames < 1 || opt_qp < 0 || opt_qp > 51 || opt_refs < 1 || opt_refs > ((opt_mmco || opt_bframes) ? 4 : 2) || (opt_bframes && (opt_refs < 2 || opt_bframes > 4)) || opt_alpha < -6 || opt_alpha > 6 || opt_beta < -6 || opt_beta > 6) { fprintf(stderr, "bad geometry\n"); return 2; }
    NMB = W * H;
    g_rng = seed * 0x9e3779b97f4a7c15ull + 264;
    w_alloc();
    mb_type = calloc((size_t)NMB, 1); mvs = calloc((size_t)NMB * 32, 2); nnz = calloc((size_t)NMB, 24); i4m = calloc((size_t)NMB, 16); refs = calloc((size_t)NMB, 16);
    FILE *f = fopen(argv[1], "wb");
    if (!f) { perror(argv[1]); return 2; }
    const int log2_fn = 8;
    {   /* SPS: Baseline, POC type 2, one reference frame */
        bw_t b = { 0 };
        if (opt_bframes || opt_cabac) { bw_put(&b, 8, 77); bw_put(&b, 8, 0x40); bw_put(&b, 8, 40); }      /* Main profile */
        else { bw_put(&b, 8, 66); bw_put(&b, 8, 0xc0); bw_put(&b, 8, 40); }
        bw_ue(&b, 0); bw_ue(&b, log2_fn - 4);
        if (opt_bframes) { bw_ue(&b, 0); bw_ue(&b, 8 - 4); }   /* pic_order_cnt_type 0, log2_max_pic_order_cnt_lsb 8 */
        else bw_ue(&b, 2);
        bw_ue(&b, (uint32_t)opt_refs); bw_put(&b, 1, 0);    /* num_ref_frames */
        bw_ue(&b, (uint32_t)(W - 1)); bw_ue(&b, (uint32_t)(H - 1));
        bw_put(&b, 1, 1); bw_put(&b, 1, (uint32_t)(opt_bframes ? opt_d8inf : 1));    /* frame_mbs_only, direct_8x8_inference */
        if (crop_bottom) { bw_put(&b, 1, 1); bw_ue(&b, 0); bw_ue(&b, 0); bw_ue(&b, 0); bw_ue(&b, (uint32_t)crop_bottom); }
        else bw_put(&b, 1, 0);
        bw_put(&b, 1, 0);
        bw_trailing(&b);
        write_nal(f, 3, 7, &b); free(b.buf);
    }
    for (int pps = 0; pps <= opt_pps_alt; pps++) {   /* PPS: CAVLC, deblocking control present */
        bw_t b = { 0 };
        bw_ue(&b, (uint32_t)pps); bw_ue(&b, 0); bw_put(&b, 1, (uint32_t)opt_cabac); bw_put(&b, 1, 0); bw_ue(&b, 0);   /* ids, entropy_coding_mode, pic_order_present, slice groups */
        bw_ue(&b, 0); bw_ue(&b, 0); bw_put(&b, 1, 0); bw_put(&b, 2, (uint32_t)(opt_implicit ? 2 : 0));   /* weighted_pred 0, weighted_bipred_idc */
        bw_se(&b, opt_qp - 26); bw_se(&b, 0); bw_se(&b, opt_cqo);
        bw_put(&b, 1, 1); bw_put(&b, 1, 0); bw_put(&b, 1, 0);
        bw_trailing(&b);
        write_nal(f, 3, 8, &b); free(b.buf);
    }
    int frame_num = 0, idr_id = 0, since_idr = 0;
    if (opt_bframes) {
        /* decode order: the IDR picture, then groups of one P picture followed by the N B pictures that are displayed in front
         * of it; picture order count = 2 x display position; B pictures are not references, so they carry the frame_num the
         * next reference picture will carry too (7.4.3) */
        mvs1 = calloc((size_t)NMB * 32, 2); refs1 = malloc((size_t)NMB * 16); memset(refs1, -1, (size_t)NMB * 16);
        for (int n = 0; n < frames; n++) {
            const int idr = n == 0, pos = idr ? 0 : (n - 1) % (opt_bframes + 1), group = idr ? 0 : (n - 1) / (opt_bframes + 1) + 1;
            const int is_b = !idr && pos > 0;
            cur_poc = 2 * (is_b ? (group - 1) * (opt_bframes + 1) + pos : group * (opt_bframes + 1));
            cur_entry = -1;
            put_slice(f, idr, !idr && !is_b, is_b, frame_num, idr_id, log2_fn, since_idr, n);
            if (!is_b && (idr || pos == 0)) frame_num = (frame_num + 1) & ((1 << log2_fn) - 1);
        }
        fclose(f);
        if (dump_mv) fclose(dump_mv);
        return 0;
    }
    for (int n = 0; n < frames; n++) {
        int idr = intra_only || n == 0 || (gop > 0 && n % gop == 0);
        if (idr) { frame_num = 0; since_idr = 0; }
        cur_pps = opt_pps_alt ? (n & 1) : 0;
        had_mmco5 = 0;
        put_slice(f, idr, !idr, 0, frame_num, idr_id, log2_fn, since_idr, n);   /* since_idr = reference pictures available (sliding window) */
        since_idr++;
        if (idr) idr_id = (idr_id + 1) & 0xffff;
        frame_num = had_mmco5 ? 1 : (frame_num + 1) & ((1 << log2_fn) - 1);
    }
    fclose(f);
    if (dump_mv) fclose(dump_mv);
    return 0;
}
