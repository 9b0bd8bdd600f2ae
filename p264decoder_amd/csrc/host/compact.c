/* compact.c - the compact link format of a picture's arrays (include/p264hip.h): host side - packing, the checks
 * p264hip_upload_compact runs before it trusts a block, and the reference expansion back into the slot layout (what the
 * device kernel k_expand_compact does, csrc/hip/kernel_expand.h).  Pure host code.  What the format stands for in the
 * reference: nothing - p264_macroblock_decode reads h->mb.* / h->dct.* in place (core/core.h:330-341, 382-390). */
#include <string.h>
#include "p264hip.h"
#include "host_cpu.h"

static uint32_t up16(uint32_t v) { return (v + 15u) & ~15u; }
static const int QUAD_FIRST[4] = { 0, 2, 8, 10 };           /* first 4x4 block (raster in the macroblock) of quadrant q */

size_t p264hip_compact_bound(const p264hip_picture_t *p)
{
    if (!p || p->mb_w < 1 || p->mb_h < 1) return 0;
    const size_t n = (size_t)p->mb_w * (size_t)p->mb_h, lists = p->slice_type == P264_SLICE_B ? 2 : 1;
    return sizeof(p264hip_compact_hdr_t) + n * 16 + 16 + lists * (n * 4 + 16 + n / 4 + 32 + n * 64 + 16) + n / 8 + 32 + n * 16 + 16
           + (size_t)p->n_coef_blocks / 8 + 32 + (size_t)p->n_coef_blocks * 32 + 64 + 512 + 64;
}

/* 0 sixteen zero vectors, 1 one vector, 2 one per 8x8 quadrant, 3 sixteen */
static int shape_of(const int32_t *v)
{
    int whole = 1, quads = 1;
    for (int i = 1; i < 16; i++) whole &= v[i] == v[0];
    if (whole) return v[0] ? 1 : 0;
    for (int q = 0; q < 4; q++) { const int b = QUAD_FIRST[q]; quads &= v[b + 1] == v[b] && v[b + 4] == v[b] && v[b + 5] == v[b]; }
    return quads ? 2 : 3;
}
static uint32_t shape_words(int sh) { return sh == 1 ? 1u : sh == 2 ? 4u : sh == 3 ? 16u : 0u; }

int64_t p264hip_pack_compact(const p264hip_picture_t *p, void *dst_, size_t cap)
{
    if (p264amd_cpu_refuse("p264hip_pack_compact")) return P264HIP_EINVAL;
    if (!p || !dst_ || p->mb_w < 1 || p->mb_h < 1) return P264HIP_EINVAL;
    if (!p->mb || !p->mv || !p->ref_idx || !p->i4modes || (p->n_coef_blocks && !p->coefs)) return P264HIP_EINVAL;
    const int isB = p->slice_type == P264_SLICE_B;
    if (isB && (!p->mv_l1 || !p->ref_idx_l1)) return P264HIP_EINVAL;
    const size_t n = (size_t)p->mb_w * (size_t)p->mb_h;
    if (n > P264HIP_COMPACT_MAX_MB) return P264HIP_EINVAL;
    if (cap < p264hip_compact_bound(p)) return P264HIP_ENOMEM;
    for (size_t i = 0; i < n; i++) {                          /* as p264hip_pack_input: every macroblock's blocks inside coefs[] */
        const p264hip_mb_t *m = &p->mb[i];
        if (m->coef_mask && (uint64_t)m->coef_index + (uint64_t)__builtin_popcount(m->coef_mask & 0x3ffffffu) > p->n_coef_blocks) return P264HIP_EINVAL;
    }
    uint8_t *dst = (uint8_t *)dst_;
    p264hip_compact_hdr_t h;
    memset(&h, 0, sizeof h);
    h.magic = P264HIP_COMPACT_MAGIC; h.n_mb = (uint32_t)n; h.n_coef_blocks = p->n_coef_blocks; h.n_lists = isB ? 2u : 1u;
    h.off_rec = (uint32_t)sizeof h;
    memcpy(dst + h.off_rec, p->mb, n * 16);
    uint32_t at = up16(h.off_rec + (uint32_t)n * 16u);
    for (uint32_t l = 0; l < h.n_lists; l++) {
        const int32_t *mv = (const int32_t *)(l ? p->mv_l1 : p->mv);
        h.list[l].off_ref = at;
        memcpy(dst + at, l ? p->ref_idx_l1 : p->ref_idx, n * 4);
        at = up16(at + (uint32_t)n * 4u);
        h.list[l].off_shape = at;
        uint8_t *shape = dst + at;
        memset(shape, 0, (n + 3) / 4 + 16);
        at = up16(at + (uint32_t)((n + 3) / 4));
        h.list[l].off_vec = at;
        int32_t *vec = (int32_t *)(dst + at);
        uint32_t nv = 0;
        for (size_t i = 0; i < n; i++) {
            const int32_t *v = mv + i * 16;
            const int sh = shape_of(v);
            shape[i >> 2] |= (uint8_t)(sh << (2 * (i & 3)));
            if (sh == 1) vec[nv++] = v[0];
            else if (sh == 2) for (int q = 0; q < 4; q++) vec[nv++] = v[QUAD_FIRST[q]];
            else if (sh == 3) { memcpy(vec + nv, v, 64); nv += 16; }
        }
        h.list[l].n_vec = nv;
        at = up16(at + nv * 4u);
    }
    /* Intra4x4 modes: only the macroblocks whose sixteen modes are not all 2 (DC - what the parser leaves everywhere else) */
    h.off_i4flag = at;
    uint8_t *i4flag = dst + at;
    memset(i4flag, 0, (n + 7) / 8 + 16);
    at = up16(at + (uint32_t)((n + 7) / 8));
    h.off_i4 = at;
    uint32_t ni4 = 0;
    for (size_t i = 0; i < n; i++) {
        const uint8_t *m = p->i4modes + 16 * i;
        int plain = 1;
        for (int k = 0; k < 16; k++) plain &= m[k] == 2;
        if (plain) continue;
        i4flag[i >> 3] |= (uint8_t)(1u << (i & 7));
        memcpy(dst + at + 16 * ni4++, m, 16);
    }
    h.n_i4 = ni4;
    at = up16(at + ni4 * 16u);
    h.off_lvflag = at;
    uint8_t *flag = dst + at;
    const uint32_t nb = p->n_coef_blocks;
    memset(flag, 0, (nb + 7) / 8 + 16);
    at = up16(at + (nb + 7) / 8);
    h.off_levels = at;
    uint8_t *lv = dst + at;
    uint32_t used = 0;
    for (uint32_t b = 0; b < nb; b++) {
        const int16_t *c = p->coefs + (size_t)b * 16;
        int narrow = 1;
        for (int k = 0; k < 16; k++) narrow &= c[k] >= -128 && c[k] <= 127;
        if (narrow) { flag[b >> 3] |= (uint8_t)(1u << (b & 7)); for (int k = 0; k < 16; k++) lv[used + k] = (uint8_t)(int8_t)c[k]; used += 16; }
        else { memcpy(lv + used, c, 32); used += 32; }
    }
    h.level_bytes = used;
    memset(lv + used, 0, 32);                                 /* (the expansion reads 32 bytes at a block's place whatever its width) */
    at = up16(at + used) + 32;
    if (isB) { h.off_weights = at; memcpy(dst + at, p->bipred_weight, 512); at += 512; }
    h.bytes = at;
    memcpy(dst, &h, sizeof h);
    return (int64_t)h.bytes;
}

/* What p264hip_upload_compact checks (O(1)): the header belongs to this picture and its sections lie inside the block, in order,
 * each at least as large as the header's own counts say.  With that the device expansion cannot read or write outside the
 * block or the slot whatever the bits say (it clamps its places to the sections: kernel_expand.h) - a block whose shape / flag
 * bits do not add up to the header's counts gives a wrong picture, not a fault: like p264hip_upload_packed, the call trusts the
 * block to come from p264hip_pack_compact; p264hip_compact_check below is the full check for blocks from anywhere else. */
int p264hip_compact_header_ok(const p264hip_picture_t *d, const void *compact, size_t bytes)
{
    if (!d || !compact || bytes < sizeof(p264hip_compact_hdr_t)) return 0;
    p264hip_compact_hdr_t h;
    memcpy(&h, compact, sizeof h);
    const uint64_t n = (uint64_t)d->mb_w * (uint64_t)d->mb_h;
    if (h.magic != P264HIP_COMPACT_MAGIC || h.n_mb != n || n < 1 || n > P264HIP_COMPACT_MAX_MB || h.n_coef_blocks != d->n_coef_blocks || h.bytes != bytes) return 0;
    if (h.n_lists != (d->slice_type == P264_SLICE_B ? 2u : 1u)) return 0;
    if (h.n_i4 > n || h.level_bytes > (uint64_t)h.n_coef_blocks * 32 || h.level_bytes < (uint64_t)h.n_coef_blocks * 16) return 0;
    uint64_t at = sizeof h;
    if (h.off_rec != at) return 0;
    at += n * 16;
    for (uint32_t l = 0; l < h.n_lists; l++) {
        if (h.list[l].n_vec > 16 * n) return 0;
        if (h.list[l].off_ref < at) return 0;
        at = (uint64_t)h.list[l].off_ref + n * 4;
        if (h.list[l].off_shape < at) return 0;
        at = (uint64_t)h.list[l].off_shape + (n + 3) / 4;
        if (h.list[l].off_vec < at) return 0;
        at = (uint64_t)h.list[l].off_vec + (uint64_t)h.list[l].n_vec * 4;
        if ((h.list[l].off_ref | h.list[l].off_shape | h.list[l].off_vec) & 15u) return 0;
    }
    if (h.off_i4flag < at) return 0;
    at = (uint64_t)h.off_i4flag + (n + 7) / 8;
    if (h.off_i4 < at) return 0;
    at = (uint64_t)h.off_i4 + (uint64_t)h.n_i4 * 16;
    if (h.off_lvflag < at) return 0;
    at = (uint64_t)h.off_lvflag + (h.n_coef_blocks + 7) / 8;
    if (h.off_levels < at) return 0;
    at = (uint64_t)h.off_levels + h.level_bytes + 32;
    if ((h.off_i4flag | h.off_i4 | h.off_lvflag | h.off_levels | h.off_weights) & 15u) return 0;
    if (h.n_lists == 2) { if (h.off_weights < at) return 0; at = (uint64_t)h.off_weights + 512; }
    else if (h.off_weights) return 0;
    return at <= h.bytes;
}

/* Everything a block from an untrusted producer should pass before the device walks it: the header (above), the counts the
 * expansion derives from the shape and flag bits are the header's, the records' coefficient ranges as p264hip_upload checks
 * them, a B picture's weights inside -64 .. 128. */
int p264hip_compact_check(const p264hip_picture_t *d, const void *compact, size_t bytes)
{
    if (!p264hip_compact_header_ok(d, compact, bytes)) return P264HIP_EINVAL;
    const uint8_t *b = (const uint8_t *)compact;
    p264hip_compact_hdr_t h;
    memcpy(&h, b, sizeof h);
    const size_t n = (size_t)d->mb_w * (size_t)d->mb_h;
    const p264hip_mb_t *rec = (const p264hip_mb_t *)(b + h.off_rec);
    for (size_t i = 0; i < n; i++)
        if (rec[i].coef_mask && (uint64_t)rec[i].coef_index + (uint64_t)__builtin_popcount(rec[i].coef_mask & 0x3ffffffu) > h.n_coef_blocks) return P264HIP_EINVAL;
    for (uint32_t l = 0; l < h.n_lists; l++) {
        const uint8_t *shape = b + h.list[l].off_shape;
        uint64_t nv = 0;
        for (size_t i = 0; i < n; i++) nv += shape_words((shape[i >> 2] >> (2 * (i & 3))) & 3);
        if (nv != h.list[l].n_vec) return P264HIP_EINVAL;
    }
    const uint8_t *i4flag = b + h.off_i4flag, *flag = b + h.off_lvflag;
    uint64_t ni4 = 0, lvb = 0;
    for (size_t i = 0; i < n; i++) ni4 += (i4flag[i >> 3] >> (i & 7)) & 1;
    for (uint32_t k = 0; k < h.n_coef_blocks; k++) lvb += ((flag[k >> 3] >> (k & 7)) & 1) ? 16 : 32;
    if (ni4 != h.n_i4 || lvb != h.level_bytes) return P264HIP_EINVAL;
    if (h.n_lists == 2) {
        int16_t w[256];
        memcpy(w, b + h.off_weights, 512);
        if (d->weighted_bipred) for (int i = 0; i < 256; i++) if (w[i] < -64 || w[i] > 128) return P264HIP_EINVAL;
    }
    return P264HIP_OK;
}

int p264hip_expand_compact(const p264hip_picture_t *d, const void *compact, size_t bytes, void *out_, size_t cap)
{
    if (p264amd_cpu_refuse("p264hip_expand_compact")) return P264HIP_EINVAL;
    p264hip_input_layout_t L;
    if (!out_ || p264hip_compact_check(d, compact, bytes) || p264hip_input_layout(d, &L) || cap < L.bytes) return P264HIP_EINVAL;
    const uint8_t *b = (const uint8_t *)compact;
    uint8_t *out = (uint8_t *)out_;
    p264hip_compact_hdr_t h;
    memcpy(&h, b, sizeof h);
    const size_t n = h.n_mb;
    memset(out, 0, L.bytes);
    memcpy(out + L.off_mb, b + h.off_rec, n * 16);
    for (uint32_t l = 0; l < h.n_lists; l++) {
        memcpy(out + (l ? L.off_ref_l1 : L.off_ref), b + h.list[l].off_ref, n * 4);
        const uint8_t *shape = b + h.list[l].off_shape;
        const int32_t *vec = (const int32_t *)(b + h.list[l].off_vec);
        int32_t *mv = (int32_t *)(out + (l ? L.off_mv_l1 : L.off_mv));
        for (size_t i = 0; i < n; i++) {
            const int sh = (shape[i >> 2] >> (2 * (i & 3))) & 3;
            int32_t *v = mv + i * 16;
            if (sh == 1) { for (int k = 0; k < 16; k++) v[k] = vec[0]; vec += 1; }
            else if (sh == 2) { for (int q = 0; q < 4; q++) { const int f = QUAD_FIRST[q]; v[f] = v[f + 1] = v[f + 4] = v[f + 5] = vec[q]; } vec += 4; }
            else if (sh == 3) { memcpy(v, vec, 64); vec += 16; }
        }
    }
    const uint8_t *i4flag = b + h.off_i4flag, *i4 = b + h.off_i4, *flag = b + h.off_lvflag, *lv = b + h.off_levels;
    uint8_t *modes = out + L.off_i4;
    memset(modes, 2, n * 16);
    for (size_t i = 0; i < n; i++) if ((i4flag[i >> 3] >> (i & 7)) & 1) { memcpy(modes + 16 * i, i4, 16); i4 += 16; }
    int16_t *co = (int16_t *)(out + L.off_coef);
    for (uint32_t k = 0; k < h.n_coef_blocks; k++) {
        if ((flag[k >> 3] >> (k & 7)) & 1) { for (int j = 0; j < 16; j++) co[(size_t)k * 16 + j] = (int16_t)(int8_t)lv[j]; lv += 16; }
        else { memcpy(co + (size_t)k * 16, lv, 32); lv += 32; }
    }
    if (h.n_lists == 2) memcpy(out + L.off_weights, b + h.off_weights, 512);
    return P264HIP_OK;
}
