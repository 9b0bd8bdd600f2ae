/* compact.c - the compact link format of a picture's arrays (include/p264hip.h): host side - packing, the header check
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
    const size_t n = (size_t)p->mb_w * (size_t)p->mb_h;
    return 64 + n * 16 + 16 + n * 4 + 16 + n / 4 + 32 + n * 64 + 16 + n * 16 + 16 + (size_t)p->n_coef_blocks / 8 + 32 + (size_t)p->n_coef_blocks * 32 + 16;
}

static int shape_of(const int32_t *v)                         /* 1 one vector, 2 one per quadrant, 3 sixteen */
{
    int whole = 1, quads = 1;
    for (int i = 1; i < 16; i++) whole &= v[i] == v[0];
    if (whole) return 1;
    for (int q = 0; q < 4; q++) { const int b = QUAD_FIRST[q]; quads &= v[b + 1] == v[b] && v[b + 4] == v[b] && v[b + 5] == v[b]; }
    return quads ? 2 : 3;
}

int64_t p264hip_pack_compact(const p264hip_picture_t *p, void *dst_, size_t cap)
{
    if (p264amd_cpu_refuse("p264hip_pack_compact")) return P264HIP_EINVAL;
    if (!p || !dst_ || p->mb_w < 1 || p->mb_h < 1 || p->slice_type == P264_SLICE_B) return P264HIP_EINVAL;
    if (!p->mb || !p->mv || !p->ref_idx || !p->i4modes || (p->n_coef_blocks && !p->coefs)) return P264HIP_EINVAL;
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
    h.magic = P264HIP_COMPACT_MAGIC; h.n_mb = (uint32_t)n; h.n_coef_blocks = p->n_coef_blocks;
    h.off_rec = 64;
    h.off_ref = up16(h.off_rec + (uint32_t)n * 16u);
    h.off_shape = up16(h.off_ref + (uint32_t)n * 4u);
    h.off_vec = up16(h.off_shape + (uint32_t)((n + 3) / 4));
    memcpy(dst + h.off_rec, p->mb, n * 16);
    memcpy(dst + h.off_ref, p->ref_idx, n * 4);
    uint8_t *shape = dst + h.off_shape;
    memset(shape, 0, (n + 3) / 4 + 16);
    int32_t *vec = (int32_t *)(dst + h.off_vec);
    const int32_t *mv = (const int32_t *)p->mv;
    uint32_t nv = 0, ni4 = 0;
    for (size_t i = 0; i < n; i++) {
        if (P264_MB_IS_INTRA(p->mb[i].mb_type)) { ni4 += p->mb[i].mb_type == P264_MB_I4x4; continue; }
        const int32_t *v = mv + i * 16;
        const int sh = shape_of(v);
        shape[i >> 2] |= (uint8_t)(sh << (2 * (i & 3)));
        if (sh == 1) vec[nv++] = v[0];
        else if (sh == 2) for (int q = 0; q < 4; q++) vec[nv++] = v[QUAD_FIRST[q]];
        else { memcpy(vec + nv, v, 64); nv += 16; }
    }
    h.n_vec = nv; h.n_i4 = ni4;
    h.off_i4 = up16(h.off_vec + nv * 4u);
    uint8_t *i4 = dst + h.off_i4;
    for (size_t i = 0, k = 0; i < n; i++) if (p->mb[i].mb_type == P264_MB_I4x4) memcpy(i4 + 16 * k++, p->i4modes + 16 * i, 16);
    h.off_lvflag = up16(h.off_i4 + ni4 * 16u);
    uint8_t *flag = dst + h.off_lvflag;
    const uint32_t nb = p->n_coef_blocks;
    memset(flag, 0, (nb + 7) / 8 + 16);
    h.off_levels = up16(h.off_lvflag + (nb + 7) / 8);
    uint8_t *lv = dst + h.off_levels;
    uint32_t at = 0;
    for (uint32_t b = 0; b < nb; b++) {
        const int16_t *c = p->coefs + (size_t)b * 16;
        int narrow = 1;
        for (int k = 0; k < 16; k++) narrow &= c[k] >= -128 && c[k] <= 127;
        if (narrow) { flag[b >> 3] |= (uint8_t)(1u << (b & 7)); for (int k = 0; k < 16; k++) lv[at + k] = (uint8_t)(int8_t)c[k]; at += 16; }
        else { memcpy(lv + at, c, 32); at += 32; }
    }
    h.level_bytes = at;
    h.bytes = up16(h.off_levels + at) + 16;                   /* (+16: the expansion may read one piece past a narrow last block) */
    memset(lv + at, 0, h.bytes - (h.off_levels + at));
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
    if (!d || !compact || bytes < 64) return 0;
    p264hip_compact_hdr_t h;
    memcpy(&h, compact, sizeof h);
    const uint64_t n = (uint64_t)d->mb_w * (uint64_t)d->mb_h;
    if (h.magic != P264HIP_COMPACT_MAGIC || h.n_mb != n || n > P264HIP_COMPACT_MAX_MB || h.n_coef_blocks != d->n_coef_blocks || h.bytes != bytes) return 0;
    if (d->slice_type == P264_SLICE_B) return 0;
    if (h.n_vec > 16 * n || h.n_i4 > n || h.level_bytes > (uint64_t)h.n_coef_blocks * 32 || h.level_bytes < (uint64_t)h.n_coef_blocks * 16) return 0;
    if (h.off_rec != 64 || h.off_ref < h.off_rec + n * 16 || h.off_shape < h.off_ref + n * 4 || h.off_vec < h.off_shape + (n + 3) / 4
        || h.off_i4 < (uint64_t)h.off_vec + (uint64_t)h.n_vec * 4 || h.off_lvflag < (uint64_t)h.off_i4 + (uint64_t)h.n_i4 * 16
        || h.off_levels < (uint64_t)h.off_lvflag + (h.n_coef_blocks + 7) / 8 || (uint64_t)h.off_levels + h.level_bytes + 16 > h.bytes) return 0;
    if ((h.off_ref | h.off_shape | h.off_vec | h.off_i4 | h.off_lvflag | h.off_levels) & 15u) return 0;
    return 1;
}

/* Everything a block from an untrusted producer should pass before the device walks it: the header is consistent with the picture,
 * the sections lie inside the block in order, and the counts the expansion derives from the shape and flag bits are the
 * header's (so that no read of it leaves its section).  Also the records' coefficient ranges, as p264hip_upload checks them. */
int p264hip_compact_check(const p264hip_picture_t *d, const void *compact, size_t bytes)
{
    if (!p264hip_compact_header_ok(d, compact, bytes)) return P264HIP_EINVAL;
    const uint8_t *b = (const uint8_t *)compact;
    p264hip_compact_hdr_t h;
    memcpy(&h, b, sizeof h);
    const size_t n = (size_t)d->mb_w * (size_t)d->mb_h;
    const p264hip_mb_t *rec = (const p264hip_mb_t *)(b + h.off_rec);
    const uint8_t *shape = b + h.off_shape, *flag = b + h.off_lvflag;
    uint64_t nv = 0, ni4 = 0, lvb = 0;
    for (size_t i = 0; i < n; i++) {
        const int sh = (shape[i >> 2] >> (2 * (i & 3))) & 3;
        const int intra = P264_MB_IS_INTRA(rec[i].mb_type);
        if ((sh == 0) != (intra != 0)) return P264HIP_EINVAL;
        nv += sh == 1 ? 1 : sh == 2 ? 4 : sh == 3 ? 16 : 0;
        ni4 += rec[i].mb_type == P264_MB_I4x4;
        if (rec[i].coef_mask && (uint64_t)rec[i].coef_index + (uint64_t)__builtin_popcount(rec[i].coef_mask & 0x3ffffffu) > h.n_coef_blocks) return P264HIP_EINVAL;
    }
    for (uint32_t k = 0; k < h.n_coef_blocks; k++) lvb += ((flag[k >> 3] >> (k & 7)) & 1) ? 16 : 32;
    if (nv != h.n_vec || ni4 != h.n_i4 || lvb != h.level_bytes) return P264HIP_EINVAL;
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
    memcpy(out + L.off_ref, b + h.off_ref, n * 4);
    const p264hip_mb_t *rec = (const p264hip_mb_t *)(b + h.off_rec);
    const uint8_t *shape = b + h.off_shape, *i4 = b + h.off_i4, *flag = b + h.off_lvflag, *lv = b + h.off_levels;
    const int32_t *vec = (const int32_t *)(b + h.off_vec);
    int32_t *mv = (int32_t *)(out + L.off_mv);
    uint8_t *modes = out + L.off_i4;
    memset(modes, 2, n * 16);
    for (size_t i = 0; i < n; i++) {
        const int sh = (shape[i >> 2] >> (2 * (i & 3))) & 3;
        int32_t *v = mv + i * 16;
        if (sh == 1) { for (int k = 0; k < 16; k++) v[k] = vec[0]; vec += 1; }
        else if (sh == 2) { for (int q = 0; q < 4; q++) { const int f = QUAD_FIRST[q]; v[f] = v[f + 1] = v[f + 4] = v[f + 5] = vec[q]; } vec += 4; }
        else if (sh == 3) { memcpy(v, vec, 64); vec += 16; }
        if (rec[i].mb_type == P264_MB_I4x4) { memcpy(modes + 16 * i, i4, 16); i4 += 16; }
    }
    int16_t *co = (int16_t *)(out + L.off_coef);
    for (uint32_t k = 0; k < h.n_coef_blocks; k++) {
        if ((flag[k >> 3] >> (k & 7)) & 1) { for (int j = 0; j < 16; j++) co[(size_t)k * 16 + j] = (int16_t)(int8_t)lv[j]; lv += 16; }
        else { memcpy(co + (size_t)k * 16, lv, 32); lv += 32; }
    }
    return P264HIP_OK;
}
