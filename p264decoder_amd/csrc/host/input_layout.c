/* input_layout.c - the layout of a picture's arrays inside an input slot of the HIP layer (include/p264hip.h), and the host
 * side of it: packing a parsed picture into one block and viewing such a block as a picture again.  Pure host code (the
 * stream fan-out packs pictures on the rank that parses them, include/p264fan.h); the device side is p264hip.hip. */
#include <string.h>
#include "p264hip.h"
#include "host_cpu.h"

static size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

int p264hip_input_layout(const p264hip_picture_t *d, p264hip_input_layout_t *o)
{
    if (!d || !o || d->mb_w < 1 || d->mb_h < 1 || d->mb_w > 4096 || d->mb_h > 4096) return P264HIP_EINVAL;
    const size_t n = (size_t)d->mb_w * (size_t)d->mb_h;
    memset(o, 0, sizeof *o);
    o->off_mb = 0;
    o->off_mv = up256(n * sizeof(p264hip_mb_t));
    o->off_ref = o->off_mv + up256(n * 64);
    o->off_i4 = o->off_ref + up256(n * 4);
    o->off_coef = o->off_i4 + up256(n * 16);
    size_t end = o->off_coef + up256((size_t)d->n_coef_blocks * 32) + 256;      /* (+256: the kernels' 16-byte loads of the last block's tail) */
    if (d->slice_type == P264_SLICE_B) {
        o->off_mv_l1 = end;
        o->off_ref_l1 = o->off_mv_l1 + up256(n * 64);
        o->off_weights = o->off_ref_l1 + up256(n * 4);
        end = o->off_weights + 512;
    }
    o->bytes = end;
    return P264HIP_OK;
}

int64_t p264hip_pack_input(const p264hip_picture_t *p, void *dst_, size_t cap)
{
    if (p264amd_cpu_refuse("p264hip_pack_input")) return P264HIP_EINVAL;
    p264hip_input_layout_t L;
    if (!p || !dst_ || p264hip_input_layout(p, &L)) return P264HIP_EINVAL;
    if (cap < L.bytes) return P264HIP_ENOMEM;
    if (!p->mb || !p->mv || !p->ref_idx || !p->i4modes || (p->n_coef_blocks && !p->coefs)) return P264HIP_EINVAL;
    const int isB = p->slice_type == P264_SLICE_B;
    if (isB && (!p->mv_l1 || !p->ref_idx_l1)) return P264HIP_EINVAL;
    const size_t n = (size_t)p->mb_w * (size_t)p->mb_h;
    /* every macroblock's packed blocks must lie inside coefs[] (the kernels index it without further checks) */
    for (size_t i = 0; i < n; i++) {
        const p264hip_mb_t *m = &p->mb[i];
        if (m->coef_mask && (uint64_t)m->coef_index + (uint64_t)__builtin_popcount(m->coef_mask & 0x3ffffffu) > p->n_coef_blocks) return P264HIP_EINVAL;
    }
    uint8_t *dst = (uint8_t *)dst_;
    memcpy(dst + L.off_mb, p->mb, n * sizeof(p264hip_mb_t));
    memcpy(dst + L.off_mv, p->mv, n * 64);
    memcpy(dst + L.off_ref, p->ref_idx, n * 4);
    memcpy(dst + L.off_i4, p->i4modes, n * 16);
    if (p->n_coef_blocks) memcpy(dst + L.off_coef, p->coefs, (size_t)p->n_coef_blocks * 32);
    if (isB) {
        memcpy(dst + L.off_mv_l1, p->mv_l1, n * 64);
        memcpy(dst + L.off_ref_l1, p->ref_idx_l1, n * 4);
        memcpy(dst + L.off_weights, p->bipred_weight, sizeof p->bipred_weight);
    }
    return (int64_t)L.bytes;
}

int p264hip_unpack_input(const p264hip_picture_t *desc, const void *packed, size_t bytes, p264hip_picture_t *pic)
{
    if (p264amd_cpu_refuse("p264hip_unpack_input")) return P264HIP_EINVAL;
    p264hip_input_layout_t L;
    if (!desc || !packed || !pic || p264hip_input_layout(desc, &L) || bytes < L.bytes) return P264HIP_EINVAL;
    const uint8_t *b = (const uint8_t *)packed;
    *pic = *desc;
    pic->mb = (const p264hip_mb_t *)(b + L.off_mb);
    pic->mv = (const int16_t *)(b + L.off_mv);
    pic->ref_idx = (const int8_t *)(b + L.off_ref);
    pic->i4modes = b + L.off_i4;
    pic->coefs = (const int16_t *)(b + L.off_coef);
    pic->mv_l1 = NULL; pic->ref_idx_l1 = NULL;
    if (desc->slice_type == P264_SLICE_B) {
        pic->mv_l1 = (const int16_t *)(b + L.off_mv_l1);
        pic->ref_idx_l1 = (const int8_t *)(b + L.off_ref_l1);
    }
    return P264HIP_OK;
}
