/* dropin.c - the reference's public decode API (include/p264_dropin.h) on top of the host
 * parser and the HIP reconstruction layer.
 *
 * p264_decoder_decode = NAL switch of decoder/decoder.c:745-806: parameter sets and slice
 * parsing stay on the CPU (parser.c); a completed picture is handed to p264hip_submit and its
 * reconstructed planes are copied into decoder-owned host memory, because the API contract is
 * host pointers (decoder/decoder.c:652-657).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "p264_dropin.h"
#include "p264parse.h"
#include "p264hip.h"

#define OUT_BUFS 2

struct p264_t {
    p264_param_t param;
    p264parse   *parser;
    p264hip_ctx *hip;
    int          generation;
    int          mb_w, mb_h;
    p264_picture_t pic;
    uint8_t     *out_mem[OUT_BUFS];
    int          out_next;
    int          device;
};

/* core/core.c:41-137.  Only the fields a decoder (or its caller) reads are meaningful; the
 * encoder-only tuning defaults are left zero. */
void p264_param_default(p264_param_t *param)
{
    memset(param, 0, sizeof *param);
    param->cpu = 0;                                /* no x86 SIMD here: p264_cpu_detect() -> 0 */
    param->i_threads = 1;
    param->i_csp = P264_CSP_I420;
    param->vui.i_vidformat = 5; param->vui.i_colorprim = 2; param->vui.i_transfer = 2; param->vui.i_colmatrix = 2;
    param->i_fps_num = 25; param->i_fps_den = 1;
    param->i_level_idc = 51;
    param->i_frame_reference = 1;
    param->i_keyint_max = 250; param->i_keyint_min = 25;
    param->i_scenecut_threshold = 40; param->b_bframe_adaptive = 1;
    param->b_deblocking_filter = 1;
    param->b_cabac = 1;
    param->rc.i_qp_constant = 26; param->rc.i_qp_min = 10; param->rc.i_qp_max = 51; param->rc.i_qp_step = 4;
    param->i_log_level = P264_LOG_INFO;
    param->i_cqm_preset = P264_CQM_FLAT;
    memset(param->cqm_4iy, 16, 16); memset(param->cqm_4ic, 16, 16);
    memset(param->cqm_4py, 16, 16); memset(param->cqm_4pc, 16, 16);
    memset(param->cqm_8iy, 16, 64); memset(param->cqm_8py, 16, 64);
    param->b_repeat_headers = 1;
}

/* core/core.c:310-336, including its loop bound: a 00 00 03 whose 03 lies within the last three
 * bytes is copied through (SURVEY A-Q10). */
int p264_nal_decode(p264_nal_t *nal, void *buf, int size)
{
    const uint8_t *src = (const uint8_t *)buf, *end = src + size;
    uint8_t *dst = nal->p_payload;
    nal->i_type = src[0] & 0x1f;
    nal->i_ref_idc = (src[0] >> 5) & 3;
    src++;
    /* same result as the byte loop of core/core.c:318-334 (a 00 00 03 is only stripped while at least four bytes remain,
     * A-Q10), but the payload moves in runs between zero bytes found by memchr */
    while (src < end) {
        const uint8_t *z = (const uint8_t *)memchr(src, 0, (size_t)(end - src));
        if (!z) z = end;
        memcpy(dst, src, (size_t)(z - src)); dst += z - src; src = z;
        if (src >= end) break;
        if (src < end - 3 && src[1] == 0 && src[2] == 3) { *dst++ = 0; *dst++ = 0; src += 3; }
        else *dst++ = *src++;
    }
    nal->i_payload = (int)(dst - nal->p_payload);
    return 0;
}

int64_t p264_mdate(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (int64_t)ts.tv_sec * 1000000 + ts.tv_nsec / 1000;
}

p264_t *p264_decoder_open(p264_param_t *param)
{
    if (p264hip_device_count() < 1) {
        fprintf(stderr, "p264amd: no HIP device: the MI355X reconstruction path cannot run and there is no CPU fallback\n");
        return NULL;
    }
    p264_t *h = (p264_t *)calloc(1, sizeof *h);
    if (!h) return NULL;
    if (param) h->param = *param;
    const char *q = getenv("P264AMD_QUIET"), *d = getenv("P264AMD_DEVICE");
    h->device = d ? atoi(d) : 0;
    h->parser = p264parse_open((q && atoi(q)) ? P264PARSE_OPT_QUIET : 0);
    if (!h->parser) { free(h); return NULL; }
    return h;
}

static void drop_device(p264_t *h)
{
    if (h->hip) { p264hip_destroy(h->hip); h->hip = NULL; }
    for (int i = 0; i < OUT_BUFS; i++) { free(h->out_mem[i]); h->out_mem[i] = NULL; }
}

/* decoder/decoder.c:304-343: new geometry -> new frame store (device) and output planes (host) */
static int ensure_device(p264_t *h)
{
    int gen = p264parse_generation(h->parser);
    if (h->hip && gen == h->generation) return 0;
    drop_device(h);
    h->mb_w = p264parse_mb_width(h->parser); h->mb_h = p264parse_mb_height(h->parser);
    if (p264hip_create(&h->hip, h->device, h->mb_w, h->mb_h, 1, p264parse_slots(h->parser), 1) != P264HIP_OK) {
        fprintf(stderr, "p264amd: %s\n", p264hip_last_error());
        h->hip = NULL;
        return -1;
    }
    /* host planes with the reference's geometry: stride W+64, 32 (16) pad lines above and below */
    int w = h->mb_w * 16, hh = h->mb_h * 16, ys = w + 64, cs = ys / 2;
    size_t ysz = (size_t)ys * (hh + 64), csz = (size_t)cs * (hh / 2 + 32);
    for (int i = 0; i < OUT_BUFS; i++) {
        h->out_mem[i] = (uint8_t *)calloc(1, ysz + 2 * csz);
        if (!h->out_mem[i]) return -1;
    }
    h->generation = gen;
    h->param.i_width = w; h->param.i_height = hh;
    return 0;
}

int p264_decoder_decode(p264_t *h, p264_picture_t **pp_pic, p264_nal_t *nal)
{
    const p264hip_picture_t *pic = NULL;
    *pp_pic = NULL;
    int rc = p264parse_nal(h->parser, nal->i_type, nal->i_ref_idc, nal->p_payload, nal->i_payload, &pic);
    if (rc < 0) { fprintf(stderr, "p264amd: nal type %d decode failed\n", nal->i_type); return -1; }
    if (rc == 0) return 0;
    if (ensure_device(h) < 0) return -1;
    if (p264hip_submit(h->hip, 0, pic) != P264HIP_OK) { fprintf(stderr, "p264amd: %s\n", p264hip_last_error()); return -1; }
    int w = h->mb_w * 16, hh = h->mb_h * 16, ys = w + 64, cs = ys / 2;
    size_t ysz = (size_t)ys * (hh + 64), csz = (size_t)cs * (hh / 2 + 32);
    uint8_t *base = h->out_mem[h->out_next];
    h->out_next = (h->out_next + 1) % OUT_BUFS;
    uint8_t *y = base + (size_t)ys * 32 + 32, *u = base + ysz + (size_t)cs * 16 + 16, *v = base + ysz + csz + (size_t)cs * 16 + 16;
    if (p264hip_read_frame(h->hip, 0, pic->dst_slot, y, ys, u, v, cs) != P264HIP_OK) {
        fprintf(stderr, "p264amd: %s\n", p264hip_last_error());
        return -1;
    }
    p264_picture_t *o = &h->pic;
    memset(o, 0, sizeof *o);
    o->i_width = w; o->i_height = hh;
    o->img.i_csp = P264_CSP_I420; o->img.i_plane = 3;
    o->img.i_stride[0] = ys; o->img.i_stride[1] = cs; o->img.i_stride[2] = cs;
    o->img.plane[0] = y; o->img.plane[1] = u; o->img.plane[2] = v;
    *pp_pic = o;
    return 0;
}

void p264_decoder_close(p264_t *h)
{
    if (!h) return;
    drop_device(h);
    p264parse_close(h->parser);
    free(h);
}
