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
#include <stdarg.h>
#include "p264_dropin.h"
#include "p264parse.h"
#include "p264hip.h"
#include "host_cpu.h"

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

/* core/core.c:153-176 */
static void log_default(void *unused, int i_level, const char *psz_fmt, va_list arg)
{
    (void)unused;
    const char *prefix = i_level == P264_LOG_ERROR ? "error" : i_level == P264_LOG_WARNING ? "warning" : i_level == P264_LOG_INFO ? "info" :
                         i_level == P264_LOG_DEBUG ? "debug" : "unknown";
    fprintf(stderr, "p264 [%s]: ", prefix);
    vfprintf(stderr, psz_fmt, arg);
}

/* core/core.c:41-137, field for field (tests/test_abi.py compares every value with the reference's own function).  The two
 * fields that cannot be equal: cpu (p264_cpu_detect() is 0 without the x86 paths) and pf_log (a function of ours with the
 * reference's behaviour). */
void p264_param_default(p264_param_t *param)
{
    memset(param, 0, sizeof *param);
    param->cpu = 0;
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
    param->rc.i_bitrate = 1000; param->rc.f_rate_tolerance = 1.0f; param->rc.f_vbv_buffer_init = 0.9f;
    param->rc.i_qp_constant = 26; param->rc.i_qp_min = 10; param->rc.i_qp_max = 51; param->rc.i_qp_step = 4;
    param->rc.f_ip_factor = 1.4f; param->rc.f_pb_factor = 1.3f;
    param->rc.psz_stat_out = "p264_2pass.log"; param->rc.psz_stat_in = "p264_2pass.log";
    param->rc.psz_rc_eq = "blurCplx^(1-qComp)";
    param->rc.f_qcompress = 0.6f; param->rc.f_qblur = 0.5f; param->rc.f_complexity_blur = 20;
    param->pf_log = log_default;
    param->i_log_level = P264_LOG_INFO;
    param->analyse.intra = 0x0001 | 0x0002;                       /* P264_ANALYSE_I4x4 | I8x8 (p264.h:59-60) */
    param->analyse.inter = 0x0001 | 0x0002 | 0x0010 | 0x0100;     /* | PSUB16x16 | BSUB16x16 */
    param->analyse.i_direct_mv_pred = 2;                          /* P264_DIRECT_PRED_TEMPORAL */
    param->analyse.i_me_method = 1;                               /* P264_ME_HEX */
    param->analyse.i_me_range = 16; param->analyse.i_subpel_refine = 5; param->analyse.b_chroma_me = 1;
    param->analyse.i_mv_range = -1; param->analyse.b_fast_pskip = 1; param->analyse.b_psnr = 1;
    param->i_cqm_preset = P264_CQM_FLAT;
    memset(param->cqm_4iy, 16, 16); memset(param->cqm_4ic, 16, 16);
    memset(param->cqm_4py, 16, 16); memset(param->cqm_4pc, 16, 16);
    memset(param->cqm_8iy, 16, 64); memset(param->cqm_8py, 16, 64);
    param->b_repeat_headers = 1;
}

/* core/core.c:181-259 (the packed and 4:2:2 / 4:4:4 layouts included, as the reference allocates them) */
void p264_picture_alloc(p264_picture_t *pic, int i_csp, int i_width, int i_height)
{
    pic->i_type = 0;                                /* P264_TYPE_AUTO */
    pic->i_qpplus1 = 0;
    pic->i_width = i_width; pic->i_height = i_height;
    pic->img.i_csp = i_csp;
    const size_t wh = (size_t)i_width * i_height;
    switch (i_csp & 0x00ff) {                       /* P264_CSP_MASK */
    case 0x0001: case 0x0004:                       /* I420, YV12 */
        pic->img.i_plane = 3;
        pic->img.plane[0] = (uint8_t *)malloc(3 * wh / 2);
        pic->img.plane[1] = pic->img.plane[0] + wh; pic->img.plane[2] = pic->img.plane[1] + wh / 4;
        pic->img.i_stride[0] = i_width; pic->img.i_stride[1] = i_width / 2; pic->img.i_stride[2] = i_width / 2;
        break;
    case 0x0002:                                    /* I422 */
        pic->img.i_plane = 3;
        pic->img.plane[0] = (uint8_t *)malloc(2 * wh);
        pic->img.plane[1] = pic->img.plane[0] + wh; pic->img.plane[2] = pic->img.plane[1] + wh / 2;
        pic->img.i_stride[0] = i_width; pic->img.i_stride[1] = i_width / 2; pic->img.i_stride[2] = i_width / 2;
        break;
    case 0x0003:                                    /* I444 */
        pic->img.i_plane = 3;
        pic->img.plane[0] = (uint8_t *)malloc(3 * wh);
        pic->img.plane[1] = pic->img.plane[0] + wh; pic->img.plane[2] = pic->img.plane[1] + wh;
        pic->img.i_stride[0] = i_width; pic->img.i_stride[1] = i_width; pic->img.i_stride[2] = i_width;
        break;
    case 0x0005:                                    /* YUYV */
        pic->img.i_plane = 1; pic->img.plane[0] = (uint8_t *)malloc(2 * wh); pic->img.i_stride[0] = 2 * i_width;
        break;
    case 0x0006: case 0x0007:                       /* RGB, BGR */
        pic->img.i_plane = 1; pic->img.plane[0] = (uint8_t *)malloc(3 * wh); pic->img.i_stride[0] = 3 * i_width;
        break;
    case 0x0008:                                    /* BGRA */
        pic->img.i_plane = 1; pic->img.plane[0] = (uint8_t *)malloc(4 * wh); pic->img.i_stride[0] = 4 * i_width;
        break;
    default:
        fprintf(stderr, "invalid CSP\n");
        pic->img.i_plane = 0;
        break;
    }
}

void p264_picture_clean(p264_picture_t *pic)
{
    free(pic->img.plane[0]);
    memset(pic, 0, sizeof *pic);                    /* just to be safe (core/core.c:266-272) */
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
    if (p264amd_cpu_refuse("p264_decoder_open")) return NULL;
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
    { const char *pp = getenv("P264AMD_PINNED_PARSE"); if (!pp || atoi(pp)) p264parse_set_allocator(h->parser, p264hip_host_alloc, p264hip_host_free); }      /* picture arrays in pinned memory */
    return h;
}

static void drop_device(p264_t *h)
{
    if (h->hip) { p264hip_destroy(h->hip); h->hip = NULL; }
    for (int i = 0; i < OUT_BUFS; i++) { p264hip_host_free(h->out_mem[i]); h->out_mem[i] = NULL; }
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
        h->out_mem[i] = (uint8_t *)p264hip_host_alloc(ysz + 2 * csz);      /* pinned: the plane copies are real DMA */
        if (!h->out_mem[i]) return -1;
        memset(h->out_mem[i], 0, ysz + 2 * csz);
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
    /* one wait per picture: the parser's arrays and the output planes are pinned, so upload, reconstruction, layout
     * conversion and the plane copies are enqueued back to back and p264hip_sync waits for all of them (the API hands the
     * planes of THIS picture back from THIS call, decoder/decoder.c:652-657: nothing can be deferred past the return) */
    if (p264hip_submit_async(h->hip, 0, pic) != P264HIP_OK) { fprintf(stderr, "p264amd: %s\n", p264hip_last_error()); return -1; }
    int w = h->mb_w * 16, hh = h->mb_h * 16, ys = w + 64, cs = ys / 2;
    size_t ysz = (size_t)ys * (hh + 64), csz = (size_t)cs * (hh / 2 + 32);
    uint8_t *base = h->out_mem[h->out_next];
    h->out_next = (h->out_next + 1) % OUT_BUFS;
    uint8_t *y = base + (size_t)ys * 32 + 32, *u = base + ysz + (size_t)cs * 16 + 16, *v = base + ysz + csz + (size_t)cs * 16 + 16;
    if (p264hip_read_frame_async(h->hip, 0, pic->dst_slot, y, ys, u, v, cs) != P264HIP_OK || p264hip_sync(h->hip) != P264HIP_OK) {
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
