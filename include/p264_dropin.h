/* p264_dropin.h - the decode half of the reference's public API, served by the MI355X build.
 *
 * libp264amd.so exports these symbols with the reference's names, argument meaning and error
 * convention (0 / -1, message on stderr), so a program written against the reference's p264.h
 * (p264.h:266 p264_param_default, :351 p264_nal_decode, :379-382 p264_decoder_open / _close /
 * _decode) links against it unchanged.  The structures below are layout-compatible with
 * p264.h:111-341: same members, same order, same types - that is the binary contract, the
 * member names are the source contract.  A caller may equally keep including the reference's
 * own header; INTEGRATION.md shows both.
 *
 * Behavioural notes (all as in the reference unless stated):
 *   - p264_nal_decode: caller owns nal->p_payload (>= input size); strips 00 00 03 except when
 *     the 03 is among the last three bytes (core/core.c:310-336).
 *   - p264_decoder_decode: one NAL per call; *pp_pic is NULL or a decoder-owned picture with
 *     MB-aligned i_width/i_height (cropping is not applied, decoder/decoder.c:313-314), planes
 *     in host memory with strides width+64 and (width+64)/2 (core/frame.c:42-63), valid until
 *     the second-next picture is output.  Output order = decode order.
 *   - p264_decoder_open fails (NULL) when no HIP device is present: there is no CPU fallback.
 *   - encoder entry points of p264.h:359-372 are not provided (the reference does not define them either).
 */
#ifndef P264_DROPIN_H
#define P264_DROPIN_H
#include <stdint.h>
#include <stdarg.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct p264_t p264_t;

#define P264_CSP_I420   0x0001
#define P264_LOG_ERROR   0
#define P264_LOG_WARNING 1
#define P264_LOG_INFO    2
#define P264_LOG_DEBUG   3
#define P264_CQM_FLAT   0

enum { NAL_UNKNOWN = 0, NAL_SLICE = 1, NAL_SLICE_DPA = 2, NAL_SLICE_DPB = 3, NAL_SLICE_DPC = 4,
       NAL_SLICE_IDR = 5, NAL_SEI = 6, NAL_SPS = 7, NAL_PPS = 8, NAL_AUD = 9 };

typedef struct { int i_start, i_end, b_force_qp, i_qp; float f_bitrate_factor; } p264_zone_t;

typedef struct {
    unsigned int cpu; int i_threads;
    int i_width, i_height, i_csp, i_level_idc, i_frame_total;
    struct { int i_sar_height, i_sar_width, i_overscan, i_vidformat, b_fullrange, i_colorprim, i_transfer, i_colmatrix, i_chroma_loc; } vui;
    int i_fps_num, i_fps_den;
    int i_frame_reference, i_keyint_max, i_keyint_min, i_scenecut_threshold;
    int i_bframe, b_bframe_adaptive, i_bframe_bias, b_bframe_pyramid;
    int b_deblocking_filter, i_deblocking_filter_alphac0, i_deblocking_filter_beta;
    int b_cabac, i_cabac_init_idc;
    int i_cqm_preset; char *psz_cqm_file;
    uint8_t cqm_4iy[16], cqm_4ic[16], cqm_4py[16], cqm_4pc[16], cqm_8iy[64], cqm_8py[64];
    void (*pf_log)(void *, int i_level, const char *psz, va_list);
    void *p_log_private; int i_log_level; int b_visualize;
    struct {
        unsigned int intra, inter;
        int b_transform_8x8, b_weighted_bipred, i_direct_mv_pred, i_chroma_qp_offset;
        int i_me_method, i_me_range, i_mv_range, i_subpel_refine, b_chroma_me, b_bframe_rdo;
        int b_mixed_references, i_trellis, b_fast_pskip, b_psnr;
    } analyse;
    struct {
        int i_qp_constant, i_qp_min, i_qp_max, i_qp_step;
        int b_cbr, i_bitrate, i_rf_constant; float f_rate_tolerance;
        int i_vbv_max_bitrate, i_vbv_buffer_size; float f_vbv_buffer_init, f_ip_factor, f_pb_factor;
        int b_stat_write; char *psz_stat_out; int b_stat_read; char *psz_stat_in;
        char *psz_rc_eq; float f_qcompress, f_qblur, f_complexity_blur;
        p264_zone_t *zones; int i_zones; char *psz_zones;
    } rc;
    int b_aud, b_repeat_headers;
} p264_param_t;

typedef struct { int i_csp, i_plane; int i_stride[4]; uint8_t *plane[4]; } p264_image_t;

typedef struct {
    int i_type, i_qpplus1; int64_t i_pts;
    int i_width, i_height;            /* decoder output: MB-aligned picture size */
    p264_image_t img;
} p264_picture_t;

typedef struct { int i_ref_idc, i_type, i_payload; uint8_t *p_payload; } p264_nal_t;

void    p264_param_default(p264_param_t *param);
int     p264_nal_decode(p264_nal_t *nal, void *buf, int size);
p264_t *p264_decoder_open(p264_param_t *param);
int     p264_decoder_decode(p264_t *h, p264_picture_t **pp_pic, p264_nal_t *nal);
void    p264_decoder_close(p264_t *h);
void p264_picture_alloc(p264_picture_t *pic, int i_csp, int i_width, int i_height);     /* p264.h:300, core/core.c:181-259 */
void p264_picture_clean(p264_picture_t *pic);                                          /* p264.h:305, core/core.c:266-272 */
int64_t p264_mdate(void);            /* microsecond clock used by the reference CLI (core/mdate.c:40-52) */

/* Extension (not in the reference): environment knobs read by p264_decoder_open -
 *   P264AMD_DEVICE=<n>  HIP device index (default 0)
 *   P264AMD_QUIET=1     suppress the SPS/PPS/size lines on stderr */

#ifdef __cplusplus
}
#endif
#endif
