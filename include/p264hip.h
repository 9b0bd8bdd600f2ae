/* p264hip.h - C ABI of the MI355X (gfx950) macroblock-reconstruction layer.
 *
 * This is the CPU->GPU seam of the decoder: the host keeps the serial CAVLC
 * parse and hands one p264hip_picture_t per coded picture to this layer, which
 * runs dequant + inverse transforms, intra prediction, motion compensation and
 * the in-loop deblocking filter as hand-written HIP kernels.
 *
 * What each entry point stands in for in the reference (lspbeyond/p264decoder):
 *   p264hip_picture_t / p264hip_mb_t  <- what p264_macroblock_decode and the deblocking
 *        driver read from h->mb.* / h->dct.*  (core/core.h:330-341, 344-463)
 *   p264hip_submit / p264hip_reconstruct <- the per-MB reconstruction driver
 *        decoder/macroblock.c:755-934 plus the picture post-process sequence
 *        decoder/decoder.c:635-661 (deblock; the border pads and half-pel planes are
 *        replaced by clamped on-the-fly interpolation inside the MC kernel)
 *   p264hip_read_frame  <- the plane aliasing at decoder/decoder.c:652-657
 *   frame slots          <- h->frames.reference[] (decoder/lists.c:152-228)
 *
 * Plain C types only (no torch / C++ types).  All functions return 0 on success and a
 * negative P264HIP_E* code on failure; p264hip_last_error() gives a message.
 */
#ifndef P264HIP_H
#define P264HIP_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P264HIP_MAX_REFS 16

/* macroblock classes (our own numbering; cf. core/macroblock.h:42-65) */
enum {
    P264_MB_I4x4   = 0,
    P264_MB_I16x16 = 1,
    P264_MB_IPCM   = 2,   /* reserved: rejected by the reference (decoder/macroblock.c:510-514) */
    P264_MB_P_L0   = 3,   /* 16x16, 16x8, 8x16 */
    P264_MB_P_8x8  = 4,
    P264_MB_P_SKIP = 5,
    P264_MB_B      = 6    /* any inter macroblock of a B picture (B_L0 / B_L1 / B_Bi / B_8x8 / direct / skip): which list(s)
                             an 8x8 quadrant predicts from is given by ref_idx[] / ref_idx_l1[] (>= 0 = used); the
                             reference's partition walk is core/macroblock.c:525-631, 677-765 */
};
#define P264_MB_IS_INTRA(t) ((t) <= P264_MB_IPCM)

enum { P264_SLICE_P = 0, P264_SLICE_B = 1, P264_SLICE_I = 2 };

enum {
    P264HIP_OK = 0,
    P264HIP_EINVAL = -1,
    P264HIP_ENODEV = -2,      /* no usable HIP device: the product never falls back to the CPU */
    P264HIP_ENOMEM = -3,
    P264HIP_EHIP = -4
};

/* coef_mask bits */
#define P264_COEF_LUMA_DC   (1u << 24)   /* Intra16x16 DC block present (16 levels, zig-zag order) */
#define P264_COEF_CHROMA_DC (1u << 25)   /* one packed block: Cb DC[0..3], Cr DC[4..7] (raster 2x2) */

/* One per macroblock, 16 bytes.  Everything is as parsed (before dequantisation). */
typedef struct p264hip_mb {
    uint8_t  mb_type;      /* P264_MB_* */
    uint8_t  qp;           /* luma QP as the reference stores it for the MB (core/macroblock.c:1247-1252) */
    uint8_t  cbp;          /* bits 0-3: luma 8x8 coded, bits 4-5: chroma (0 none, 1 DC, 2 DC+AC) */
    uint8_t  intra_modes;  /* bits 0-1 intra16x16_pred_mode, bits 4-5 intra_chroma_pred_mode (as coded) */
    uint32_t coef_mask;    /* bit b<24: 4x4 block b has total_coeff>0 (0-15 luma in decode order
                              core/macroblock.h:194-201, 16-19 Cb, 20-23 Cr); | P264_COEF_* */
    uint32_t coef_index;   /* index, in 16-level blocks, of this MB's first packed block */
    uint8_t  avail;        /* P264_AVAIL_*: neighbouring MBs usable for intra prediction
                              (core/macroblock.c:926-1033; picture border / slice membership) */
    uint8_t  edges;        /* P264_EDGE_*: which MB edges the loop filter touches (core/frame.c:524) */
    uint16_t flags;        /* reserved, 0 */
} p264hip_mb_t;

#define P264_AVAIL_LEFT     1
#define P264_AVAIL_TOP      2
#define P264_AVAIL_TOPRIGHT 4
#define P264_AVAIL_TOPLEFT  8
#define P264_EDGE_LEFT      1   /* filter the left MB edge  */
#define P264_EDGE_TOP       2   /* filter the top MB edge   */
#define P264_EDGE_INNER     4   /* filter the inner edges   */

/* Packed coefficient stream: for each MB, in this order and only when present:
 *   [luma DC][chroma DC][block 0]...[block 23]; each entry is int16[16] in scan order
 *   (AC-only blocks - I16x16 luma and all chroma - hold their 15 levels in [0..14], [15]=0). */

typedef struct p264hip_picture {
    int32_t  mb_w, mb_h;
    int32_t  slice_type;            /* P264_SLICE_P / P264_SLICE_I */
    int32_t  chroma_qp_offset;      /* pps chroma_qp_index_offset */
    int32_t  deblock;               /* run the loop filter (decoder/decoder.c:639) */
    int32_t  alpha_c0_offset;       /* used unshifted, as the reference does (core/frame.c:476-478) */
    int32_t  beta_offset;
    int32_t  dst_slot;              /* frame-store slot reconstructed into */
    int32_t  n_ref;                 /* list-0 length */
    int32_t  ref_slot[P264HIP_MAX_REFS];
    uint32_t n_coef_blocks;         /* entries in coefs[] */
    uint32_t frame_num;             /* informational */
    const p264hip_mb_t *mb;         /* [mb_w*mb_h] raster order */
    const int16_t      *mv;         /* [mb][16][2] quarter-pel, 4x4 blocks in raster order inside the MB */
    const int8_t       *ref_idx;    /* [mb][4] per 8x8 (raster), -1 for intra */
    const uint8_t      *i4modes;    /* [mb][16] Intra4x4PredMode per block in decode order (0..8) */
    const int16_t      *coefs;      /* [n_coef_blocks][16] */
    /* ---- B pictures only (slice_type == P264_SLICE_B; ignored otherwise).  A quadrant whose list-X index is negative
     * does not predict from list X and its list-X vectors must be 0 (the loop filter compares them, core/frame.c:565-577);
     * both indices >= 0 = bi-prediction: the two predictions are combined as the reference does (core/macroblock.c:543-583):
     * weighted_bipred == 0: (p0 + p1 + 1) >> 1 (core/mc.c:76-88); != 0: clip((p0 * w + p1 * (64 - w) + 32) >> 6) with
     * w = bipred_weight[ref_idx * 16 + ref_idx_l1] (core/mc.c:106-132; the implicit weights of core/macroblock.c:1400-1430,
     * computed by the host from the picture order counts). */
    const int16_t      *mv_l1;      /* [mb][16][2] */
    const int8_t       *ref_idx_l1; /* [mb][4] */
    int32_t             n_ref_l1;   /* list-1 length */
    int32_t             weighted_bipred;
    int32_t             ref_slot_l1[P264HIP_MAX_REFS];
    int16_t             bipred_weight[P264HIP_MAX_REFS * P264HIP_MAX_REFS];   /* [ref_idx][ref_idx_l1], -64 .. 128 */
} p264hip_picture_t;

typedef struct p264hip_ctx p264hip_ctx;

/* Device context: n_streams independent frame stores of `slots` frames each (4:2:0, MB-aligned,
 * no padding; kept macroblock-tiled in HBM - the host only ever sees planes, through
 * p264hip_read_frame / p264hip_write_frame) and `max_pictures` device-resident picture inputs. */
int  p264hip_create(p264hip_ctx **out, int device, int mb_w, int mb_h,
                    int n_streams, int slots, int max_pictures);
void p264hip_destroy(p264hip_ctx *ctx);
const char *p264hip_last_error(void);
int  p264hip_device_count(void);

/* Copy parsed pictures host->HBM into resident input slots [first, first+n). Asynchronous. */
int  p264hip_upload(p264hip_ctx *ctx, int first, const p264hip_picture_t *pics, int n);
/* The same without the final wait: the copies are only enqueued.  The picture's arrays must stay untouched until
 * a marker taken after this call has been reached (p264hip_marker / p264hip_marker_wait) or p264hip_sync returns;
 * they should live in pinned memory (p264hip_host_alloc) for the copies to be real DMA transfers. */
int  p264hip_upload_async(p264hip_ctx *ctx, int slot, const p264hip_picture_t *pic);
/* Both notice arrays that already lie in host memory the way an input slot is laid out (p264hip_input_layout_t below: the
 * parser of this library builds its pictures like that) and then copy records, vectors, indices, modes and levels in ONE
 * transfer instead of five.  Host -> HBM copies queued so far by the two calls (diagnostic; -1 without a context): */
int64_t p264hip_upload_copies(p264hip_ctx *ctx);
/* Pinned host memory for picture inputs and frame downloads; usable without a context.  Ordinary (huge) pages registered with
 * the runtime - the CPU writes and re-reads them at full speed, which it does not on hipHostMalloc'ed memory;
 * P264AMD_HOST_ALLOC = 0 gives hipHostMalloc back. */
void *p264hip_host_alloc(size_t bytes);
void  p264hip_host_free(void *p);
/* A marker is a point in the context's stream; marker_wait blocks the calling host thread until everything
 * enqueued before the marker has completed.  Markers are small integers >= 0 (a ring of P264HIP_MARKERS). */
#define P264HIP_MARKERS 8
int  p264hip_marker(p264hip_ctx *ctx);
int  p264hip_marker_wait(p264hip_ctx *ctx, int marker);
/* Device-side copy of resident picture `src` into slot `dst` (private HBM copy; bench set-up). */
int  p264hip_clone_picture(p264hip_ctx *ctx, int dst, int src);

/* ---- the layout of a picture's arrays inside its input slot, and ways into the slot that do not pass through
 * p264hip_upload's five copies.  An input slot is ONE block of device memory with the arrays at 256-byte aligned offsets
 * (a pure function of mb_w, mb_h, n_coef_blocks and slice_type).  A host buffer packed in this layout goes into a slot
 * with one copy (p264hip_upload_packed); a producer that can write device memory itself - the RCCL transport of the
 * stream fan-out receiving a picture from another rank (include/p264fan.h) - asks for the slot's address
 * (p264hip_input_reserve), fills it and commits it.  Nothing of this has a counterpart in the reference. */
typedef struct p264hip_input_layout {
    size_t off_mb, off_mv, off_ref, off_i4, off_coef;      /* p264hip_mb_t[n], int16[n][16][2], int8[n][4], uint8[n][16], int16[n_coef_blocks][16] */
    size_t off_mv_l1, off_ref_l1, off_weights;             /* B pictures only (0 otherwise): list-1 vectors, indices, bipred_weight[] */
    size_t bytes;                                          /* of the whole block */
} p264hip_input_layout_t;
int  p264hip_input_layout(const p264hip_picture_t *desc, p264hip_input_layout_t *out);
/* host side, no device involved: the picture's arrays copied into `dst` (cap >= layout.bytes) in that layout; the
 * macroblock records are checked as p264hip_upload checks them (coefficient ranges inside coefs[]).  Returns the bytes used
 * or a negative P264HIP_E* code. */
int64_t p264hip_pack_input(const p264hip_picture_t *pic, void *dst, size_t cap);
/* the inverse view: *pic = *desc with its array pointers set into `packed` (nothing is copied) */
int  p264hip_unpack_input(const p264hip_picture_t *desc, const void *packed, size_t bytes, p264hip_picture_t *pic);
/* packed host buffer -> slot, one asynchronous copy (the buffer stays untouched until a marker / p264hip_sync; the records are
 * trusted to have been checked by p264hip_pack_input) */
int  p264hip_upload_packed(p264hip_ctx *ctx, int slot, const p264hip_picture_t *desc, const void *packed, size_t bytes);
/* ---- the compact LINK format (round 6).  A picture's arrays as they travel where the link is the bound (a packed upload over
 * PCIe, a scatter over xGMI / TCP): 1.42 MB per 1080p P picture of the bench stream in the slot layout, ~0.5 MB compact -
 *   records and reference indices verbatim (16 + 4 bytes per macroblock and list);
 *   vectors by shape, 2 bits per macroblock and list: 0 sixteen zero vectors (intra macroblocks, an unused list, a block at
 *   rest), 1 one vector (16x16 / skip), 2 one per 8x8 quadrant, 3 all sixteen;
 *   Intra4x4 modes only for the macroblocks whose sixteen modes are not all 2 (DC: what a parser leaves everywhere else), one
 *   flag bit per macroblock;
 *   coded levels as sixteen int8 per block where every level of the block fits (one flag bit per block), else sixteen int16;
 *   B pictures: the same for list 1, and the 512 bytes of bipred_weight[].
 * Nothing is lost: p264hip_expand_compact (host; the reference of the device kernel) gives back the slot layout byte for byte
 * (p264hip_pack_input's block, up to the padding between its sections).
 * p264hip_upload_compact copies the block into a staging area of the slot (one asynchronous copy, on a side stream of the context)
 * and queues its expansion: ONE kernel expands every picture uploaded since the last one, in front of the next
 * p264hip_reconstruct (or clone / sync), which also is where the context's stream starts to wait for the copies - the caller's
 * block must stay untouched until a marker taken AFTER that call has been reached (or p264hip_sync has returned).
 * Like p264hip_upload_packed the call trusts the block to come from the packer (which checks the records as p264hip_upload
 * does): it checks the header (O(1): p264hip_compact_header_ok - sections inside the block, in order, large enough for the
 * header's counts) and the device clamps every place it derives from the block's bits to its section - an inconsistent block
 * gives a wrong picture, never an access outside the block or the slot.  p264hip_compact_check is the full check (counts implied
 * by the shape and flag bits, records, weights) for blocks from anywhere else. */
#define P264HIP_COMPACT_MAGIC 0x43343632u           /* "264C" */
#define P264HIP_COMPACT_MAX_MB 8192                 /* macroblocks per picture the device expansion takes (its offsets live in LDS) */
typedef struct p264hip_compact_hdr {                /* 128 bytes; sections follow at 16-byte aligned offsets from the block's start */
    uint32_t magic, n_mb, n_coef_blocks, bytes;     /* bytes: of the whole block */
    uint32_t n_lists;                               /* 1; 2 for a B picture */
    uint32_t off_rec, off_i4flag, off_i4, off_lvflag, off_levels, off_weights;      /* off_weights: 0 unless n_lists == 2 */
    uint32_t n_i4, level_bytes;                     /* macroblocks with an Intra4x4 mode entry, bytes of levels */
    struct { uint32_t off_ref, off_shape, off_vec, n_vec; } list[2];               /* n_vec: vectors (dwords) in the list's vec section */
    uint32_t reserved[11];
} p264hip_compact_hdr_t;
/* bytes a compact block of this picture needs at most */
size_t  p264hip_compact_bound(const p264hip_picture_t *pic);
int64_t p264hip_pack_compact(const p264hip_picture_t *pic, void *dst, size_t cap);          /* bytes used, or a negative P264HIP_E* code */
/* host: compact block -> the slot layout (p264hip_input_layout of desc), `out` of at least layout.bytes */
int     p264hip_expand_compact(const p264hip_picture_t *desc, const void *compact, size_t bytes, void *out, size_t cap);
int     p264hip_compact_header_ok(const p264hip_picture_t *desc, const void *compact, size_t bytes);   /* 1 / 0; what upload_compact checks */
int     p264hip_compact_check(const p264hip_picture_t *desc, const void *compact, size_t bytes);       /* 0 or P264HIP_EINVAL; the full check, no device */
int     p264hip_upload_compact(p264hip_ctx *ctx, int slot, const p264hip_picture_t *desc, const void *compact, size_t bytes);

/* device producers: reserve makes room in `slot` for a picture described by desc (scalar fields; its pointers are ignored)
 * and returns where its packed arrays are to be written (it waits for the context's stream only if work that still uses the
 * slot's previous picture is in flight: one wait covers a whole round of reserves); the slot becomes usable with commit, which
 * the caller issues once its writes have completed (the context's stream does not wait for anybody else's).  commit queues
 * the record check p264hip_upload runs on the host (every macroblock's packed blocks inside coefs[]) as a kernel over the
 * block; p264hip_reconstruct reads the verdicts of a batch's committed pictures with one wait and fails with P264HIP_EINVAL
 * where a block is inconsistent - the producer is not trusted. */
int  p264hip_input_reserve(p264hip_ctx *ctx, int slot, const p264hip_picture_t *desc, void **dev, size_t *bytes);
int  p264hip_input_commit(p264hip_ctx *ctx, int slot);
/* frame `slot` of `stream` converted to planar I420 (Y, U, V back to back, MB-aligned) in device buffer number `index` of
 * the context (buffers are created on demand and live as long as the context); enqueued on the context's stream - *dev is
 * readable by other streams / RCCL after p264hip_sync or a marker. */
int  p264hip_frame_planar_device(p264hip_ctx *ctx, int stream, int slot, int index, void **dev, size_t *bytes);
/* plain synchronous copies between host and device memory (the fan-out's TCP transport uses them where it stands in for a
 * device-to-device transport in tests) */
int  p264hip_copy_to_device(void *dev, const void *host, size_t bytes);
int  p264hip_copy_from_device(void *host, const void *dev, size_t bytes);

/* Reconstruct a batch: picture input slot pic_ids[i] is decoded into stream streams[i].
 * All pictures of one call are mutually independent (different streams).  Asynchronous. */
int  p264hip_reconstruct(p264hip_ctx *ctx, const int *pic_ids, const int *streams, int n);

/* upload + reconstruct of one picture for one stream (the p264_decoder_decode path) */
int  p264hip_submit(p264hip_ctx *ctx, int stream, const p264hip_picture_t *pic);

int  p264hip_sync(p264hip_ctx *ctx);

/* Frame-store access (synchronous).  Strides in bytes; planes are mb_w*16 x mb_h*16 (luma). */
int  p264hip_read_frame(p264hip_ctx *ctx, int stream, int slot,
                        uint8_t *y, int y_stride, uint8_t *u, uint8_t *v, int c_stride);
int  p264hip_write_frame(p264hip_ctx *ctx, int stream, int slot,
                         const uint8_t *y, int y_stride, const uint8_t *u, const uint8_t *v, int c_stride);
/* The asynchronous pair for callers that keep their buffers pinned (p264hip_host_alloc): nothing waits until
 * p264hip_sync / a marker.  submit_async = upload_async + reconstruct; read_frame_async enqueues the layout conversion and
 * the three plane copies behind whatever was submitted before.  (The drop-in path uses them: one wait per picture.) */
int  p264hip_submit_async(p264hip_ctx *ctx, int stream, const p264hip_picture_t *pic);
int  p264hip_read_frame_async(p264hip_ctx *ctx, int stream, int slot,
                              uint8_t *y, int y_stride, uint8_t *u, uint8_t *v, int c_stride);

/* Timing hooks used by bench.py: HIP events on the context's own stream.
 * kernel index: 0 inter (MC + residual), 1 intra, 2 deblock, 3 whole reconstruct call. */
#define P264HIP_NKERNELS 4
int  p264hip_timing_enable(p264hip_ctx *ctx, int on);
int  p264hip_timing_read(p264hip_ctx *ctx, double ms_sum[P264HIP_NKERNELS], int64_t count[P264HIP_NKERNELS]);
int  p264hip_timing_reset(p264hip_ctx *ctx);

/* What the last p264hip_reconstruct call launched (the workgroup shapes follow from the batch size and the device's compute
 * units; tests assert that a batch of the bench's size selects the bench's shapes, bench.py records them).  Diagnostics only. */
typedef struct p264hip_launch_info {
    int32_t pictures;              /* pictures of the batch */
    int32_t compute_units;         /* of the context's device */
    int32_t mc_wgs_per_picture;    /* k_mc: workgroups per picture (0: no inter launch) */
    int32_t intra_waves;           /* k_intra / k_intra_sparse: wavefronts per workgroup */
    int32_t edge_info_fused;       /* edge-info workgroups per picture inside the k_intra_sparse launch (0: own launch k_deblock_bs) */
    int32_t deblock_pics_per_wg, deblock_rb_log2, deblock_waves, deblock_wgs;
    int32_t deblock_odd_single;    /* 1: odd pictures per workgroup - pairs in bands of 4 rows, the last picture alone in bands of 8 */
    int32_t reserved[6];
} p264hip_launch_info_t;
int  p264hip_last_launch(p264hip_ctx *ctx, p264hip_launch_info_t *out);

/* Properties of the library build.  Bit 0 (P264HIP_BUILD_TIMING): compiled with -DP264AMD_TIMING_BUILD, which unlocks the
 * EXPM_* / EXPD_* switches of the kernel headers - pieces of the kernels compiled out to time the rest.  Such a build
 * produces wrong pictures; p264hip_create refuses to run in it unless P264AMD_TIMING_BUILD_OK=1 is set in the environment
 * (scratch/variants_run.sh does), bench.py records the flag and the test suite asserts it is clear. */
#define P264HIP_BUILD_TIMING 1
int  p264hip_build_info(void);

#ifdef __cplusplus
}
#endif
#endif
