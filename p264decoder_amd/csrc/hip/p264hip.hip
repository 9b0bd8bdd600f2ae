// p264hip.hip - host side of the C ABI in include/p264hip.h: device context, frame stores,
// resident picture inputs, batch launch of the reconstruction kernels, timing hooks.
//
// One context = one GPU = one HIP stream.  The product has no CPU reconstruction path: when no
// device is usable every entry point fails with P264HIP_ENODEV.
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <vector>
#include <mutex>
#include <unordered_map>
#include "p264hip.h"
#include "device_common.h"
#define P264HIP_K_DEBLOCK_DECL_ONLY          // k_deblock lives in k_deblock.hip (its own compiler options: kernel_deblock.h)
#include "kernel_deblock.h"
#include "kernel_mc.h"
#include "kernel_intra.h"
#include "kernel_expand.h"

static thread_local char g_err[512] = "";
static int fail(int code, const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
    return code;
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(P264HIP_EHIP, "%s: %s", #x, hipGetErrorString(e_)); } while (0)

extern "C" const char *p264hip_last_error(void) { return g_err; }

extern "C" int p264hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int p264hip_build_info(void)
{
#ifdef P264AMD_TIMING_BUILD
    return P264HIP_BUILD_TIMING;
#else
    return 0;
#endif
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct PicSlot {                       // one device-resident parsed picture
    uint8_t *dev = nullptr;
    size_t   cap = 0;                  // bytes allocated
    size_t   off_mv = 0, off_ref = 0, off_i4 = 0, off_coef = 0, off_mv_l1 = 0, off_ref_l1 = 0, off_weights = 0;   // p264hip_input_layout_t (the last three: B pictures)
    size_t   bytes = 0;                // bytes in use
    p264hip_picture_t meta;            // scalar fields only; pointers unused
    bool     valid = false, reserved = false;   // reserved: p264hip_input_reserve handed the block out, commit is pending
    uint8_t *stage = nullptr; size_t stage_cap = 0;   // p264hip_upload_compact: the compact block as it arrived; pending: its expansion into dev has not been launched yet
    bool     pending = false;
    int      stage_cs = 0;                            // the side stream that carried the pending block (a second block for the same slot follows on the same one)
    bool     unchecked = false;                  // committed by a device producer: the record check (k_check_records) has been queued, its verdict not yet read
    uint64_t last_use = 0;                       // epoch of the last work queued on the context's stream that reads or writes the block
};

#define BATCH_RING 4
#define COPY_STREAMS 4

// device frame layout (strips, device_common.h) <-> planar staging (host boundary only): one thread per dword of the frame
__global__ void k_tile_convert(uint8_t *frame, uint8_t *planar, Geom g, int to_planar)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.n_mb * 96) return;
    const int mb = i / 96, d = i - mb * 96;
    const int mx = mb % g.mb_w, my = mb / g.mb_w;
    size_t po, fo;
    if (d < 64) {
        po = (size_t)(my * 16 + (d >> 2)) * g.w + mx * 16 + (d & 3) * 4;
        fo = mb_luma_off(g, mx, my) + d * 4;
    } else {
        const int e = d - 64, p = e >> 4, r = (e >> 1) & 7, dd = e & 1;
        po = (size_t)g.w * g.h + (size_t)p * g.cw * g.ch + (size_t)(my * 8 + r) * g.cw + mx * 8 + dd * 4;
        fo = mb_chroma_off(g, mx, my) + r * 16 + p * 8 + dd * 4;
    }
    uint32_t *t = (uint32_t *)(frame + fo), *q = (uint32_t *)(planar + po);
    if (to_planar) *q = *t; else *t = *q;
}

// The check p264hip_upload runs on the host, for pictures that arrive in device memory (p264hip_input_reserve / _commit): every
// macroblock's packed blocks must lie inside coefs[] - the kernels index the coefficient stream without further checks
// (include/p264hip.h).  One flag per input slot.
__global__ void k_check_records(const p264hip_mb_t *mb, int n_mb, uint32_t n_coef_blocks, int *bad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_mb) return;
    const uint4 r = gload4(mb + i);
    if (r.y && (uint64_t)r.z + (uint64_t)__popc(r.y & 0x3ffffffu) > (uint64_t)n_coef_blocks) atomicOr(bad, 1);
}

struct p264hip_ctx {
    int device = 0, n_cu = 256;
    hipStream_t stream = nullptr;
    Geom g;
    int n_streams = 0, slots = 0, max_pictures = 0;
    uint8_t *frames = nullptr;
    size_t frame_bytes = 0;
    std::vector<PicSlot> pics;
    PicDev *h_batch[BATCH_RING] = {}, *d_batch[BATCH_RING] = {};
    hipEvent_t batch_free[BATCH_RING] = {};
    int batch_cap = 0, ring = 0;
    int *d_status = nullptr;
    int *d_slot_bad = nullptr;             // [max_pictures]: k_check_records' verdict per input slot
    std::vector<int> pending;              // input slots whose compact block waits for k_expand_compact (launched in front of the next reconstruct / clone / sync)
    ExpandJob *h_jobs = nullptr, *d_jobs = nullptr; int jobs_cap = 0; hipEvent_t jobs_free = nullptr;
    // compact blocks travel on COPY_STREAMS side streams, round robin: a copy of ~0.5 MB costs ~18 us of fixed latency beside ~9 us of
    // transfer (measured, round 6: 512 copies per step on the context's one stream ran at 19.6 GB/s) - side by side the latencies overlap.
    // The expansion kernel (context's stream) waits for the side streams' copies; a side stream waits for the last expansion before it
    // overwrites a staging area.
    hipStream_t cstream[COPY_STREAMS] = {}; hipEvent_t cdone[COPY_STREAMS] = {}; bool cdirty[COPY_STREAMS] = {}, cwaited[COPY_STREAMS] = {};
    hipEvent_t expand_done = nullptr; int next_cs = 0;
    uint64_t upload_copies = 0;            // host -> HBM copies queued by p264hip_upload / _upload_async (one per picture whose arrays lie like a slot)
    uint64_t epoch = 0, done_epoch = 0;    // work queued on the stream / known to have completed (a slot is free for a new producer once its last_use is done)
    EdgeInfo *d_edge = nullptr;            // [batch_cap][n_mb], scratch between k_deblock_bs and k_deblock
    uint32_t *d_mc = nullptr;              // [batch_cap][ml.words], motion-compensation work lists (k_mc_sort -> k_mc, k_mc_second)
    uint8_t *d_is_intra = nullptr;         // [batch_cap][n_mb], 1 = intra macroblock (k_mc_sort -> k_intra's collect pass; P / B pictures)
    McLayout ml;
    std::vector<int> stream_seen;          // p264hip_reconstruct: batch index + 1 that last named a stream in the current call
    uint8_t *d_planar = nullptr;           // planar staging for p264hip_read_frame / p264hip_write_frame
    std::vector<uint8_t *> planar_pool;    // p264hip_frame_planar_device: planar I420 frames that stay on the device
    // tuning knobs, read from the environment ONCE (p264hip_create); 0 = built-in choice
    int tune_mc_wgs = 0, tune_intra_waves = 0, tune_rb_log2 = 0, tune_pics_per_wg = 0, tune_db_waves = 0, tune_bs_fused = -1, tune_odd_single = -1;
    p264hip_launch_info_t last = {};       // what the last p264hip_reconstruct launched
    hipEvent_t markers[P264HIP_MARKERS] = {};
    int next_marker = 0;
    bool timing = false;
    struct Stamp { hipEvent_t a, b; int k; };
    std::vector<Stamp> stamps;
    std::vector<hipEvent_t> event_pool;
    double ms_sum[P264HIP_NKERNELS] = {};
    int64_t ms_cnt[P264HIP_NKERNELS] = {};
};

static uint8_t *frame_ptr(p264hip_ctx *c, int stream, int slot)
{
    return c->frames + ((size_t)stream * c->slots + slot) * c->frame_bytes;
}

extern "C" int p264hip_create(p264hip_ctx **out, int device, int mb_w, int mb_h, int n_streams, int slots, int max_pictures)
{
    if (!out || mb_w < 1 || mb_w > 2047 || mb_h < 1 || mb_h > MAX_MB_ROWS ||    /* (11 bits of macroblock column in a work-list entry) */
        n_streams < 1 || slots < 1 || slots > P264HIP_MAX_REFS + 1 || max_pictures < 1)
        return fail(P264HIP_EINVAL, "p264hip_create: bad argument (mb %dx%d, streams %d, slots %d, pictures %d)", mb_w, mb_h, n_streams, slots, max_pictures);
#ifdef P264AMD_TIMING_BUILD
    {   // a build with pieces of the kernels compiled out: its pictures are wrong, it only runs for whoever asks for exactly that
        const char *ok = getenv("P264AMD_TIMING_BUILD_OK");
        if (!ok || strcmp(ok, "1") != 0)
            return fail(P264HIP_EINVAL, "this library is a timing build (-DP264AMD_TIMING_BUILD: kernels with pieces compiled out, wrong pictures); "
                                        "set P264AMD_TIMING_BUILD_OK=1 to run it anyway");
    }
#endif
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(P264HIP_ENODEV, "no HIP device available: the MI355X reconstruction path cannot run (there is no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(P264HIP_EINVAL, "device %d out of range (have %d)", device, ndev);
    HIPCHK(hipSetDevice(device));
    p264hip_ctx *c = new p264hip_ctx();
    c->device = device;
    c->n_streams = n_streams; c->slots = slots; c->max_pictures = max_pictures;
    Geom &g = c->g;
    g.mb_w = mb_w; g.mb_h = mb_h; g.n_mb = mb_w * mb_h;
    g.w = mb_w * 16; g.h = mb_h * 16; g.cw = g.w / 2; g.ch = g.h / 2;
    g.ystrip = (uint32_t)g.h * 16u; g.cstrip = (uint32_t)g.ch * 16u; g.coff = (uint32_t)g.n_mb * MB_LUMA_BYTES;
    c->frame_bytes = align_up((size_t)g.n_mb * (MB_LUMA_BYTES + MB_CHROMA_BYTES), 256);      // strip layout (device_common.h)
    // (one stream's store is addressed by 32-bit offsets through one buffer descriptor, and the kernels use MC_OOB as "an offset
    // beyond any store" for loads that must return zeros - kernel_mc.h: the store has to end at or below it)
    if (c->frame_bytes * (size_t)slots > (size_t)MC_OOB) { delete c; return fail(P264HIP_EINVAL, "frame store of one stream exceeds %u bytes (%d slots of %zu bytes)", MC_OOB, slots, c->frame_bytes); }
    {   // locality band of the motion-compensation lists: 16 macroblock rows unless that makes more than MC_MAX_BANDS bands
        int band_log2 = 4;
        if (const char *e = getenv("P264AMD_MC_BAND_LOG2")) { int v = atoi(e); if (v >= 0 && v <= 9) band_log2 = v; }
        while (((mb_h + (1 << band_log2) - 1) >> band_log2) > MC_MAX_BANDS) band_log2++;
        c->ml = mc_layout(mb_w, mb_h, band_log2);
    }
    c->stream_seen.assign((size_t)n_streams, 0);
    c->pics.resize((size_t)max_pictures);
    { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && v > 0) c->n_cu = v; }
    c->last.compute_units = c->n_cu;                        // (p264hip_last_launch before the first batch: the device's size)
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (const char *env = getenv("P264AMD_MC_WGS_PER_PIC")) c->tune_mc_wgs = atoi(env);
    if (const char *env = getenv("P264AMD_INTRA_WAVES")) c->tune_intra_waves = atoi(env);
    if (const char *env = getenv("P264AMD_DEBLOCK_RB_LOG2")) c->tune_rb_log2 = atoi(env);
    if (const char *env = getenv("P264AMD_DEBLOCK_ODD_SINGLE")) c->tune_odd_single = atoi(env) != 0;   /* 0: never, 1: whenever the shape allows it */
    if (const char *env = getenv("P264AMD_DEBLOCK_PICS_PER_WG")) c->tune_pics_per_wg = atoi(env);
    if (const char *env = getenv("P264AMD_DEBLOCK_WAVES")) c->tune_db_waves = atoi(env);
    if (const char *env = getenv("P264AMD_BS_FUSED")) { c->tune_bs_fused = atoi(env); if (c->tune_bs_fused > 16) c->tune_bs_fused = 16; }   // 0: own launch; n: n edge-info workgroups per picture in the k_intra_sparse launch
    if (e == hipSuccess) e = hipMalloc((void **)&c->frames, c->frame_bytes * (size_t)n_streams * slots);
    if (e == hipSuccess) e = hipMemsetAsync(c->frames, 0, c->frame_bytes * (size_t)n_streams * slots, c->stream);
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_status, sizeof(int));
    if (e == hipSuccess) e = hipMemsetAsync(c->d_status, 0, sizeof(int), c->stream);
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_slot_bad, sizeof(int) * (size_t)max_pictures);
    if (e == hipSuccess) e = hipMemsetAsync(c->d_slot_bad, 0, sizeof(int) * (size_t)max_pictures, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        int rc = fail(e == hipErrorOutOfMemory ? P264HIP_ENOMEM : P264HIP_EHIP, "p264hip_create: %s", hipGetErrorString(e));
        p264hip_destroy(c);
        return rc;
    }
    *out = c;
    return P264HIP_OK;
}

extern "C" void p264hip_destroy(p264hip_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto &p : c->pics) { if (p.dev) (void)hipFree(p.dev); if (p.stage) (void)hipFree(p.stage); }
    if (c->h_jobs) (void)hipHostFree(c->h_jobs);
    if (c->d_jobs) (void)hipFree(c->d_jobs);
    if (c->jobs_free) (void)hipEventDestroy(c->jobs_free);
    if (c->expand_done) (void)hipEventDestroy(c->expand_done);
    for (int i = 0; i < COPY_STREAMS; i++) { if (c->cstream[i]) { (void)hipStreamSynchronize(c->cstream[i]); (void)hipStreamDestroy(c->cstream[i]); } if (c->cdone[i]) (void)hipEventDestroy(c->cdone[i]); }
    for (int i = 0; i < BATCH_RING; i++) {
        if (c->h_batch[i]) (void)hipHostFree(c->h_batch[i]);
        if (c->d_batch[i]) (void)hipFree(c->d_batch[i]);
        if (c->batch_free[i]) (void)hipEventDestroy(c->batch_free[i]);
    }
    for (auto &s : c->stamps) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    if (c->frames) (void)hipFree(c->frames);
    if (c->d_edge) (void)hipFree(c->d_edge);
    if (c->d_mc) (void)hipFree(c->d_mc);
    if (c->d_is_intra) (void)hipFree(c->d_is_intra);
    if (c->d_planar) (void)hipFree(c->d_planar);
    for (uint8_t *p : c->planar_pool) if (p) (void)hipFree(p);
    for (auto &m : c->markers) if (m) (void)hipEventDestroy(m);
    if (c->d_status) (void)hipFree(c->d_status);
    if (c->d_slot_bad) (void)hipFree(c->d_slot_bad);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// arrays: the picture's host arrays are there to be checked too (false: only the descriptor - the arrays are already packed)
static int check_pic(p264hip_ctx *c, const p264hip_picture_t *p, bool arrays)
{
    if (p->mb_w != c->g.mb_w || p->mb_h != c->g.mb_h)
        return fail(P264HIP_EINVAL, "picture is %dx%d MBs, context is %dx%d", p->mb_w, p->mb_h, c->g.mb_w, c->g.mb_h);
    if (p->dst_slot < 0 || p->dst_slot >= c->slots) return fail(P264HIP_EINVAL, "dst_slot %d out of range", p->dst_slot);
    if (p->n_ref < 0 || p->n_ref > P264HIP_MAX_REFS) return fail(P264HIP_EINVAL, "n_ref %d out of range", p->n_ref);
    for (int i = 0; i < p->n_ref; i++)
        if (p->ref_slot[i] < 0 || p->ref_slot[i] >= c->slots) return fail(P264HIP_EINVAL, "ref_slot[%d]=%d out of range", i, p->ref_slot[i]);
    if (p->slice_type != P264_SLICE_P && p->slice_type != P264_SLICE_B && p->slice_type != P264_SLICE_I) return fail(P264HIP_EINVAL, "slice_type %d", p->slice_type);
    if (p->slice_type == P264_SLICE_P && p->n_ref < 1) return fail(P264HIP_EINVAL, "P picture without reference");
    if (p->slice_type == P264_SLICE_B) {
        if (p->n_ref < 1 || p->n_ref_l1 < 1 || p->n_ref_l1 > P264HIP_MAX_REFS) return fail(P264HIP_EINVAL, "B picture: list lengths %d / %d", p->n_ref, p->n_ref_l1);
        if (arrays && (!p->mv_l1 || !p->ref_idx_l1)) return fail(P264HIP_EINVAL, "B picture without list-1 arrays");
        if (p->n_coef_blocks >= (1u << MCE_W_SHIFT)) return fail(P264HIP_EINVAL, "B picture with %u coefficient blocks (limit %u: the second pass packs the weight beside the index)", p->n_coef_blocks, 1u << MCE_W_SHIFT);
        for (int i = 0; i < p->n_ref_l1; i++)
            if (p->ref_slot_l1[i] < 0 || p->ref_slot_l1[i] >= c->slots) return fail(P264HIP_EINVAL, "ref_slot_l1[%d]=%d out of range", i, p->ref_slot_l1[i]);
        if (p->weighted_bipred)
            for (int i = 0; i < P264HIP_MAX_REFS * P264HIP_MAX_REFS; i++)
                if (p->bipred_weight[i] < -64 || p->bipred_weight[i] > 128) return fail(P264HIP_EINVAL, "bipred_weight[%d]=%d out of range (-64 .. 128)", i, p->bipred_weight[i]);
    }
    if (!arrays) return 0;
    if (!p->mb || !p->mv || !p->ref_idx || !p->i4modes || (p->n_coef_blocks && !p->coefs)) return fail(P264HIP_EINVAL, "null picture array");
    // every macroblock's packed blocks must lie inside coefs[] (the kernels index it without further checks)
    const int n_mb = c->g.n_mb;
    for (int i = 0; i < n_mb; i++) {
        const p264hip_mb_t &m = p->mb[i];
        if (m.coef_mask && (uint64_t)m.coef_index + (uint64_t)__builtin_popcount(m.coef_mask & 0x3ffffffu) > p->n_coef_blocks)
            return fail(P264HIP_EINVAL, "macroblock %d: coefficient blocks [%u, +%d) outside coefs[%u]", i, m.coef_index,
                        __builtin_popcount(m.coef_mask & 0x3ffffffu), p->n_coef_blocks);
    }
    return 0;
}

// room for a picture of this layout in slot `id`; the slot's offsets follow the layout (include/p264hip.h)
static int slot_prepare(p264hip_ctx *c, PicSlot &s, const p264hip_input_layout_t &L)
{
    if (L.bytes > s.cap) {
        if (s.dev) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(s.dev)); s.dev = nullptr; s.cap = 0; }
        size_t cap = L.bytes + L.bytes / 4;
        hipError_t e = hipMalloc((void **)&s.dev, cap);
        if (e != hipSuccess) return fail(P264HIP_ENOMEM, "hipMalloc(%zu) for picture input: %s", cap, hipGetErrorString(e));
        s.cap = cap;
    }
    s.off_mv = L.off_mv; s.off_ref = L.off_ref; s.off_i4 = L.off_i4; s.off_coef = L.off_coef;
    s.off_mv_l1 = L.off_mv_l1; s.off_ref_l1 = L.off_ref_l1; s.off_weights = L.off_weights; s.bytes = L.bytes;
    return 0;
}
static void slot_meta(PicSlot &s, const p264hip_picture_t *p)
{
    s.meta = *p;
    s.meta.mb = nullptr; s.meta.mv = nullptr; s.meta.ref_idx = nullptr; s.meta.i4modes = nullptr; s.meta.coefs = nullptr; s.meta.mv_l1 = nullptr; s.meta.ref_idx_l1 = nullptr;
}

// ---- compact link format (include/p264hip.h, kernel_expand.h) ----
// a slot that gets new content by another road no longer waits for its compact block's expansion
static void unpend(p264hip_ctx *c, int id)
{
    PicSlot &s = c->pics[(size_t)id];
    if (!s.pending) return;
    s.pending = false;
    for (size_t i = 0; i < c->pending.size(); i++) if (c->pending[i] == id) { c->pending.erase(c->pending.begin() + (long)i); break; }
}
// ONE launch expands every compact block uploaded since the last one
static int expand_pending(p264hip_ctx *c)
{
    const int n = (int)c->pending.size();
    if (!n) return 0;
    if (n > c->jobs_cap) {
        if (c->jobs_free) HIPCHK(hipEventSynchronize(c->jobs_free));
        if (c->h_jobs) (void)hipHostFree(c->h_jobs);
        if (c->d_jobs) (void)hipFree(c->d_jobs);
        c->h_jobs = nullptr; c->d_jobs = nullptr;
        const int cap = n + n / 2 + 16;
        HIPCHK(hipHostMalloc((void **)&c->h_jobs, (size_t)cap * sizeof(ExpandJob), hipHostMallocDefault));
        HIPCHK(hipMalloc((void **)&c->d_jobs, (size_t)cap * sizeof(ExpandJob)));
        if (!c->jobs_free) HIPCHK(hipEventCreateWithFlags(&c->jobs_free, hipEventDisableTiming));
        c->jobs_cap = cap;
    } else HIPCHK(hipEventSynchronize(c->jobs_free));        // the copy that last read h_jobs is done
    for (int i = 0; i < n; i++) {
        PicSlot &s = c->pics[(size_t)c->pending[(size_t)i]];
        c->h_jobs[i] = ExpandJob{ s.stage, s.dev, (uint32_t)s.off_mv, (uint32_t)s.off_ref, (uint32_t)s.off_i4, (uint32_t)s.off_coef,
                                  (uint32_t)s.off_mv_l1, (uint32_t)s.off_ref_l1, (uint32_t)s.off_weights, 0u };
        s.pending = false;
    }
    c->pending.clear();
    for (int i = 0; i < COPY_STREAMS; i++) {                 // the blocks' copies (side streams) in front of the kernel that reads them
        if (!c->cdirty[i]) continue;
        HIPCHK(hipEventRecord(c->cdone[i], c->cstream[i]));
        HIPCHK(hipStreamWaitEvent(c->stream, c->cdone[i], 0));
        c->cdirty[i] = false;
    }
    HIPCHK(hipMemcpyAsync(c->d_jobs, c->h_jobs, (size_t)n * sizeof(ExpandJob), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipEventRecord(c->jobs_free, c->stream));
    hipLaunchKernelGGL(k_expand_compact, dim3((unsigned)n), dim3(EXPAND_THREADS), 0, c->stream, (const ExpandJob *)c->d_jobs);
    HIPCHK(hipGetLastError());
    if (!c->expand_done) HIPCHK(hipEventCreateWithFlags(&c->expand_done, hipEventDisableTiming));
    HIPCHK(hipEventRecord(c->expand_done, c->stream));        // the staging areas may be overwritten behind this
    for (int i = 0; i < COPY_STREAMS; i++) c->cwaited[i] = false;
    return 0;
}

static int upload_one(p264hip_ctx *c, int id, const p264hip_picture_t *p)
{
    int rc = check_pic(c, p, true);
    if (rc) return rc;
    unpend(c, id);
    PicSlot &s = c->pics[(size_t)id];
    const size_t n = (size_t)c->g.n_mb;
    p264hip_input_layout_t L;
    if (p264hip_input_layout(p, &L)) return fail(P264HIP_EINVAL, "picture layout");
    if ((rc = slot_prepare(c, s, L))) return rc;
    if (p->slice_type == P264_SLICE_B) {
        HIPCHK(hipMemcpyAsync(s.dev + L.off_mv_l1, p->mv_l1, n * 64, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(s.dev + L.off_ref_l1, p->ref_idx_l1, n * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(s.dev + L.off_weights, p->bipred_weight, sizeof p->bipred_weight, hipMemcpyHostToDevice, c->stream));
    }
    // The caller's arrays already lie in host memory the way the slot is laid out (the parser builds its pictures like that since
    // round 6, csrc/host/parser.c: picbuf_t): ONE copy for records, vectors, indices, modes and levels instead of five - the
    // pipeline's uploads ran at 12 GB/s in pieces, a packed block goes at 35 (bench.py: extras.upload_inclusive).
    const uint8_t *hb = (const uint8_t *)p->mb;
    const bool as_slot = (const uint8_t *)p->mv == hb + L.off_mv && (const uint8_t *)p->ref_idx == hb + L.off_ref && (const uint8_t *)p->i4modes == hb + L.off_i4
                         && (p->n_coef_blocks == 0 || (const uint8_t *)p->coefs == hb + L.off_coef);
    if (as_slot) HIPCHK(hipMemcpyAsync(s.dev, hb, L.off_coef + (size_t)p->n_coef_blocks * 32, hipMemcpyHostToDevice, c->stream));
    else {
        HIPCHK(hipMemcpyAsync(s.dev, p->mb, n * sizeof(p264hip_mb_t), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(s.dev + L.off_mv, p->mv, n * 64, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(s.dev + L.off_ref, p->ref_idx, n * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(s.dev + L.off_i4, p->i4modes, n * 16, hipMemcpyHostToDevice, c->stream));
        if (p->n_coef_blocks)
            HIPCHK(hipMemcpyAsync(s.dev + L.off_coef, p->coefs, (size_t)p->n_coef_blocks * 32, hipMemcpyHostToDevice, c->stream));
    }
    c->upload_copies += as_slot ? 1 : 5;
    slot_meta(s, p);
    s.valid = true; s.unchecked = false; s.last_use = ++c->epoch;
    return 0;
}

// ---- other ways into a slot (include/p264hip.h): a packed host block in one copy; a device producer (reserve / commit) ----
extern "C" int p264hip_upload_packed(p264hip_ctx *c, int slot, const p264hip_picture_t *desc, const void *packed, size_t bytes)
{
    if (!c || !desc || !packed || slot < 0 || slot >= c->max_pictures) return fail(P264HIP_EINVAL, "p264hip_upload_packed: bad argument (slot %d)", slot);
    HIPCHK(hipSetDevice(c->device));
    int rc = check_pic(c, desc, false);
    if (rc) return rc;
    p264hip_input_layout_t L;
    if (p264hip_input_layout(desc, &L) || bytes != L.bytes) return fail(P264HIP_EINVAL, "p264hip_upload_packed: %zu bytes, the layout has %zu", bytes, L.bytes);
    unpend(c, slot);
    PicSlot &s = c->pics[(size_t)slot];
    if ((rc = slot_prepare(c, s, L))) return rc;
    HIPCHK(hipMemcpyAsync(s.dev, packed, L.bytes, hipMemcpyHostToDevice, c->stream));
    slot_meta(s, desc);
    s.valid = true; s.unchecked = false; s.last_use = ++c->epoch;
    return P264HIP_OK;
}

extern "C" int p264hip_upload_compact(p264hip_ctx *c, int slot, const p264hip_picture_t *desc, const void *compact, size_t bytes)
{
    if (!c || !desc || !compact || slot < 0 || slot >= c->max_pictures) return fail(P264HIP_EINVAL, "p264hip_upload_compact: bad argument (slot %d)", slot);
    HIPCHK(hipSetDevice(c->device));
    int rc = check_pic(c, desc, false);
    if (rc) return rc;
    if (!p264hip_compact_header_ok(desc, compact, bytes)) return fail(P264HIP_EINVAL, "p264hip_upload_compact: the block is not a consistent compact picture of %dx%d macroblocks with %u coefficient blocks", desc->mb_w, desc->mb_h, desc->n_coef_blocks);
    p264hip_input_layout_t L;
    if (p264hip_input_layout(desc, &L)) return fail(P264HIP_EINVAL, "picture layout");
    PicSlot &s = c->pics[(size_t)slot];
    if ((rc = slot_prepare(c, s, L))) return rc;
    if (bytes > s.stage_cap) {
        if (s.stage) { HIPCHK(hipDeviceSynchronize()); HIPCHK(hipFree(s.stage)); s.stage = nullptr; s.stage_cap = 0; }
        const size_t cap = bytes + bytes / 4;
        hipError_t e = hipMalloc((void **)&s.stage, cap);
        if (e != hipSuccess) return fail(P264HIP_ENOMEM, "hipMalloc(%zu) for a compact picture: %s", cap, hipGetErrorString(e));
        s.stage_cap = cap;
    }
    // (a slot whose block is still waiting for its expansion gets the newer block on the SAME side stream: two copies into one
    // staging area on two streams would land in any order)
    int cs = s.stage_cs;
    if (!s.pending) { cs = c->next_cs; c->next_cs = (c->next_cs + 1) % COPY_STREAMS; s.stage_cs = cs; }
    if (!c->cstream[cs]) {
        HIPCHK(hipStreamCreateWithFlags(&c->cstream[cs], hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&c->cdone[cs], hipEventDisableTiming));
    }
    if (!c->cwaited[cs]) {
        // the last expansion kernel (it read the staging areas) is in front of this side stream's copies; nothing else is - a
        // copy into a staging area never touches the slot's arrays, so step t + 1's blocks travel while step t is reconstructed
        if (c->expand_done) HIPCHK(hipStreamWaitEvent(c->cstream[cs], c->expand_done, 0));
        c->cwaited[cs] = true;
    }
    HIPCHK(hipMemcpyAsync(s.stage, compact, bytes, hipMemcpyHostToDevice, c->cstream[cs]));
    c->cdirty[cs] = true;
    c->upload_copies += 1;
    slot_meta(s, desc);
    if (!s.pending) { s.pending = true; c->pending.push_back(slot); }
    s.valid = true; s.unchecked = false; s.reserved = false; s.last_use = ++c->epoch;
    return P264HIP_OK;
}

extern "C" int p264hip_input_reserve(p264hip_ctx *c, int slot, const p264hip_picture_t *desc, void **dev, size_t *bytes)
{
    if (!c || !desc || !dev || !bytes || slot < 0 || slot >= c->max_pictures) return fail(P264HIP_EINVAL, "p264hip_input_reserve: bad argument (slot %d)", slot);
    HIPCHK(hipSetDevice(c->device));
    int rc = check_pic(c, desc, false);
    if (rc) return rc;
    p264hip_input_layout_t L;
    if (p264hip_input_layout(desc, &L)) return fail(P264HIP_EINVAL, "picture layout");
    unpend(c, slot);
    PicSlot &s = c->pics[(size_t)slot];
    s.valid = false;
    // Whatever still reads the slot's previous picture on the context's stream must be through before somebody else writes
    // it.  One wait covers everything queued so far: the first reserve of a round waits (if the last batch is still running),
    // the others find their slots' work already known to be done - not one hipStreamSynchronize per picture.
    if (s.last_use > c->done_epoch) { const uint64_t upto = c->epoch; HIPCHK(hipStreamSynchronize(c->stream)); c->done_epoch = upto; }
    if ((rc = slot_prepare(c, s, L))) return rc;
    slot_meta(s, desc);
    s.reserved = true;
    *dev = s.dev; *bytes = L.bytes;
    return P264HIP_OK;
}

extern "C" int p264hip_input_commit(p264hip_ctx *c, int slot)
{
    if (!c || slot < 0 || slot >= c->max_pictures || !c->pics[(size_t)slot].reserved) return fail(P264HIP_EINVAL, "p264hip_input_commit: slot %d is not reserved", slot);
    PicSlot &s = c->pics[(size_t)slot];
    HIPCHK(hipSetDevice(c->device));
    // The producer wrote the arrays (it says they are complete on the device: include/p264hip.h); their records get the check
    // p264hip_upload runs on the host, on the device - a block from a peer that packed it wrongly must become an error, not
    // an out-of-bounds read.  The verdict is read once per batch, by p264hip_reconstruct.
    HIPCHK(hipMemsetAsync(c->d_slot_bad + slot, 0, sizeof(int), c->stream));
    const int n_mb = c->g.n_mb;
    hipLaunchKernelGGL(k_check_records, dim3((n_mb + 255) / 256), dim3(256), 0, c->stream, (const p264hip_mb_t *)s.dev, n_mb, s.meta.n_coef_blocks, c->d_slot_bad + slot);
    HIPCHK(hipGetLastError());
    s.reserved = false; s.valid = true; s.unchecked = true; s.last_use = ++c->epoch;
    return P264HIP_OK;
}

extern "C" int p264hip_frame_planar_device(p264hip_ctx *c, int stream, int slot, int index, void **dev, size_t *bytes)
{
    if (!c || !dev || !bytes || stream < 0 || stream >= c->n_streams || slot < 0 || slot >= c->slots || index < 0 || index >= 4096)
        return fail(P264HIP_EINVAL, "p264hip_frame_planar_device: bad argument (stream %d slot %d buffer %d)", stream, slot, index);
    HIPCHK(hipSetDevice(c->device));
    const Geom &g = c->g;
    const size_t sz = (size_t)g.w * g.h + 2 * (size_t)g.cw * g.ch;
    if ((size_t)index >= c->planar_pool.size()) c->planar_pool.resize((size_t)index + 1, nullptr);
    if (!c->planar_pool[(size_t)index]) {
        hipError_t e = hipMalloc((void **)&c->planar_pool[(size_t)index], sz);
        if (e != hipSuccess) { c->planar_pool[(size_t)index] = nullptr; return fail(P264HIP_ENOMEM, "hipMalloc(%zu) for a planar frame: %s", sz, hipGetErrorString(e)); }
    }
    const int n_dw = g.n_mb * 96;
    hipLaunchKernelGGL(k_tile_convert, dim3((n_dw + 255) / 256), dim3(256), 0, c->stream, frame_ptr(c, stream, slot), c->planar_pool[(size_t)index], g, 1);
    HIPCHK(hipGetLastError());
    *dev = c->planar_pool[(size_t)index]; *bytes = sz;
    return P264HIP_OK;
}

extern "C" int p264hip_copy_to_device(void *dev, const void *host, size_t bytes)
{
    if (!dev || !host) return fail(P264HIP_EINVAL, "p264hip_copy_to_device: null argument");
    HIPCHK(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice));
    return P264HIP_OK;
}
extern "C" int p264hip_copy_from_device(void *host, const void *dev, size_t bytes)
{
    if (!dev || !host) return fail(P264HIP_EINVAL, "p264hip_copy_from_device: null argument");
    HIPCHK(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
    return P264HIP_OK;
}

extern "C" int p264hip_upload(p264hip_ctx *c, int first, const p264hip_picture_t *pics, int n)
{
    if (!c || !pics || n < 0 || first < 0 || first + n > c->max_pictures) return fail(P264HIP_EINVAL, "p264hip_upload: bad range [%d,+%d)", first, n);
    HIPCHK(hipSetDevice(c->device));
    for (int i = 0; i < n; i++) { int rc = upload_one(c, first + i, &pics[i]); if (rc) return rc; }
    // sources are pageable host memory owned by the caller: make sure they are consumed before returning
    HIPCHK(hipStreamSynchronize(c->stream));
    return P264HIP_OK;
}

extern "C" int p264hip_upload_async(p264hip_ctx *c, int slot, const p264hip_picture_t *pic)
{
    if (!c || !pic || slot < 0 || slot >= c->max_pictures) return fail(P264HIP_EINVAL, "p264hip_upload_async: bad slot %d", slot);
    HIPCHK(hipSetDevice(c->device));
    return upload_one(c, slot, pic);
}

// Host buffers the device reads by DMA (the parsers write their pictures into them: p264parse_set_allocator).  Ordinary pages,
// advised as huge pages where the buffer is large, touched and then pinned with hipHostRegister - NOT hipHostMalloc: round 5
// found the parser threads 36 % slower on hipHostMalloc'ed buffers with or without a device at work (7.0 s against 5.15 s per 3 072
// 1080p pictures on 16 threads; registered 4 KB pages 5.7 s, registered huge pages 5.55 s, scratch/r5_pipe2.sh / r5_pipe3.sh) - the
// parser reads its own output back all the time (neighbour vectors, coefficient counts) through 4 KB translations.  The
// end-to-end pipeline went from 6.5 - 6.9 k to 8.3 k frames/s.  P264AMD_HOST_ALLOC = 0 (hipHostMalloc, coherent) / 1
// (non-coherent) / 2 (registered 4 KB pages) / 3 (the default) stay as a knob for the experiment.
static std::mutex g_host_mu;
static std::unordered_map<void *, int> g_host_registered;     // pointers from posix_memalign + hipHostRegister
static int host_alloc_mode()
{
    static int mode = -1;
    if (mode < 0) { const char *e = getenv("P264AMD_HOST_ALLOC"); int m = e ? atoi(e) : 3; mode = (m < 0 || m > 3) ? 3 : m; }
    return mode;
}
extern "C" void *p264hip_host_alloc(size_t bytes)
{
    void *p = nullptr;
    const size_t n = bytes ? bytes : 1;
    const int mode = host_alloc_mode();
    if (mode >= 2) {
        const bool huge = mode == 3 && n >= ((size_t)256 << 10);
        const size_t al = huge ? ((size_t)2 << 20) : 4096, rounded = (n + al - 1) & ~(al - 1);
        if (posix_memalign(&p, al, rounded) == 0) {
            if (huge) (void)madvise(p, rounded, MADV_HUGEPAGE);
            for (size_t o = 0; o < rounded; o += 4096) ((volatile char *)p)[o] = 0;          // (faulted in before they are pinned)
            if (hipHostRegister(p, rounded, hipHostRegisterPortable) == hipSuccess) {
                std::lock_guard<std::mutex> lk(g_host_mu);
                g_host_registered[p] = 1;
                return p;
            }
            (void)hipGetLastError();
            free(p); p = nullptr;
        }
        // (could not be registered: pinned memory of the runtime's own)
    }
    if (hipHostMalloc(&p, n, hipHostMallocPortable | (mode == 1 ? hipHostMallocNonCoherent : 0)) != hipSuccess) return nullptr;
    return p;
}

extern "C" void p264hip_host_free(void *p)
{
    if (!p) return;
    bool registered;
    { std::lock_guard<std::mutex> lk(g_host_mu); registered = g_host_registered.erase(p) != 0; }
    if (registered) { (void)hipHostUnregister(p); free(p); }
    else (void)hipHostFree(p);
}

extern "C" int p264hip_marker(p264hip_ctx *c)
{
    if (!c) return fail(P264HIP_EINVAL, "null context");
    HIPCHK(hipSetDevice(c->device));
    const int m = c->next_marker; c->next_marker = (c->next_marker + 1) % P264HIP_MARKERS;
    if (!c->markers[m]) HIPCHK(hipEventCreateWithFlags(&c->markers[m], hipEventDisableTiming));
    HIPCHK(hipEventRecord(c->markers[m], c->stream));
    return m;
}

extern "C" int p264hip_marker_wait(p264hip_ctx *c, int marker)
{
    if (!c || marker < 0 || marker >= P264HIP_MARKERS || !c->markers[marker]) return fail(P264HIP_EINVAL, "p264hip_marker_wait: bad marker %d", marker);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipEventSynchronize(c->markers[marker]));
    return P264HIP_OK;
}

extern "C" int p264hip_clone_picture(p264hip_ctx *c, int dst, int src)
{
    if (!c || dst < 0 || src < 0 || dst >= c->max_pictures || src >= c->max_pictures || dst == src || !c->pics[(size_t)src].valid)
        return fail(P264HIP_EINVAL, "p264hip_clone_picture: bad slots %d <- %d", dst, src);
    HIPCHK(hipSetDevice(c->device));
    unpend(c, dst);
    { const int rc = expand_pending(c); if (rc) return rc; }   // (src may still be a compact block)
    PicSlot &d = c->pics[(size_t)dst], &s = c->pics[(size_t)src];
    size_t need = s.bytes;
    if (need > d.cap) {
        if (d.dev) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(d.dev)); d.dev = nullptr; d.cap = 0; }
        hipError_t e = hipMalloc((void **)&d.dev, need);
        if (e != hipSuccess) return fail(P264HIP_ENOMEM, "hipMalloc(%zu) for picture clone: %s", need, hipGetErrorString(e));
        d.cap = need;
    }
    HIPCHK(hipMemcpyAsync(d.dev, s.dev, need, hipMemcpyDeviceToDevice, c->stream));
    // the verdict of the record check is kept per slot (k_check_records -> d_slot_bad[slot]): an unchecked block takes its
    // verdict along (behind the check on the same stream), any other clone clears what an earlier tenant of dst left there
    if (s.unchecked) HIPCHK(hipMemcpyAsync(c->d_slot_bad + dst, c->d_slot_bad + src, sizeof(int), hipMemcpyDeviceToDevice, c->stream));
    else HIPCHK(hipMemsetAsync(c->d_slot_bad + dst, 0, sizeof(int), c->stream));
    d.off_mv = s.off_mv; d.off_ref = s.off_ref; d.off_i4 = s.off_i4; d.off_coef = s.off_coef; d.off_mv_l1 = s.off_mv_l1; d.off_ref_l1 = s.off_ref_l1; d.off_weights = s.off_weights; d.bytes = s.bytes;
    d.meta = s.meta; d.valid = true; d.unchecked = s.unchecked; d.last_use = s.last_use = ++c->epoch;
    return P264HIP_OK;
}

static hipEvent_t get_event(p264hip_ctx *c)
{
    if (!c->event_pool.empty()) { hipEvent_t e = c->event_pool.back(); c->event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr; (void)hipEventCreate(&e); return e;
}

struct ScopedStamp {                   // HIP events on the context's own stream, around one launch
    p264hip_ctx *c; int k; hipEvent_t a = nullptr, b = nullptr;
    ScopedStamp(p264hip_ctx *c_, int k_) : c(c_), k(k_) { if (c->timing) { a = get_event(c); b = get_event(c); (void)hipEventRecord(a, c->stream); } }
    ~ScopedStamp() { if (c->timing) { (void)hipEventRecord(b, c->stream); c->stamps.push_back({ a, b, k }); } }
};

extern "C" int p264hip_reconstruct(p264hip_ctx *c, const int *pic_ids, const int *streams, int n)
{
    if (!c || !pic_ids || !streams || n < 1) return fail(P264HIP_EINVAL, "p264hip_reconstruct: bad argument");
    HIPCHK(hipSetDevice(c->device));
    { const int rc = expand_pending(c); if (rc) return rc; }   // compact uploads since the last launch: one expansion kernel for all of them
    if (n > c->batch_cap) {
        HIPCHK(hipStreamSynchronize(c->stream));
        for (int i = 0; i < BATCH_RING; i++) {
            if (c->h_batch[i]) (void)hipHostFree(c->h_batch[i]);
            if (c->d_batch[i]) (void)hipFree(c->d_batch[i]);
            c->h_batch[i] = nullptr; c->d_batch[i] = nullptr;
            HIPCHK(hipHostMalloc((void **)&c->h_batch[i], (size_t)n * sizeof(PicDev), hipHostMallocDefault));
            HIPCHK(hipMalloc((void **)&c->d_batch[i], (size_t)n * sizeof(PicDev)));
            if (!c->batch_free[i]) HIPCHK(hipEventCreateWithFlags(&c->batch_free[i], hipEventDisableTiming));
            HIPCHK(hipEventRecord(c->batch_free[i], c->stream));
        }
        if (c->d_edge) (void)hipFree(c->d_edge);
        if (c->d_mc) (void)hipFree(c->d_mc);
        if (c->d_is_intra) (void)hipFree(c->d_is_intra);
        c->d_edge = nullptr; c->d_mc = nullptr; c->d_is_intra = nullptr;
        HIPCHK(hipMalloc((void **)&c->d_is_intra, (size_t)n * c->g.n_mb));
        HIPCHK(hipMalloc((void **)&c->d_edge, (size_t)n * c->g.n_mb * sizeof(EdgeInfo)));
        HIPCHK(hipMalloc((void **)&c->d_mc, (size_t)n * c->ml.words * sizeof(uint32_t)));
        c->batch_cap = n;
    }
    {   // pictures that came through p264hip_input_commit: the verdicts of their record checks, one wait for the whole batch
        bool any_unchecked = false;
        for (int i = 0; i < n; i++) { const int id = pic_ids[i]; if (id >= 0 && id < c->max_pictures && c->pics[(size_t)id].unchecked) any_unchecked = true; }
        if (any_unchecked) {
            std::vector<int> bad((size_t)c->max_pictures);
            const uint64_t upto = c->epoch;
            HIPCHK(hipMemcpyAsync(bad.data(), c->d_slot_bad, sizeof(int) * bad.size(), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            c->done_epoch = upto;
            for (int i = 0; i < n; i++) {
                const int id = pic_ids[i];
                if (id < 0 || id >= c->max_pictures || !c->pics[(size_t)id].unchecked) continue;
                if (bad[(size_t)id]) { c->pics[(size_t)id].valid = false; c->pics[(size_t)id].unchecked = false; return fail(P264HIP_EINVAL, "picture slot %d: a macroblock's coefficient blocks lie outside coefs[] (the block a device producer committed is inconsistent)", id); }
                c->pics[(size_t)id].unchecked = false;
            }
        }
    }
    const int r = c->ring; c->ring = (c->ring + 1) % BATCH_RING;
    HIPCHK(hipEventSynchronize(c->batch_free[r]));            // the copy that last used this staging buffer is done
    PicDev *hb = c->h_batch[r];
    bool any_p = false, any_b = false, any_i = false;       // any picture with inter macroblocks / any B picture / any I picture
    for (int i = 0; i < n; i++) {                          // two pictures of one call must not share a stream: they would race on its frames
        const int st = streams[i];
        if (st < 0 || st >= c->n_streams) return fail(P264HIP_EINVAL, "stream %d out of range", st);
        c->stream_seen[(size_t)st] = 0;
    }
    for (int i = 0; i < n; i++) {
        int id = pic_ids[i], st = streams[i];
        if (id < 0 || id >= c->max_pictures || !c->pics[(size_t)id].valid) return fail(P264HIP_EINVAL, "picture slot %d is empty", id);
        if (c->stream_seen[(size_t)st]) return fail(P264HIP_EINVAL, "stream %d is named twice in one batch (entries %d and %d)", st, c->stream_seen[(size_t)st] - 1, i);
        c->stream_seen[(size_t)st] = i + 1;
        const PicSlot &s = c->pics[(size_t)id];
        PicDev &d = hb[i];
        memset(&d, 0, sizeof d);
        d.mb = (const p264hip_mb_t *)s.dev;
        d.mv = (const int *)(s.dev + s.off_mv);
        d.ref_idx = (const int8_t *)(s.dev + s.off_ref);
        d.i4modes = s.dev + s.off_i4;
        d.coefs = (const int16_t *)(s.dev + s.off_coef);
        d.dst = frame_ptr(c, st, s.meta.dst_slot);
        d.store = frame_ptr(c, st, 0);
        d.store_bytes = (uint32_t)(c->frame_bytes * (size_t)c->slots);
        d.dst_off = (uint32_t)(c->frame_bytes * (size_t)s.meta.dst_slot);
        for (int k = 0; k < P264HIP_MAX_REFS; k++)
            d.ref_off[k] = (uint32_t)(c->frame_bytes * (size_t)(k < s.meta.n_ref ? s.meta.ref_slot[k] : (s.meta.n_ref ? s.meta.ref_slot[0] : s.meta.dst_slot)));
        d.n_ref = s.meta.n_ref; d.slice_type = s.meta.slice_type;
        d.chroma_qp_offset = s.meta.chroma_qp_offset; d.deblock = s.meta.deblock;
        d.alpha_off = s.meta.alpha_c0_offset; d.beta_off = s.meta.beta_offset;
        if (s.meta.slice_type == P264_SLICE_B) {
            d.mv_l1 = (const int *)(s.dev + s.off_mv_l1);
            d.ref_idx_l1 = (const int8_t *)(s.dev + s.off_ref_l1);
            d.bipred_w = (const int16_t *)(s.dev + s.off_weights);
            d.n_ref_l1 = s.meta.n_ref_l1; d.weighted = s.meta.weighted_bipred;
            for (int k = 0; k < P264HIP_MAX_REFS; k++)
                d.ref_off_l1[k] = (uint32_t)(c->frame_bytes * (size_t)(k < s.meta.n_ref_l1 ? s.meta.ref_slot_l1[k] : s.meta.ref_slot_l1[0]));
            any_b = true;
        }
        any_p |= s.meta.slice_type != P264_SLICE_I;
        any_i |= s.meta.slice_type == P264_SLICE_I;
    }
    // from here on work that reads the batch's input slots is (about to be) queued: the slots carry the new epoch BEFORE the first
    // launch, so that whichever way this function returns - a launch error half-way included - a later p264hip_input_reserve
    // of one of them waits for the stream instead of letting a peer overwrite a block under running kernels
    ++c->epoch;
    for (int i = 0; i < n; i++) c->pics[(size_t)pic_ids[i]].last_use = c->epoch;
    ScopedStamp whole(c, 3);
    HIPCHK(hipMemcpyAsync(c->d_batch[r], hb, (size_t)n * sizeof(PicDev), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipEventRecord(c->batch_free[r], c->stream));
    const Geom g = c->g;
    const uint32_t inv_mbw = (uint32_t)(((1ull << 32) - 1) / (unsigned)g.mb_w);
    bool bs_fused = false;
    p264hip_launch_info_t &li = c->last;
    memset(&li, 0, sizeof li);
    li.pictures = n; li.compute_units = c->n_cu;
    if (any_p) {
        // motion compensation + residual of all inter macroblocks: device-side counting sort of the work items by what the
        // interpolation has to do, then the luma and chroma kernels over the sorted lists (kernel_mc.h), side by side
        ScopedStamp t(c, 0);
        const McLayout ml = c->ml;
        if (any_b) hipLaunchKernelGGL(k_mc_sort_b, dim3(n), dim3(MC_SORT_THREADS), 0, c->stream, c->d_batch[r], c->d_mc, g, ml, inv_mbw, c->d_is_intra);
        else hipLaunchKernelGGL(k_mc_sort, dim3(n), dim3(MC_SORT_THREADS), 0, c->stream, c->d_batch[r], c->d_mc, g, ml, inv_mbw, c->d_is_intra);
        // one launch for luma / chroma, macroblock / quadrant items (k_mc): every picture gets the same number of workgroups,
        // which split into the four roles on the device.  Enough workgroups per picture to fill the chip a few times over,
        // no more than there can be chunks (four wavefronts per workgroup, one chunk per wavefront pass).
        int wgs = (c->n_cu * 48 + n - 1) / n;   // (a dozen rounds of workgroups on small batches.  Round 5, 256 pictures per launch: 48 / 64 / 96 / 128 / 192 / 256
                                           //  per picture -> 0.685 / 0.687 / 0.691 / 0.704 / 0.724 / 0.751 ms for the stage - a wavefront's first chunk has no prefetch)
        if (wgs < 48) wgs = 48;            // (2048 pictures: 24 / 32 / 48 / 64 / 96 per picture -> 5.34 / 5.31 / 5.24 / 5.31 / 5.34 ms; with 24 the stage's reads grew by half:
                                           //  a picture's roles drift apart and stop sharing reference lines in L2)
        int max_wgs = 0;
        for (int l = 0; l < ML_LISTS; l++) max_wgs += (int)(ml.max_chunks[l] + 3) / 4;
        if (wgs > max_wgs) wgs = max_wgs;
        if (c->tune_mc_wgs >= 4 && c->tune_mc_wgs <= max_wgs) wgs = c->tune_mc_wgs;
        if (wgs < 4) wgs = 4;
        li.mc_wgs_per_picture = wgs;
        hipLaunchKernelGGL(k_mc, dim3(((size_t)wgs * n + 7) / 8 * 8), dim3(256), 0, c->stream, (const PicDev *)c->d_batch[r], (const uint32_t *)c->d_mc, g, ml,
                           wgs, wgs * n, (uint32_t)(((1ull << 32) - 1) / (unsigned)wgs));
        // B pictures: the blocks that predict from both lists get their second prediction (and their residual) in a second pass
        if (any_b)
            hipLaunchKernelGGL(k_mc_second, dim3(((size_t)wgs * n + 7) / 8 * 8), dim3(256), 0, c->stream, (const PicDev *)c->d_batch[r], (const uint32_t *)c->d_mc, g, ml,
                               wgs, wgs * n, (uint32_t)(((1ull << 32) - 1) / (unsigned)wgs));
    }
    {
        ScopedStamp t(c, 1);
        // one workgroup per picture and role: 16 wavefronts while every picture can have a CU to itself, else 8 or 4 so that two or
        // four pictures share a CU (measured +19 % at 512 and +9 % at 1024 pictures; 2048: 2 / 4 / 8 / 16 -> 1.02 / 0.87 / 0.90 / 1.10 ms)
        int intra_waves = n > 2 * c->n_cu ? INTRA_ROW_WAVES / 4 : n > c->n_cu ? INTRA_ROW_WAVES / 2 : INTRA_ROW_WAVES;
        if (c->tune_intra_waves >= 1 && c->tune_intra_waves <= INTRA_ROW_WAVES) intra_waves = c->tune_intra_waves;
        li.intra_waves = intra_waves;
        // luma and chroma of a picture are independent chains: as two workgroups they run side by side
        if (any_i) hipLaunchKernelGGL(k_intra, dim3(n, 2), dim3(intra_waves * 64), (size_t)intra_waves * sizeof(IntraLds), c->stream, c->d_batch[r], g, c->d_status, (const uint8_t *)c->d_is_intra);
        else {
            // (P / B pictures only.  Batches without B pictures: the loop filter's edge info is computed by extra workgroups of this
            // launch, kernel_intra.h)
            bs_fused = !any_b && c->tune_bs_fused != 0;
            const int bs_wgs = bs_fused ? (c->tune_bs_fused > 0 ? c->tune_bs_fused : INTRA_BS_WGS) : 0;
            li.edge_info_fused = bs_wgs;
            hipLaunchKernelGGL(k_intra_sparse, dim3((unsigned)n * (2 + bs_wgs)), dim3(intra_waves * 64), (size_t)intra_waves * sizeof(IntraLds), c->stream, c->d_batch[r], g, c->d_status,
                               (const uint8_t *)c->d_is_intra, c->d_edge, inv_mbw, bs_wgs);
        }
    }
    {
        ScopedStamp t(c, 2);
        // edge info (boundary strengths, averaged QPs per edge class): everything about an edge that does not depend on samples
        if (any_b) hipLaunchKernelGGL(k_deblock_bs<true>, dim3((g.n_mb + 255) / 256, n), dim3(256), 0, c->stream, c->d_batch[r], g, c->d_edge, inv_mbw);
        else if (!bs_fused) hipLaunchKernelGGL(k_deblock_bs<false>, dim3((g.n_mb + 255) / 256, n), dim3(256), 0, c->stream, c->d_batch[r], g, c->d_edge, inv_mbw);
        // pictures per workgroup = as many as it takes to cover the batch with one workgroup per CU (a second, half-empty round
        // of workgroups costs more than sharing a workgroup: 1280 pictures as 320 workgroups of 4 took 5.35 ms, as 256 of 5 ...)
        int per_wg = (n + c->n_cu - 1) / c->n_cu;
        if (per_wg < 1) per_wg = 1;
        if (per_wg > MAX_PICS_PER_WG) per_wg = MAX_PICS_PER_WG;
        // bands of 8 rows of one picture per wavefront while every picture has a CU to itself, else 4 rows of two pictures; a
        // workgroup with more pictures than a wavefront holds (8 >> rb_log2) works on them in groups (units = band x group).
        // (Round 4: 2 rows x 4 pictures ran as fast, 2.70 ms, but every second row's bottom lines cross a band - 0.4 GB more
        // through memory per launch.)
        // Which shape: the stage's time follows the wavefront-iterations it issues (round 5: 1.07 ms per 16 units of 127 iterations at
        // 2 ... 15 pictures per workgroup, half-empty units included - it is bound by vector-instruction issue).  Bands of 4 rows hold
        // two pictures per wavefront: an odd picture count leaves one unit in every band half empty; bands of 8 rows hold one picture,
        // but run 8 iterations longer and the last band of a 68-row picture is half empty.  Take the cheaper one.
        auto units_cost = [&](int lg) {
            const int rows = 1 << lg, pw = 8 >> lg;
            const long bands = (g.mb_h + rows - 1) / rows, groups = (per_wg + pw - 1) / pw;
            return bands * groups * (long)(g.mb_w + 1 + DB_LAG * (rows - 1));
        };
        int rb_log2 = units_cost(3) < units_cost(2) ? 3 : 2;
        // an odd number (>= 3) of pictures: the pairs in bands of 4 rows, the last picture alone in bands of 8 (k_deblock, odd_single)
        const long cost_mixed = (per_wg >= 3 && (per_wg & 1)) ? ((g.mb_h + 3) / 4) * (long)(per_wg / 2) * (g.mb_w + 1 + 3 * DB_LAG) + ((g.mb_h + 7) / 8) * (long)(g.mb_w + 1 + 7 * DB_LAG) : -1;
        int odd_single = cost_mixed >= 0 && cost_mixed < units_cost(rb_log2);
        if (odd_single) rb_log2 = 2;
        if (c->tune_rb_log2 >= 1 && c->tune_rb_log2 <= 3) { rb_log2 = c->tune_rb_log2; odd_single = 0; }
        if (c->tune_pics_per_wg >= 1 && c->tune_pics_per_wg <= MAX_PICS_PER_WG) { per_wg = c->tune_pics_per_wg; odd_single = 0; }
        if (c->tune_odd_single >= 0) odd_single = c->tune_odd_single && rb_log2 == 2 && per_wg >= 3 && (per_wg & 1);
        const int n_bands = (g.mb_h + (1 << rb_log2) - 1) >> rb_log2;
        const int n_units = !odd_single ? n_bands * ((per_wg + (8 >> rb_log2) - 1) / (8 >> rb_log2)) : n_bands * (per_wg / 2) + (g.mb_h + 7) / 8;
        int waves = n_units < ROW_WAVES ? n_units : ROW_WAVES;
        if (c->tune_db_waves >= 1 && c->tune_db_waves < waves) waves = c->tune_db_waves;
        li.deblock_pics_per_wg = per_wg; li.deblock_rb_log2 = rb_log2; li.deblock_waves = waves; li.deblock_wgs = (n + per_wg - 1) / per_wg;
        li.deblock_odd_single = odd_single;
        hipLaunchKernelGGL(k_deblock, dim3((n + per_wg - 1) / per_wg), dim3(waves * 64), 0, c->stream, c->d_batch[r], g,
                           (const EdgeInfo *)c->d_edge, c->d_status, n, rb_log2, per_wg, odd_single);
    }
    HIPCHK(hipGetLastError());
    return P264HIP_OK;
}

extern "C" int64_t p264hip_upload_copies(p264hip_ctx *c) { return c ? (int64_t)c->upload_copies : -1; }

extern "C" int p264hip_last_launch(p264hip_ctx *c, p264hip_launch_info_t *out)
{
    if (!c || !out) return fail(P264HIP_EINVAL, "p264hip_last_launch: null argument");
    *out = c->last;
    return P264HIP_OK;
}

extern "C" int p264hip_submit(p264hip_ctx *c, int stream, const p264hip_picture_t *pic)
{
    if (!c || !pic || stream < 0 || stream >= c->n_streams || stream >= c->max_pictures)
        return fail(P264HIP_EINVAL, "p264hip_submit: bad argument (stream %d)", stream);
    int rc = p264hip_upload(c, stream, pic, 1);               // input slot `stream` is this stream's staging slot
    if (rc) return rc;
    return p264hip_reconstruct(c, &stream, &stream, 1);
}

extern "C" int p264hip_submit_async(p264hip_ctx *c, int stream, const p264hip_picture_t *pic)
{
    if (!c || !pic || stream < 0 || stream >= c->n_streams || stream >= c->max_pictures)
        return fail(P264HIP_EINVAL, "p264hip_submit_async: bad argument (stream %d)", stream);
    int rc = p264hip_upload_async(c, stream, pic);
    if (rc) return rc;
    return p264hip_reconstruct(c, &stream, &stream, 1);
}

static int drain_stamps(p264hip_ctx *c)
{
    for (auto &s : c->stamps) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) { c->ms_sum[s.k] += ms; c->ms_cnt[s.k]++; }
        c->event_pool.push_back(s.a); c->event_pool.push_back(s.b);
    }
    c->stamps.clear();
    return 0;
}

#ifdef EXPD_STAMPS
extern "C" int p264hip_db_stamps_read(unsigned long long *h, size_t bytes);     // k_deblock.hip
#endif
extern "C" int p264hip_sync(p264hip_ctx *c)
{
#ifdef EXPD_STAMPS
    if (const char *path = getenv("P264AMD_STAMPS_OUT")) {     // diagnostic build: the clock stamps of one k_deblock wavefront
        static unsigned long long h[256 * 8];
        (void)hipDeviceSynchronize();
        if (p264hip_db_stamps_read(h, sizeof h) == 0) {
            FILE *f = fopen(path, "w");
            if (f) { for (int i = 0; i < 256; i++) { for (int k = 0; k < 8; k++) fprintf(f, "%llu ", h[i * 8 + k]); fprintf(f, "\n"); } fclose(f); }
        }
    }
#endif
    if (!c) return fail(P264HIP_EINVAL, "null context");
    HIPCHK(hipSetDevice(c->device));
    { const int rc = expand_pending(c); if (rc) return rc; }
    { const uint64_t upto = c->epoch; HIPCHK(hipStreamSynchronize(c->stream)); c->done_epoch = upto; }
    drain_stamps(c);
    int st = 0;
    HIPCHK(hipMemcpy(&st, c->d_status, sizeof st, hipMemcpyDeviceToHost));
    if (st) {
        (void)hipMemset(c->d_status, 0, sizeof(int));
        return fail(P264HIP_EHIP, "a macroblock-row dependency wait timed out on the device (status %d)", st);
    }
    return P264HIP_OK;
}

static int frame_io(p264hip_ctx *c, int stream, int slot, uint8_t *y, int ys, uint8_t *u, uint8_t *v, int cs, bool read)
{
    if (!c || stream < 0 || stream >= c->n_streams || slot < 0 || slot >= c->slots || !y || !u || !v || ys < c->g.w || cs < c->g.cw)
        return fail(P264HIP_EINVAL, "frame access: bad argument (stream %d slot %d strides %d/%d)", stream, slot, ys, cs);
    HIPCHK(hipSetDevice(c->device));
    int rc = p264hip_sync(c);
    if (rc) return rc;
    // frames live in the strip layout on the device; the host sees planes, through a planar staging buffer
    uint8_t *f = frame_ptr(c, stream, slot);
    const Geom &g = c->g;
    const size_t ysz = (size_t)g.w * g.h, csz = (size_t)g.cw * g.ch;
    if (!c->d_planar) HIPCHK(hipMalloc((void **)&c->d_planar, ysz + 2 * csz));
    uint8_t *s = c->d_planar;
    hipMemcpyKind k = read ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice;
    struct { uint8_t *host; int hs; uint8_t *dev; int w, h; } pl[3] = {
        { y, ys, s, g.w, g.h }, { u, cs, s + ysz, g.cw, g.ch }, { v, cs, s + ysz + csz, g.cw, g.ch } };
    const int n_dw = g.n_mb * 96;
    if (read) {
        hipLaunchKernelGGL(k_tile_convert, dim3((n_dw + 255) / 256), dim3(256), 0, c->stream, f, s, g, 1);
        HIPCHK(hipStreamSynchronize(c->stream));
        for (auto &p : pl) HIPCHK(hipMemcpy2D(p.host, (size_t)p.hs, p.dev, (size_t)p.w, (size_t)p.w, (size_t)p.h, k));
    } else {
        for (auto &p : pl) HIPCHK(hipMemcpy2D(p.dev, (size_t)p.w, p.host, (size_t)p.hs, (size_t)p.w, (size_t)p.h, k));
        hipLaunchKernelGGL(k_tile_convert, dim3((n_dw + 255) / 256), dim3(256), 0, c->stream, f, s, g, 0);
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    HIPCHK(hipGetLastError());
    return P264HIP_OK;
}

extern "C" int p264hip_read_frame_async(p264hip_ctx *c, int stream, int slot, uint8_t *y, int ys, uint8_t *u, uint8_t *v, int cs)
{
    if (!c || stream < 0 || stream >= c->n_streams || slot < 0 || slot >= c->slots || !y || !u || !v || ys < c->g.w || cs < c->g.cw)
        return fail(P264HIP_EINVAL, "frame access: bad argument (stream %d slot %d strides %d/%d)", stream, slot, ys, cs);
    HIPCHK(hipSetDevice(c->device));
    const Geom &g = c->g;
    const size_t ysz = (size_t)g.w * g.h, csz = (size_t)g.cw * g.ch;
    if (!c->d_planar) HIPCHK(hipMalloc((void **)&c->d_planar, ysz + 2 * csz));
    uint8_t *s = c->d_planar;
    const int n_dw = g.n_mb * 96;
    hipLaunchKernelGGL(k_tile_convert, dim3((n_dw + 255) / 256), dim3(256), 0, c->stream, frame_ptr(c, stream, slot), s, g, 1);
    HIPCHK(hipMemcpy2DAsync(y, (size_t)ys, s, (size_t)g.w, (size_t)g.w, (size_t)g.h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpy2DAsync(u, (size_t)cs, s + ysz, (size_t)g.cw, (size_t)g.cw, (size_t)g.ch, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpy2DAsync(v, (size_t)cs, s + ysz + csz, (size_t)g.cw, (size_t)g.cw, (size_t)g.ch, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipGetLastError());
    return P264HIP_OK;
}

extern "C" int p264hip_read_frame(p264hip_ctx *c, int stream, int slot, uint8_t *y, int ys, uint8_t *u, uint8_t *v, int cs)
{ return frame_io(c, stream, slot, y, ys, u, v, cs, true); }

extern "C" int p264hip_write_frame(p264hip_ctx *c, int stream, int slot, const uint8_t *y, int ys, const uint8_t *u, const uint8_t *v, int cs)
{ return frame_io(c, stream, slot, (uint8_t *)y, ys, (uint8_t *)u, (uint8_t *)v, cs, false); }

extern "C" int p264hip_timing_enable(p264hip_ctx *c, int on)
{
    if (!c) return fail(P264HIP_EINVAL, "null context");
    int rc = p264hip_sync(c);
    c->timing = on != 0;
    return rc;
}

extern "C" int p264hip_timing_reset(p264hip_ctx *c)
{
    if (!c) return fail(P264HIP_EINVAL, "null context");
    int rc = p264hip_sync(c);
    for (int i = 0; i < P264HIP_NKERNELS; i++) { c->ms_sum[i] = 0; c->ms_cnt[i] = 0; }
    return rc;
}

extern "C" int p264hip_timing_read(p264hip_ctx *c, double ms_sum[P264HIP_NKERNELS], int64_t count[P264HIP_NKERNELS])
{
    if (!c || !ms_sum || !count) return fail(P264HIP_EINVAL, "null argument");
    int rc = p264hip_sync(c);
    for (int i = 0; i < P264HIP_NKERNELS; i++) { ms_sum[i] = c->ms_sum[i]; count[i] = c->ms_cnt[i]; }
    return rc;
}
