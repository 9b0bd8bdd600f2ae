/* host-only sanitizer build: the HIP layer is absent, every p264hip entry point fails */
#include <stddef.h>
#include "p264hip.h"
int p264hip_create(p264hip_ctx **o, int d, int w, int h, int n, int s, int m) { (void)o;(void)d;(void)w;(void)h;(void)n;(void)s;(void)m; return P264HIP_ENODEV; }
void p264hip_destroy(p264hip_ctx *c) { (void)c; }
const char *p264hip_last_error(void) { return "stub"; }
int p264hip_device_count(void) { return 0; }
int p264hip_upload(p264hip_ctx *c, int f, const p264hip_picture_t *p, int n) { (void)c;(void)f;(void)p;(void)n; return -1; }
int p264hip_upload_async(p264hip_ctx *c, int s, const p264hip_picture_t *p) { (void)c;(void)s;(void)p; return -1; }
void *p264hip_host_alloc(size_t b) { (void)b; return NULL; }
void p264hip_host_free(void *p) { (void)p; }
int p264hip_marker(p264hip_ctx *c) { (void)c; return -1; }
int p264hip_marker_wait(p264hip_ctx *c, int m) { (void)c;(void)m; return -1; }
int p264hip_reconstruct(p264hip_ctx *c, const int *a, const int *b, int n) { (void)c;(void)a;(void)b;(void)n; return -1; }
int p264hip_submit(p264hip_ctx *c, int s, const p264hip_picture_t *p) { (void)c;(void)s;(void)p; return -1; }
int p264hip_sync(p264hip_ctx *c) { (void)c; return -1; }
int p264hip_read_frame(p264hip_ctx *c, int s, int sl, uint8_t *y, int ys, uint8_t *u, uint8_t *v, int cs) { (void)c;(void)s;(void)sl;(void)y;(void)ys;(void)u;(void)v;(void)cs; return -1; }
int p264hip_clone_picture(p264hip_ctx *c, int d, int s) { (void)c;(void)d;(void)s; return -1; }
int p264hip_write_frame(p264hip_ctx *c, int s, int sl, const uint8_t *y, int ys, const uint8_t *u, const uint8_t *v, int cs) { (void)c;(void)s;(void)sl;(void)y;(void)ys;(void)u;(void)v;(void)cs; return -1; }
int p264hip_timing_enable(p264hip_ctx *c, int on) { (void)c;(void)on; return -1; }
int p264hip_timing_read(p264hip_ctx *c, double *a, int64_t *b) { (void)c;(void)a;(void)b; return -1; }
int p264hip_timing_reset(p264hip_ctx *c) { (void)c; return -1; }
int p264hip_submit_async(p264hip_ctx *c, int s, const p264hip_picture_t *p) { (void)c;(void)s;(void)p; return -1; }
int p264hip_read_frame_async(p264hip_ctx *c, int s, int sl, uint8_t *y, int ys, uint8_t *u, uint8_t *v, int cs) { (void)c;(void)s;(void)sl;(void)y;(void)ys;(void)u;(void)v;(void)cs; return -1; }
int p264hip_upload_packed(p264hip_ctx *c, int s, const p264hip_picture_t *d, const void *p, size_t n) { (void)c;(void)s;(void)d;(void)p;(void)n; return -1; }
int p264hip_input_reserve(p264hip_ctx *c, int s, const p264hip_picture_t *d, void **dev, size_t *n) { (void)c;(void)s;(void)d;(void)dev;(void)n; return -1; }
int p264hip_input_commit(p264hip_ctx *c, int s) { (void)c;(void)s; return -1; }
int p264hip_frame_planar_device(p264hip_ctx *c, int s, int sl, int i, void **dev, size_t *n) { (void)c;(void)s;(void)sl;(void)i;(void)dev;(void)n; return -1; }
int p264hip_last_launch(p264hip_ctx *c, p264hip_launch_info_t *o) { (void)c;(void)o; return -1; }
int p264hip_build_info(void) { return 0; }
int p264hip_copy_to_device(void *d, const void *h, size_t n) { (void)d;(void)h;(void)n; return -1; }
int p264hip_copy_from_device(void *h, const void *d, size_t n) { (void)h;(void)d;(void)n; return -1; }
/* the RCCL transport lives with the HIP code: not part of the host-only build */
#include "p264fan.h"
int p264fan_rccl_unique_id(uint8_t id[128]) { (void)id; return -1; }
int p264fan_rccl_transport(p264fan_transport_t *t, int r, int w, const uint8_t id[128], int d) { (void)t;(void)r;(void)w;(void)id;(void)d; return -1; }
int64_t p264hip_upload_copies(p264hip_ctx *c) { (void)c; return -1; }
int p264hip_upload_compact(p264hip_ctx *c, int s, const p264hip_picture_t *d, const void *p, size_t n) { (void)c;(void)s;(void)d;(void)p;(void)n; return -1; }
