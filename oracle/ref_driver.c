/* oracle/ref_driver.c - TEST INFRASTRUCTURE ONLY.
 *
 * A small driver of OUR OWN that links the *real* reference decoder objects
 * (compiled in place from /root/reference by oracle/Makefile into
 * oracle/_ref/) and drives them through the reference's public API
 * (p264.h:266,351,379-382).  It replaces the reference CLI
 * (p264decoder.c:164-381), which we do not compile because it needs a
 * generated config.h.
 *
 *   ref_driver hash   in.264          -> one "frame_idx sha256 width height" line per picture
 *   ref_driver yuv    in.264 out.yuv  -> MB-aligned planar I420, like write_frame (p264decoder.c:126-156)
 *   ref_driver time   in.264 [loops]  -> decode only, print "frames N usec T fps F"
 *
 * Used to (1) pin the oracle restatement and the product parser against the
 * reference itself and (2) as the "reference" CPU baseline in bench.py.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <time.h>
#include "p264.h"
#include "sha256.h"

static uint8_t *slurp(const char *path, size_t *n)
{
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t *b = malloc((size_t)sz + 8);
    if (fread(b, 1, (size_t)sz, f) != (size_t)sz) { perror("fread"); exit(2); }
    memset(b + sz, 0, 8);
    fclose(f); *n = (size_t)sz; return b;
}

static double now_us(void)
{
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

typedef void (*frame_cb)(p264_picture_t *pic, int idx, void *ud);

/* Walk Annex-B NAL units and feed each to the reference decoder. */
static int decode_stream(const uint8_t *buf, size_t n, frame_cb cb, void *ud)
{
    p264_param_t param;
    p264_param_default(&param);
    p264_t *h = p264_decoder_open(&param);
    if (!h) { fprintf(stderr, "p264_decoder_open failed\n"); exit(2); }
    p264_nal_t nal; memset(&nal, 0, sizeof nal);
    nal.p_payload = malloc(n + 16);
    int frames = 0;
    size_t i = 0;
    /* find first start code */
    while (i + 3 <= n && !(buf[i] == 0 && buf[i+1] == 0 && buf[i+2] == 1)) i++;
    while (i + 3 <= n) {
        size_t start = i + 3, j = start;
        while (j + 3 <= n && !(buf[j] == 0 && buf[j+1] == 0 && buf[j+2] == 1)) j++;
        size_t end = (j + 3 <= n) ? j : n;
        size_t next = end;
        while (end > start && buf[end-1] == 0) end--;      /* zeros before a start code belong to it */
        if (end > start) {
            p264_picture_t *pic = NULL;
            p264_nal_decode(&nal, (void *)(buf + start), (int)(end - start));
            p264_decoder_decode(h, &pic, &nal);
            if (pic) { if (cb) cb(pic, frames, ud); frames++; }
        }
        i = next;
    }
    p264_decoder_close(h);
    free(nal.p_payload);
    return frames;
}

static void cb_hash(p264_picture_t *pic, int idx, void *ud)
{
    (void)ud;
    sha256_t c; sha256_init(&c);
    for (int p = 0; p < 3; p++) {
        int w = p ? pic->i_width >> 1 : pic->i_width, hgt = p ? pic->i_height >> 1 : pic->i_height;
        const uint8_t *s = pic->img.plane[p];
        for (int y = 0; y < hgt; y++, s += pic->img.i_stride[p]) sha256_update(&c, s, (size_t)w);
    }
    uint8_t d[32]; char hex[65]; sha256_final(&c, d); sha256_hex(d, hex);
    printf("%d %s %d %d\n", idx, hex, pic->i_width, pic->i_height);
}

static void cb_yuv(p264_picture_t *pic, int idx, void *ud)
{
    (void)idx; FILE *f = ud;
    for (int p = 0; p < 3; p++) {
        int w = p ? pic->i_width >> 1 : pic->i_width, hgt = p ? pic->i_height >> 1 : pic->i_height;
        const uint8_t *s = pic->img.plane[p];
        for (int y = 0; y < hgt; y++, s += pic->img.i_stride[p]) fwrite(s, 1, (size_t)w, f);
    }
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s hash|yuv|time in.264 [out.yuv|loops]\n", argv[0]); return 2; }
    size_t n; uint8_t *buf = slurp(argv[2], &n);
    if (!strcmp(argv[1], "hash")) {
        decode_stream(buf, n, cb_hash, NULL);
    } else if (!strcmp(argv[1], "yuv") && argc >= 4) {
        FILE *f = fopen(argv[3], "wb"); if (!f) { perror(argv[3]); return 2; }
        decode_stream(buf, n, cb_yuv, f); fclose(f);
    } else if (!strcmp(argv[1], "time")) {
        int loops = argc >= 4 ? atoi(argv[3]) : 1, frames = 0;
        double t0 = now_us();
        for (int l = 0; l < loops; l++) frames += decode_stream(buf, n, NULL, NULL);
        double t1 = now_us();
        printf("frames %d usec %.0f fps %.3f\n", frames, t1 - t0, frames * 1e6 / (t1 - t0));
    } else return 2;
    free(buf);
    return 0;
}
