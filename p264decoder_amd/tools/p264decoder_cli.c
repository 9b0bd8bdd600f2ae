/* p264decoder_cli.c - command-line decoder on the drop-in API.
 *
 * Same interface and observable behaviour as the reference's CLI (p264decoder.c:83-89 usage,
 * :164-381 Decode): `-d <test.264> [recon.yuv] [origin.yuv]`; the stream must begin with a 4-byte
 * start code (:219); every picture the decoder returns is appended to recon.yuv as MB-aligned planar
 * I420 (Y, U, V rows of i_width / i_width/2 bytes, :126-156); frame count and frames/s go to stderr;
 * origin.yuv is opened and never read, as in the reference.  Only p264_dropin.h is used: the program
 * builds unchanged against the reference's p264.h + library.
 *
 * Unlike the reference it maps the whole file instead of sliding a 3 MB window, so NAL units are
 * not limited to 3 000 000 bytes (:48).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "p264_dropin.h"
#include "p264parse.h"

static int usage(void)
{
    fprintf(stderr, "p264 Decoder (MI355X build):\n\n      -d <test.264> [recon.yuv] [origin.yuv]\n");
    return -1;
}

static void append_picture(const p264_picture_t *pic, FILE *out)
{
    for (int p = 0; p < 3; p++) {
        const int w = p ? pic->i_width >> 1 : pic->i_width, h = p ? pic->i_height >> 1 : pic->i_height;
        const uint8_t *row = pic->img.plane[p];
        for (int y = 0; y < h; y++, row += pic->img.i_stride[p]) fwrite(row, 1, (size_t)w, out);
    }
}

int main(int argc, char **argv)
{
    if (argc < 3 || strcmp(argv[1], "-d")) return usage();
    fprintf(stderr, "decoding start...\n");
    FILE *in = fopen(argv[2], "rb");
    if (!in) { fprintf(stderr, "open h264 stream file: %s failed\n", argv[2]); return -1; }
    FILE *rec = argc >= 4 ? fopen(argv[3], "wb") : NULL;
    FILE *ref = argc == 5 ? fopen(argv[4], "rb") : NULL;

    fseek(in, 0, SEEK_END);
    long size = ftell(in);
    fseek(in, 0, SEEK_SET);
    uint8_t *buf = (uint8_t *)malloc(size > 0 ? (size_t)size : 1);
    if (!buf || (size > 0 && fread(buf, 1, (size_t)size, in) != (size_t)size)) { fprintf(stderr, "reading %s failed\n", argv[2]); return -1; }
    fclose(in);

    p264_param_t param;
    p264_param_default(&param);
    p264_t *h = p264_decoder_open(&param);
    if (!h) { fprintf(stderr, "p264_decoder_open failed\n"); return -1; }
    const int64_t t0 = p264_mdate();

    if (size < 4) { fprintf(stderr, "the h264 stream file is too small, even can't include the first start code\n"); return -1; }
    if (buf[0] || buf[1] || buf[2] || buf[3] != 1) { fprintf(stderr, "confirm the first start code failed\n"); return -1; }

    p264_nal_t nal;
    nal.p_payload = (uint8_t *)malloc((size_t)size + 8);
    int frames = 0;
    int64_t pos = 0, off = 0, len = 0;
    while (p264_annexb_next(buf, size, &pos, &off, &len)) {
        p264_picture_t *pic = NULL;
        p264_nal_decode(&nal, buf + off, (int)len);
        p264_decoder_decode(h, &pic, &nal);
        if (pic) {
            frames++;
            if (rec) append_picture(pic, rec);
        }
    }
    const int64_t t1 = p264_mdate();
    if (frames > 0) {
        fprintf(stderr, "decoded total %d frames \n", frames);
        fprintf(stderr, "decoding speed: %.2f fps\n", (double)frames * 1e6 / (double)(t1 - t0));
    }
    p264_decoder_close(h);
    if (rec) fclose(rec);
    if (ref) fclose(ref);
    free(nal.p_payload);
    free(buf);
    return 0;
}
