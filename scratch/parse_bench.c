// parse-only rate of the host bitstream layer on one stream (profiling aid): parse_bench file.264 [repeats]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "p264parse.h"
#include "p264_dropin.h"
int main(int argc, char **argv)
{
    FILE *f = fopen(argv[1], "rb"); if (!f) return 1;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t *buf = malloc(n + 16), *rbsp = malloc(n + 16); if (fread(buf, 1, n, f) != (size_t)n) return 1;
    int reps = argc > 2 ? atoi(argv[2]) : 5, pics = 0;
    struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int r = 0; r < reps; r++) {
        p264parse *p = p264parse_open(1);
        int64_t pos = 0;
        for (;;) {
            int64_t start, len;
            int rc = p264_annexb_next(buf, n, &pos, &start, &len);
            if (rc <= 0) break;
            p264_nal_t nal; nal.p_payload = rbsp;
            p264_nal_decode(&nal, (void *)(buf + start), (int)len);
            const p264hip_picture_t *pic = NULL;
            if (p264parse_nal(p, nal.i_type, nal.i_ref_idc, nal.p_payload, nal.i_payload, &pic) == 1) pics++;
        }
        p264parse_close(p);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    double s = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    printf("%d pictures in %.3f s: %.1f pictures/s\n", pics, s, pics / s);
    return 0;
}
