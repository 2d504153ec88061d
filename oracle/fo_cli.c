/*
 * fo_cli.c -- ORACLE command line (test infrastructure): generate synthetic
 * frames, encode with the CPU restatement, decode to the reference's Y4M
 * layout (F/fileIO.cpp:134-176).  Also the timed CPU baseline of bench.py.
 */
#include "fo.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

typedef struct {
    FILE *f;
    int first;
} y4m_out;

static void y4m_cb(fo_ctx *c, void *user)
{
    y4m_out *o = (y4m_out *)user;
    if (o->first) {
        fprintf(o->f, "YUV4MPEG2 C420jpeg W%d H%d F24:1 Ip A1:1%c", c->W, c->H, 0x0a);
        o->first = 0;
    }
    fprintf(o->f, "FRAME%c", 0x0a);
    fwrite(c->L, 1, (size_t)c->W * c->H, o->f);
    fwrite(c->C[0], 1, (size_t)c->Wc * c->Hc, o->f);
    fwrite(c->C[1], 1, (size_t)c->Wc * c->Hc, o->f);
}

int main(int argc, char **argv)
{
    if (argc >= 8 && !strcmp(argv[1], "gen")) {
        int W = atoi(argv[2]), H = atoi(argv[3]), n = atoi(argv[4]);
        uint64_t seed = strtoull(argv[5], 0, 10);
        int A = atoi(argv[6]);
        FILE *f = fopen(argv[7], "wb");
        uint8_t *b = malloc((size_t)W * H * 3 / 2);
        for (int t = 0; t < n; t++) {
            fo_gen_frame(W, H, t, seed, A, b, b + W * H, b + W * H + W * H / 4);
            fwrite(b, 1, (size_t)W * H * 3 / 2, f);
        }
        fclose(f);
        return 0;
    }
    if (argc >= 12 && !strcmp(argv[1], "enc")) {
        int W = atoi(argv[2]), H = atoi(argv[3]), n = atoi(argv[4]);
        int qp = atoi(argv[5]), window = atoi(argv[6]), maxdiff = atoi(argv[7]), ie = atoi(argv[8]), basic = atoi(argv[9]);
        size_t fsz = (size_t)W * H * 3 / 2;
        uint8_t *in = malloc(fsz * n), *out = malloc(fsz * n * 2 + 65536), *rec = malloc(fsz * n);
        FILE *f = fopen(argv[10], "rb");
        if (!f || fread(in, 1, fsz * n, f) != fsz * n) {
            fprintf(stderr, "short input\n");
            return 1;
        }
        fclose(f);
        fo_ctx *c = fo_create(W, H);
        fo_set_params(c, qp, basic, window, maxdiff, ie);
        double t0 = now();
        size_t m = fo_encode_stream(c, in, n, out, fsz * n * 2 + 65536, rec);
        double t1 = now();
        f = fopen(argv[11], "wb");
        fwrite(out, 1, m, f);
        fclose(f);
        if (argc >= 13) {
            f = fopen(argv[12], "wb");
            fwrite(rec, 1, fsz * n, f);
            fclose(f);
        }
        printf("{\"bytes\": %zu, \"seconds\": %.6f, \"mbs\": %d, \"mb_per_s\": %.1f, \"types\": [%d,%d,%d,%d,%d]}\n", m,
               t1 - t0, c->nmb * n, c->nmb * n / (t1 - t0), c->type_count[0], c->type_count[1], c->type_count[2],
               c->type_count[3], c->type_count[4]);
        return 0;
    }
    if (argc >= 4 && !strcmp(argv[1], "dec")) {
        FILE *f = fopen(argv[2], "rb");
        if (!f) return 1;
        fseek(f, 0, SEEK_END);
        long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        uint8_t *s = malloc((size_t)sz);
        if (fread(s, 1, (size_t)sz, f) != (size_t)sz) return 1;
        fclose(f);
        y4m_out o = {fopen(argv[3], "wb"), 1};
        double t0 = now();
        int n = fo_decode_stream(s, (size_t)sz, y4m_cb, &o, NULL);
        double t1 = now();
        fclose(o.f);
        printf("{\"pictures\": %d, \"seconds\": %.6f}\n", n, t1 - t0);
        return 0;
    }
    fprintf(stderr,
            "usage: fo_cli gen W H n seed noise out.yuv | enc W H n qp window maxdiff intraEvery basic in.yuv out.264 "
            "[recon.yuv] | dec in.264 out.y4m\n");
    return 2;
}
