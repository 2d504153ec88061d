/*
 * fo_gen.c -- ORACLE-side copy of the deterministic synthetic 4:2:0 source
 * (test infrastructure).  Integer-only so that this C code and the numpy
 * version in h264-fer_amd/synth.py agree bit for bit.  Spec:
 *   lcg(s)   = s*6364136223846793005 + 1442695040888963407 (mod 2^64)
 *   tex[r][q] (64x64) = ((lcg chain from seed) >> 33) % 25 - 12
 *   noise(x,y,t): s = (seed ^ (0x9E3779B97F4A7C15*(t+1))) + y*W + x;
 *                 s = lcg(s); s ^= s>>29; s = lcg(s); n = (s>>33) % (2A+1) - A
 *   Y = clip(16,235, 40 + tri(x+2t,192) + tri(y+t,128) + tex[(y+t)&63][(x+2t)&63] + n)
 *   U = 104 + tri(xc+t,96)/2,  V = 104 + tri(yc+t,96)/2
 *   tri(v,p) = (m = v mod p) < p/2 ? m : p-m
 */
#include "fo.h"

static inline uint64_t lcg(uint64_t s) { return s * 6364136223846793005ULL + 1442695040888963407ULL; }
static inline int tri(int v, int p)
{
    int m = v % p;
    return m < p / 2 ? m : p - m;
}

void fo_gen_frame(int W, int H, int t, uint64_t seed, int A, uint8_t *Y, uint8_t *U, uint8_t *V)
{
    int tex[64][64];
    uint64_t s = seed;
    for (int r = 0; r < 64; r++)
        for (int q = 0; q < 64; q++) {
            s = lcg(s);
            tex[r][q] = (int)((s >> 33) % 25) - 12;
        }
    uint64_t base = seed ^ (0x9E3779B97F4A7C15ULL * (uint64_t)(t + 1));
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            uint64_t z = base + (uint64_t)y * (uint64_t)W + (uint64_t)x;
            z = lcg(z);
            z ^= z >> 29;
            z = lcg(z);
            int n = A > 0 ? (int)((z >> 33) % (uint64_t)(2 * A + 1)) - A : 0;
            int v = 40 + tri(x + 2 * t, 192) + tri(y + t, 128) + tex[(y + t) & 63][(x + 2 * t) & 63] + n;
            Y[y * W + x] = (uint8_t)(v < 16 ? 16 : (v > 235 ? 235 : v));
        }
    for (int y = 0; y < H / 2; y++)
        for (int x = 0; x < W / 2; x++) {
            U[y * (W / 2) + x] = (uint8_t)(104 + tri(x + t, 96) / 2);
            V[y * (W / 2) + x] = (uint8_t)(104 + tri(y + t, 96) / 2);
        }
}
