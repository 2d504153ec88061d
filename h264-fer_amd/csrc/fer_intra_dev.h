// fer_intra_dev.h -- intra prediction device helpers shared by the encoder's mode decision
// (fer_intra.hip) and the decoder's reconstruction (fer_decode.hip): the nine Intra4x4
// predictors (F/intra.cpp:140-292), neighbour fetch (F/intra.cpp:294-378), Intra16x16 and chroma
// prediction samples (F/intra.cpp:426-498,568-687) over an LDS window of the macroblock.
#pragma once
#include "fer_dev.h"

#define P4(x, y) (((x) == -1) ? p[(y) + 1] : p[(x) + 5])

// nine Intra4x4 predictors, F/intra.cpp:140-292; p = corner, 4 left, 8 top
__device__ void pred4x4(int mode, const int p[13], int o[16])
{
    switch (mode) {
    case 0:
        for (int i = 0; i < 16; i++) o[i] = p[5 + (i & 3)];
        break;
    case 1:
        for (int i = 0; i < 16; i++) o[i] = p[1 + (i >> 2)];
        break;
    case 2: {
        int r = 128;
        if (p[0] != -1)
            r = (p[5] + p[6] + p[7] + p[8] + p[1] + p[2] + p[3] + p[4] + 4) >> 3;
        else if (p[1] != -1)
            r = (p[1] + p[2] + p[3] + p[4] + 2) >> 2;
        else if (p[5] != -1)
            r = (p[5] + p[6] + p[7] + p[8] + 2) >> 2;
        for (int i = 0; i < 16; i++) o[i] = r;
        break;
    }
    case 3:
        for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++)
                o[y * 4 + x] = (x == 3 && y == 3) ? (p[11] + 3 * p[12] + 2) >> 2
                                                  : (p[5 + x + y] + 2 * p[6 + x + y] + p[7 + x + y] + 2) >> 2;
        break;
    case 4:
        for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++) {
                if (x > y)
                    o[y * 4 + x] = (P4(x - y - 2, -1) + 2 * P4(x - y - 1, -1) + P4(x - y, -1) + 2) >> 2;
                else if (x < y)
                    o[y * 4 + x] = (P4(-1, y - x - 2) + 2 * P4(-1, y - x - 1) + P4(-1, y - x) + 2) >> 2;
                else
                    o[y * 4 + x] = (P4(0, -1) + 2 * P4(-1, -1) + P4(-1, 0) + 2) >> 2;
            }
        break;
    case 5:
        for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++) {
                int z = 2 * x - y, v;
                if (z >= 0 && (z & 1) == 0)
                    v = (P4(x - (y >> 1) - 1, -1) + P4(x - (y >> 1), -1) + 1) >> 1;
                else if (z >= 0)
                    v = (P4(x - (y >> 1) - 2, -1) + 2 * P4(x - (y >> 1) - 1, -1) + P4(x - (y >> 1), -1) + 2) >> 2;
                else if (z == -1)
                    v = (P4(-1, 0) + 2 * P4(-1, -1) + P4(0, -1) + 2) >> 2;
                else
                    v = (P4(-1, y - 1) + 2 * P4(-1, y - 2) + P4(-1, y - 3) + 2) >> 2;
                o[y * 4 + x] = v;
            }
        break;
    case 6:
        for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++) {
                int z = 2 * y - x, v;
                if (z >= 0 && (z & 1) == 0)
                    v = (P4(-1, y - (x >> 1) - 1) + P4(-1, y - (x >> 1)) + 1) >> 1;
                else if (z >= 0)
                    v = (P4(-1, y - (x >> 1) - 2) + 2 * P4(-1, y - (x >> 1) - 1) + P4(-1, y - (x >> 1)) + 2) >> 2;
                else if (z == -1)
                    v = (P4(-1, 0) + 2 * P4(-1, -1) + P4(0, -1) + 2) >> 2;
                else
                    v = (P4(x - 1, -1) + 2 * P4(x - 2, -1) + P4(x - 3, -1) + 2) >> 2;
                o[y * 4 + x] = v;
            }
        break;
    case 7:
        for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++) {
                int b = 5 + x + (y >> 1);
                o[y * 4 + x] = (y & 1) ? (p[b] + 2 * p[b + 1] + p[b + 2] + 2) >> 2 : (p[b] + p[b + 1] + 1) >> 1;
            }
        break;
    default:
        for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++) {
                int z = x + 2 * y, v;
                if (z > 5)
                    v = p[4];
                else if (z == 5)
                    v = (p[3] + 3 * p[4] + 2) >> 2;
                else if ((z & 1) == 0)
                    v = (P4(-1, y + (x >> 1)) + P4(-1, y + (x >> 1) + 1) + 1) >> 1;
                else
                    v = (P4(-1, y + (x >> 1)) + 2 * P4(-1, y + (x >> 1) + 1) + P4(-1, y + (x >> 1) + 2) + 2) >> 2;
                o[y * 4 + x] = v;
            }
        break;
    }
}

// LDS picture window of the macroblock: row 0 / column 0 hold the neighbours.
//   fr[0][0] corner, fr[0][1..16] top, fr[0][17..20] top-right, fr[1..16][0] left, fr[1..16][1..16] MB
struct IntraLds {
    int16_t fr[17][24];     // -1 = unavailable
    int16_t cfr[2][9][12];  // chroma: same layout, 8x8
    uint8_t predC[2][8][8];
    int16_t lv4[16][16];    // Intra4x4 luma levels
    int16_t lv16[16][16];   // Intra16x16 AC levels (15 used)
    int16_t dc16[16];
    int16_t cdc[2][4], cac[2][4][16];
    int dcraw[16];
    int dcdeq[16];
    int key4[144];
    uint8_t tc16[16], tc4[16], tcc[2][4];
    uint8_t mode4[16], flag4[16];
};

// fetch p[13] of block blk from the window (F/intra.cpp:294-378)
__device__ void fetch4(const IntraLds &L, int blk, bool lastcol, int p[13])
{
    int x0 = c_bx[blk], y0 = c_by[blk];
    p[0] = L.fr[y0][x0];
    for (int i = 0; i < 4; i++) p[1 + i] = L.fr[y0 + 1 + i][x0];
    for (int i = 0; i < 4; i++) p[5 + i] = L.fr[y0][x0 + 1 + i];
    if (p[5] == -1) {
        for (int i = 9; i < 13; i++) p[i] = -1;
    } else {
        bool edge = (x0 == 12 && lastcol) || (x0 == 12 && y0 > 0);
        if (edge || blk == 3 || blk == 11)
            for (int i = 9; i < 13; i++) p[i] = p[8];
        else
            for (int i = 9; i < 13; i++) p[i] = L.fr[y0][x0 + 5 + i - 9];
    }
}

__device__ __forceinline__ bool mode4_avail(int m, const int p[13])
{
    if ((m == 0 || m == 3 || m == 7) && p[5] == -1) return false;
    if ((m == 1 || m == 8) && p[1] == -1) return false;
    if ((m == 4 || m == 5 || m == 6) && p[0] == -1) return false;
    return true;
}

// Intra16x16 prediction sample, F/intra.cpp:426-498
struct P16 {
    int dc, a, b, c;
};
__device__ __forceinline__ int pred16_px(const IntraLds &L, const P16 &q, int mode, int x, int y)
{
    if (mode == 0) return L.fr[0][1 + x];
    if (mode == 1) return L.fr[1 + y][0];
    if (mode == 2) return q.dc;
    return clip255((q.a + q.b * (x - 7) + q.c * (y - 7) + 16) >> 5);
}


// DC value and plane parameters of Intra16x16 prediction from the window
__device__ __forceinline__ void pred16_params(const IntraLds &L, bool availL, bool availT, P16 &q)
{
    int sx = 0, sy = 0, Hh = 0, V = 0;
    for (int i = 0; i < 16; i++) {
        sx += L.fr[0][1 + i];
        sy += L.fr[1 + i][0];
    }
    q.dc = 128;
    if (availL && availT)
        q.dc = (sx + sy + 16) >> 5;
    else if (availL)
        q.dc = (sy + 8) >> 4;
    else if (availT)
        q.dc = (sx + 8) >> 4;
    for (int i = 0; i <= 7; i++) {
        Hh += (i + 1) * (L.fr[0][1 + 8 + i] - L.fr[0][1 + 6 - i]);  // p(6-i,-1): i = 7 -> corner fr[0][0]
        V += (i + 1) * (L.fr[1 + 8 + i][0] - L.fr[1 + 6 - i][0]);
    }
    q.a = (L.fr[16][0] + L.fr[0][16]) << 4;
    q.b = (5 * Hh + 32) >> 6;
    q.c = (5 * V + 32) >> 6;
}

// one chroma prediction sample (F/intra.cpp:568-687); cf = window of the plane
__device__ __forceinline__ int pred_chroma_px(const int16_t (*cf)[12], int chroma_mode, int cx, int cy, bool availL,
                                              bool availT)
{
    if (chroma_mode == 1) return cf[1 + cy][0];
    if (chroma_mode == 2) return cf[0][1 + cx];
    if (chroma_mode == 3) {
        int Hh = 0, V = 0;
        for (int i = 0; i <= 3; i++) {
            Hh += (i + 1) * (cf[0][1 + 4 + i] - cf[0][1 + 2 - i]);
            V += (i + 1) * (cf[1 + 4 + i][0] - cf[1 + 2 - i][0]);
        }
        int a = (cf[8][0] + cf[0][8]) << 4, b = (34 * Hh + 32) >> 6, c = (34 * V + 32) >> 6;
        return clip255((a + b * (cx - 3) + c * (cy - 3) + 16) >> 5);
    }
    int x0 = cx & 4, y0 = cy & 4;
    int sx = 0, sy = 0;
    for (int i = 0; i < 4; i++) {
        sx += cf[0][1 + x0 + i];
        sy += cf[1 + y0 + i][0];
    }
    int v = 128;
    if ((x0 == 0 && y0 == 0) || (x0 > 0 && y0 > 0)) {
        if (availT && availL)
            v = (sx + sy + 4) >> 3;
        else if (availL)
            v = (sy + 2) >> 2;
        else if (availT)
            v = (sx + 2) >> 2;
    } else if (x0 > 0 && y0 == 0) {
        if (availT)
            v = (sx + 2) >> 2;
        else if (availL)
            v = (sy + 2) >> 2;
    } else {
        if (availL)
            v = (sy + 2) >> 2;
        else if (availT)
            v = (sx + 2) >> 2;
    }
    return v;
}
