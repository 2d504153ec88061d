/*
 * fo_intra.c -- ORACLE (test infrastructure): rows a10, a11 (and the decoder's
 * intraPrediction) of SURVEY.md 8a.  Reference: F/intra.cpp.
 *
 * Neighbour sample vectors use the reference's layouts:
 *   4x4  : p[0]=p(-1,-1), p[1..4]=p(-1,0..3), p[5..12]=p(0..7,-1)      (F/intra.cpp:138,294)
 *   16x16: p[0]=corner,   p[1..16]=left,      p[17..32]=top            (F/intra.cpp:424,500)
 *   chroma: p[0]=corner,  p[1..8]=left,       p[9..16]=top             (F/intra.cpp:565,690)
 * Unavailable samples are -1.
 */
#include "fo.h"
#include <limits.h>
#include <string.h>

int fo_pred_class(const fo_ctx *c, int mb_type);

static inline int clip255(int x) { return x < 0 ? 0 : (x > 255 ? 255 : x); }

#define P4(x, y) (((x) == -1) ? p[(y) + 1] : p[(x) + 5])

/* F/intra.cpp:140-292 */
void fo_intra4x4_pred(int mode, const int p[14], int pred[4][4])
{
    int x, y;
    switch (mode) {
    case 0:
        for (y = 0; y < 4; y++)
            for (x = 0; x < 4; x++) pred[y][x] = P4(x, -1);
        break;
    case 1:
        for (y = 0; y < 4; y++)
            for (x = 0; x < 4; x++) pred[y][x] = P4(-1, y);
        break;
    case 2: {
        int r = 128;
        if (P4(-1, -1) != -1)
            r = (P4(0, -1) + P4(1, -1) + P4(2, -1) + P4(3, -1) + P4(-1, 0) + P4(-1, 1) + P4(-1, 2) + P4(-1, 3) + 4) >> 3;
        else if (P4(-1, 0) != -1)
            r = (P4(-1, 0) + P4(-1, 1) + P4(-1, 2) + P4(-1, 3) + 2) >> 2;
        else if (P4(0, -1) != -1)
            r = (P4(0, -1) + P4(1, -1) + P4(2, -1) + P4(3, -1) + 2) >> 2;
        for (y = 0; y < 4; y++)
            for (x = 0; x < 4; x++) pred[y][x] = r;
        break;
    }
    case 3:
        for (y = 0; y < 4; y++)
            for (x = 0; x < 4; x++) {
                if (x == 3 && y == 3)
                    pred[y][x] = (P4(6, -1) + 3 * P4(7, -1) + 2) >> 2;
                else
                    pred[y][x] = (P4(x + y, -1) + (P4(x + y + 1, -1) << 1) + P4(x + y + 2, -1) + 2) >> 2;
            }
        break;
    case 4:
        for (y = 0; y < 4; y++)
            for (x = 0; x < 4; x++) {
                if (x > y)
                    pred[y][x] = (P4(x - y - 2, -1) + (P4(x - y - 1, -1) << 1) + P4(x - y, -1) + 2) >> 2;
                else if (x < y)
                    pred[y][x] = (P4(-1, y - x - 2) + (P4(-1, y - x - 1) << 1) + P4(-1, y - x) + 2) >> 2;
                else
                    pred[y][x] = (P4(0, -1) + (P4(-1, -1) << 1) + P4(-1, 0) + 2) >> 2;
            }
        break;
    case 5:
        for (y = 0; y < 4; y++)
            for (x = 0; x < 4; x++) {
                int z = (x << 1) - y;
                if (z == 0 || z == 2 || z == 4 || z == 6)
                    pred[y][x] = (P4(x - (y >> 1) - 1, -1) + P4(x - (y >> 1), -1) + 1) >> 1;
                else if (z == 1 || z == 3 || z == 5)
                    pred[y][x] =
                        (P4(x - (y >> 1) - 2, -1) + (P4(x - (y >> 1) - 1, -1) << 1) + P4(x - (y >> 1), -1) + 2) >> 2;
                else if (z == -1)
                    pred[y][x] = (P4(-1, 0) + (P4(-1, -1) << 1) + P4(0, -1) + 2) >> 2;
                else
                    pred[y][x] = (P4(-1, y - 1) + (P4(-1, y - 2) << 1) + P4(-1, y - 3) + 2) >> 2;
            }
        break;
    case 6:
        for (y = 0; y < 4; y++)
            for (x = 0; x < 4; x++) {
                int z = (y << 1) - x;
                if (z == 0 || z == 2 || z == 4 || z == 6)
                    pred[y][x] = (P4(-1, y - (x >> 1) - 1) + P4(-1, y - (x >> 1)) + 1) >> 1;
                else if (z == 1 || z == 3 || z == 5)
                    pred[y][x] =
                        (P4(-1, y - (x >> 1) - 2) + (P4(-1, y - (x >> 1) - 1) << 1) + P4(-1, y - (x >> 1)) + 2) >> 2;
                else if (z == -1)
                    pred[y][x] = (P4(-1, 0) + (P4(-1, -1) << 1) + P4(0, -1) + 2) >> 2;
                else
                    pred[y][x] = (P4(x - 1, -1) + (P4(x - 2, -1) << 1) + P4(x - 3, -1) + 2) >> 2;
            }
        break;
    case 7:
        for (y = 0; y < 4; y++)
            for (x = 0; x < 4; x++) {
                if (y == 0 || y == 2)
                    pred[y][x] = (P4(x + (y >> 1), -1) + P4(x + (y >> 1) + 1, -1) + 1) >> 1;
                else
                    pred[y][x] =
                        (P4(x + (y >> 1), -1) + (P4(x + (y >> 1) + 1, -1) << 1) + P4(x + (y >> 1) + 2, -1) + 2) >> 2;
            }
        break;
    case 8:
        for (y = 0; y < 4; y++)
            for (x = 0; x < 4; x++) {
                int z = x + (y << 1);
                if (z == 0 || z == 2 || z == 4)
                    pred[y][x] = (P4(-1, y + (x >> 1)) + P4(-1, y + (x >> 1) + 1) + 1) >> 1;
                else if (z == 1 || z == 3)
                    pred[y][x] =
                        (P4(-1, y + (x >> 1)) + (P4(-1, y + (x >> 1) + 1) << 1) + P4(-1, y + (x >> 1) + 2) + 2) >> 2;
                else if (z == 5)
                    pred[y][x] = (P4(-1, 2) + 3 * P4(-1, 3) + 2) >> 2;
                else
                    pred[y][x] = P4(-1, 3);
            }
        break;
    }
}

/* F/intra.cpp:294-378 */
void fo_intra4x4_fetch(fo_ctx *c, int blk, int p[14])
{
    int xP = (c->cur % c->mbw) << 4, yP = (c->cur / c->mbw) << 4;
    int x0 = fo_blk_xy[blk][0], y0 = fo_blk_xy[blk][1];
    int x = xP + x0, y = yP + y0;
    const uint8_t *L = c->L;
    int W = c->W;
    p[13] = 0;
    p[0] = (x - 1 < 0 || y - 1 < 0) ? -1 : L[(y - 1) * W + x - 1];
    for (int i = 1; i < 5; i++) p[i] = (x - 1 < 0) ? -1 : L[(y + i - 1) * W + x - 1];
    if (y - 1 < 0) {
        for (int i = 5; i < 13; i++) p[i] = -1;
    } else {
        for (int i = 5; i < 9; i++) p[i] = L[(y - 1) * W + x + i - 5];
        int edge = (x + 4 >= W) || (x0 == 12 && y0 > 0);
        if (edge || blk == 3 || blk == 11) {
            for (int i = 9; i < 13; i++) p[i] = L[(y - 1) * W + x + 3];
        } else {
            for (int i = 9; i < 13; i++) p[i] = L[(y - 1) * W + x + 4 + i - 9];
        }
    }
}

#define P16(x, y) (((x) == -1) ? p[(y) + 1] : p[(x) + 17])

/* F/intra.cpp:426-498 */
void fo_intra16_pred(int mode, const int p[33], int pred[16][16])
{
    int x, y;
    switch (mode) {
    case 0:
        for (y = 0; y < 16; y++)
            for (x = 0; x < 16; x++) pred[y][x] = P16(x, -1);
        break;
    case 1:
        for (y = 0; y < 16; y++)
            for (x = 0; x < 16; x++) pred[y][x] = P16(-1, y);
        break;
    case 2: {
        int sx = 0, sy = 0;
        for (int i = 0; i < 16; i++) {
            sx += P16(i, -1);
            sy += P16(-1, i);
        }
        int r = 128;
        if (p[0] != -1)
            r = (sx + sy + 16) >> 5;
        else if (p[1] != -1)
            r = (sy + 8) >> 4;
        else if (p[17] != -1)
            r = (sx + 8) >> 4;
        for (y = 0; y < 16; y++)
            for (x = 0; x < 16; x++) pred[y][x] = r;
        break;
    }
    case 3: {
        int Hh = 0, V = 0;
        for (int i = 0; i <= 7; i++) {
            Hh += (i + 1) * (P16(8 + i, -1) - P16(6 - i, -1));
            V += (i + 1) * (P16(-1, 8 + i) - P16(-1, 6 - i));
        }
        int a = (P16(-1, 15) + P16(15, -1)) << 4;
        int b = (5 * Hh + 32) >> 6;
        int cc = (5 * V + 32) >> 6;
        for (y = 0; y < 16; y++)
            for (x = 0; x < 16; x++) pred[y][x] = clip255((a + b * (x - 7) + cc * (y - 7) + 16) >> 5);
        break;
    }
    }
}

/* F/intra.cpp:500-533 */
void fo_intra16_fetch(fo_ctx *c, int p[33])
{
    int xP = (c->cur % c->mbw) << 4, yP = (c->cur / c->mbw) << 4;
    const uint8_t *L = c->L;
    int W = c->W;
    p[0] = (xP - 1 >= 0 && yP - 1 >= 0) ? L[(yP - 1) * W + xP - 1] : -1;
    for (int i = 1; i < 17; i++) p[i] = (xP - 1 >= 0) ? L[(yP + i - 1) * W + xP - 1] : -1;
    for (int i = 17; i < 33; i++) p[i] = (yP - 1 >= 0) ? L[(yP - 1) * W + xP + i - 17] : -1;
}

#define PC(x, y) (((x) == -1) ? p[(y) + 1] : p[(x) + 9])

/* F/intra.cpp:568-687 (the transposed store of blocks 0 and 3 at :603 fills a
 * constant, so it is the same as the plain store). */
static void chroma_pred(int mode, const int p[17], int pred[8][8])
{
    int x, y;
    switch (mode) {
    case 0:
        for (int blk = 0; blk < 4; blk++) {
            int x0 = (blk & 1) << 2, y0 = (blk >> 1) << 2;
            int sx = 0, sy = 0;
            for (int i = 0; i < 4; i++) {
                sx += PC(i + x0, -1);
                sy += PC(-1, i + y0);
            }
            int left = PC(-1, y0) != -1, top = PC(x0, -1) != -1;
            int r = 128;
            if ((x0 == 0 && y0 == 0) || (x0 > 0 && y0 > 0)) {
                if (top && left)
                    r = (sx + sy + 4) >> 3;
                else if (left)
                    r = (sy + 2) >> 2;
                else if (top)
                    r = (sx + 2) >> 2;
            } else if (x0 > 0 && y0 == 0) {
                if (top)
                    r = (sx + 2) >> 2;
                else if (left)
                    r = (sy + 2) >> 2;
            } else {
                if (left)
                    r = (sy + 2) >> 2;
                else if (top)
                    r = (sx + 2) >> 2;
            }
            for (y = 0; y < 4; y++)
                for (x = 0; x < 4; x++) pred[y + y0][x + x0] = r;
        }
        break;
    case 1:
        for (y = 0; y < 8; y++)
            for (x = 0; x < 8; x++) pred[y][x] = PC(-1, y);
        break;
    case 2:
        for (y = 0; y < 8; y++)
            for (x = 0; x < 8; x++) pred[y][x] = PC(x, -1);
        break;
    case 3: {
        int Hh = 0, V = 0;
        for (int i = 0; i <= 3; i++) Hh += (i + 1) * (PC(4 + i, -1) - PC(2 - i, -1));
        for (int i = 0; i <= 3; i++) V += (i + 1) * (PC(-1, 4 + i) - PC(-1, 2 - i));
        int a = (PC(-1, 7) + PC(7, -1)) << 4;
        int b = (34 * Hh + 32) >> 6;
        int cc = (34 * V + 32) >> 6;
        for (y = 0; y < 8; y++)
            for (x = 0; x < 8; x++) pred[y][x] = clip255((a + b * (x - 3) + cc * (y - 3) + 16) >> 5);
        break;
    }
    }
}

/* F/intra.cpp:690-767; uses c->chroma_mode (intra_chroma_pred_mode) */
void fo_intra_chroma(fo_ctx *c, int predCr[8][8], int predCb[8][8])
{
    int pb[17], pr[17];
    int xM = (c->cur % c->mbw) << 3, yM = (c->cur / c->mbw) << 3;
    int Wc = c->Wc;
    if (xM - 1 < 0 || yM - 1 < 0) {
        pb[0] = pr[0] = -1;
    } else {
        pb[0] = c->C[0][(yM - 1) * Wc + xM - 1];
        pr[0] = c->C[1][(yM - 1) * Wc + xM - 1];
    }
    for (int i = 1; i < 9; i++) {
        if (xM - 1 < 0) {
            pb[i] = pr[i] = -1;
        } else {
            pb[i] = c->C[0][(yM + i - 1) * Wc + xM - 1];
            pr[i] = c->C[1][(yM + i - 1) * Wc + xM - 1];
        }
    }
    for (int i = 9; i < 17; i++) {
        if (yM - 1 < 0) {
            pb[i] = pr[i] = -1;
        } else {
            pb[i] = c->C[0][(yM - 1) * Wc + xM + i - 9];
            pr[i] = c->C[1][(yM - 1) * Wc + xM + i - 9];
        }
    }
    chroma_pred(c->chroma_mode, pb, predCb);
    chroma_pred(c->chroma_mode, pr, predCr);
}

/* F/intra.cpp:27-74 (6.4.10.4): neighbour A (left) / B (up) of a 4x4 luma block */
static const int nbr4[16][2] = {{5, 10}, {0, 11}, {7, 0},  {2, 1},  {1, 14}, {4, 15}, {3, 4},   {6, 5},
                                {13, 2}, {8, 3},  {15, 8}, {10, 9}, {9, 6},  {12, 7}, {11, 12}, {14, 13}};

static void nbr_addr(const fo_ctx *c, int blk, int isA, int *mb, int *nblk)
{
    if (isA) {
        if (blk == 0 || blk == 2 || blk == 8 || blk == 10) {
            if (c->cur % c->mbw == 0) {
                *mb = -1;
                *nblk = -1;
            } else {
                *mb = c->cur - 1;
                *nblk = nbr4[blk][0];
            }
        } else {
            *mb = c->cur;
            *nblk = nbr4[blk][0];
        }
    } else {
        if (blk == 0 || blk == 1 || blk == 4 || blk == 5) {
            if (c->cur < c->mbw) {
                *mb = -1;
                *nblk = -1;
            } else {
                *mb = c->cur - c->mbw;
                *nblk = nbr4[blk][1];
            }
        } else {
            *mb = c->cur;
            *nblk = nbr4[blk][1];
        }
    }
}

/* predIntra4x4PredMode of F/intra.cpp:77-136 / :878-942 */
static int pred_i4_mode(const fo_ctx *c, int blk)
{
    int mbA, mbB, bA, bB;
    nbr_addr(c, blk, 1, &mbA, &bA);
    nbr_addr(c, blk, 0, &mbB, &bB);
    int mA, mB;
    if (mbA == -1 || mbB == -1 || c->constrained_intra == 1) {
        mA = mB = 2;
    } else {
        mA = (fo_pred_class(c, c->mb_type[mbA]) != 0) ? 2 : c->i4mode[(mbA << 4) + bA];
        mB = (fo_pred_class(c, c->mb_type[mbB]) != 0) ? 2 : c->i4mode[(mbB << 4) + bB];
    }
    return (mA <= mB) ? mA : mB;
}

/* decoder: getIntra4x4PredMode F/intra.cpp:77-136 */
static void get_i4_mode(fo_ctx *c, int blk)
{
    int pm = pred_i4_mode(c, blk);
    int idx = (c->cur << 4) + blk;
    if (c->prev_flag[blk])
        c->i4mode[idx] = pm;
    else
        c->i4mode[idx] = (c->rem_mode[blk] < pm) ? c->rem_mode[blk] : c->rem_mode[blk] + 1;
}

/* encoder: setIntra4x4PredMode F/intra.cpp:878-942 */
static void set_i4_mode(fo_ctx *c, int blk)
{
    int pm = pred_i4_mode(c, blk);
    int m = c->i4mode[(c->cur << 4) + blk];
    if (m == pm) {
        c->prev_flag[blk] = 1;
    } else {
        c->prev_flag[blk] = 0;
        c->rem_mode[blk] = (m < pm) ? m : m - 1;
    }
}

/* decoder: intraPrediction F/intra.cpp:770-812 */
void fo_intraPrediction_dec(fo_ctx *c, int predL[16][16], int predCr[8][8], int predCb[8][8])
{
    if (fo_pred_class(c, c->cur_mb_type) == 0) {
        for (int blk = 0; blk < 16; blk++) {
            get_i4_mode(c, blk);
            int p[14], pr[4][4];
            fo_intra4x4_fetch(c, blk, p);
            fo_intra4x4_pred(c->i4mode[(c->cur << 4) + blk], p, pr);
            int x0 = fo_blk_xy[blk][0], y0 = fo_blk_xy[blk][1];
            for (int y = 0; y < 4; y++)
                for (int x = 0; x < 4; x++) predL[y0 + y][x0 + x] = pr[y][x];
            fo_transformDecoding4x4Luma(c, c->lv.Lumalevel, predL, blk, c->QPy);
        }
    } else {
        int t = c->cur_mb_type;
        if (c->slice_type == 0) t -= 5;
        int mode = (t - 1) & 3; /* I_Macroblock_Modes[t][4] */
        int p[33];
        fo_intra16_fetch(c, p);
        fo_intra16_pred(mode, p, predL);
    }
    fo_intra_chroma(c, predCr, predCb);
}

/* Sum of |quantised coefficients| of one predicted 4x4 block: F/intra.cpp:819-850 */
static int cost4x4(fo_ctx *c, int pr[4][4], int blk)
{
    int xP = (c->cur % c->mbw) << 4, yP = (c->cur / c->mbw) << 4;
    int x0 = fo_blk_xy[blk][0], y0 = fo_blk_xy[blk][1];
    int d[4][4], r[4][4], s = 0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) d[i][j] = c->L[(yP + y0 + i) * c->W + xP + x0 + j] - pr[i][j];
    fo_forwardResidual(c->QPy, d, r, 0);
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) s += r[i][j] < 0 ? -r[i][j] : r[i][j];
    return s;
}

static int cost16x16(fo_ctx *c, int predL[16][16])
{
    int s = 0;
    for (int blk = 0; blk < 16; blk++) {
        int x0 = fo_blk_xy[blk][0], y0 = fo_blk_xy[blk][1];
        int pr[4][4];
        for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++) pr[y][x] = predL[y0 + y][x0 + x];
        s += cost4x4(c, pr, blk);
    }
    return s;
}

/* F/intra.cpp:949-1110, CPU path (OpenCLEnabled == false) */
int fo_intraPredictionEncoding(fo_ctx *c, int predL[16][16], int predCr[8][8], int predCb[8][8])
{
    static const int intraToChroma[4] = {2, 1, 0, 3};
    int mode16 = 0, chosenChroma = 0;
    int p[33];
    fo_intra16_fetch(c, p);
    int min16 = INT_MAX;
    for (int i = 0; i < 4; i++) {
        if ((i == 0 && p[17] == -1) || (i == 1 && p[1] == -1) || (i == 3 && p[0] == -1)) continue;
        fo_intra16_pred(i, p, predL);
        int s = cost16x16(c, predL);
        if (s < min16) {
            min16 = s;
            mode16 = i;
            chosenChroma = intraToChroma[i];
        }
    }
    fo_intra16_pred(mode16, p, predL);
    c->chroma_mode = chosenChroma;
    fo_intra_chroma(c, predCr, predCb);
    unsigned min = fo_coded_mb_size(c, mode16, predL, predCb, predCr);

    c->mb_type[c->cur] = 0;
    for (int blk = 0; blk < 16; blk++) {
        int min4 = INT_MAX;
        int q[14], pr[4][4];
        fo_intra4x4_fetch(c, blk, q);
        for (int m = 0; m < 9; m++) {
            if ((m == 0 && q[5] == -1) || (m == 1 && q[1] == -1) || (m == 3 && q[5] == -1) ||
                (m == 4 && q[0] == -1) || (m == 5 && q[0] == -1) || (m == 6 && q[0] == -1) ||
                (m == 7 && q[5] == -1) || (m == 8 && q[1] == -1))
                continue;
            fo_intra4x4_pred(m, q, pr);
            int s = cost4x4(c, pr, blk);
            if (s < min4) {
                c->i4mode[(c->cur << 4) + blk] = m;
                min4 = s;
                if (min4 == 0) break;
            }
        }
    }

    int xP = (c->cur % c->mbw) << 4, yP = (c->cur / c->mbw) << 4;
    uint8_t orig[16][16];
    c->mb_type[c->cur] = 0;
    for (int blk = 0; blk < 16; blk++) {
        set_i4_mode(c, blk);
        int q[14], pr[4][4], d[4][4], r[4][4];
        fo_intra4x4_fetch(c, blk, q);
        fo_intra4x4_pred(c->i4mode[(c->cur << 4) + blk], q, pr);
        int x0 = fo_blk_xy[blk][0], y0 = fo_blk_xy[blk][1];
        for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++) {
                predL[y0 + y][x0 + x] = pr[y][x];
                orig[y0 + y][x0 + x] = c->L[(yP + y0 + y) * c->W + xP + x0 + x];
                d[y][x] = c->L[(yP + y0 + y) * c->W + xP + x0 + x] - pr[y][x];
            }
        fo_forwardResidual(c->QPy, d, r, 0);
        fo_scan(r, c->lv.Lumalevel[blk], 0);
        fo_transformDecoding4x4Luma(c, c->lv.Lumalevel, predL, blk, c->QPy);
    }
    unsigned bits4 = fo_coded_mb_size(c, -1, predL, predCb, predCr);
    if (c->dbg_mbsize) {
        c->dbg_mbsize[c->cur][0] = (int)min;
        c->dbg_mbsize[c->cur][1] = (int)bits4;
    }
    if (bits4 < min) return -1;
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) c->L[(yP + i) * c->W + xP + j] = orig[i][j];
    fo_intra16_fetch(c, p);
    fo_intra16_pred(mode16, p, predL);
    return mode16;
}
