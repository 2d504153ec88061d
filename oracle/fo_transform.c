/*
 * fo_transform.c -- ORACLE (test infrastructure): rows a1..a9 of SURVEY.md 8a.
 * Forward 4x4 core, quantisers, Hadamards, dequantisers, inverse transforms,
 * zig-zag scans, picture construction and the per-MB driver
 * quantizationTransform.  Reference: F/quantizationTransform.cpp,
 * F/scaleTransform.cpp, F/inttransform.cpp.
 */
#include "fo.h"
#include <string.h>

static inline int clip255(int x) { return x < 0 ? 0 : (x > 255 ? 255 : x); }
static inline int clip3(int lo, int hi, int x) { return x < lo ? lo : (x > hi ? hi : x); }

/* F/quantizationTransform.cpp:41-100.  The shifts-and-adds of the reference
 * are the constants 256, 416 (=256+128+32) and 208 (=128+64+16). */
void fo_forwardTransform4x4(const int r[4][4], int d[4][4])
{
    int h[4][4], f[4][4];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) h[i][j] = (r[i][j] == 0) ? 0 : (r[i][j] * 64 - 32);
    for (int j = 0; j < 4; j++) {
        int a = h[0][j], b = h[1][j], c = h[2][j], e = h[3][j];
        f[0][j] = (256 * a + 256 * b + 256 * c + 256 * e + 512) >> 10;
        f[1][j] = (416 * a + 208 * b - 208 * c - 416 * e + 512) >> 10;
        f[2][j] = (256 * a - 256 * b - 256 * c + 256 * e + 512) >> 10;
        f[3][j] = (208 * a - 416 * b + 416 * c - 208 * e + 512) >> 10;
    }
    for (int i = 0; i < 4; i++) {
        int a = f[i][0], b = f[i][1], c = f[i][2], e = f[i][3];
        d[i][0] = (256 * a + 256 * b + 256 * c + 256 * e + 512) >> 10;
        d[i][1] = (416 * a + 208 * b - 208 * c - 416 * e + 512) >> 10;
        d[i][2] = (256 * a - 256 * b - 256 * c + 256 * e + 512) >> 10;
        d[i][3] = (208 * a - 416 * b + 416 * c - 208 * e + 512) >> 10;
    }
}

/* F/quantizationTransform.cpp:183-223 (the `Intra` argument is unused there) */
void fo_quantResidual(const int d[4][4], int c[4][4], int qP, int keepDC)
{
    int q6 = qP / 6, m = qP % 6;
    if (qP < 24) {
        int qbits = 4 - q6, adjust = 1 << (3 - q6);
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
                int t = ((d[i][j] * (1 << qbits)) - adjust) * fo_level_quantize(m, i, j);
                c[i][j] = (t + 16384) >> 15;
            }
    } else {
        int qbits = q6 - 4;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
                int t = (d[i][j] >> qbits) * fo_level_quantize(m, i, j);
                c[i][j] = (t + 16384) >> 15;
            }
    }
    if (keepDC) c[0][0] = d[0][0];
}

void fo_forwardResidual(int qP, const int in[4][4], int out[4][4], int keepDC)
{
    int d[4][4];
    fo_forwardTransform4x4(in, d);
    fo_quantResidual(d, out, qP, keepDC);
}

/* F/quantizationTransform.cpp:105-152 + :227-260 */
void fo_forwardDCLumaIntra(int qP, const int f[4][4], int c[4][4])
{
    int g[4][4], e[4][4], d[4][4], t[4][4];
    for (int j = 0; j < 4; j++) {
        g[0][j] = f[0][j] + f[3][j];
        g[1][j] = f[1][j] + f[2][j];
        g[2][j] = f[1][j] - f[2][j];
        g[3][j] = f[0][j] - f[3][j];
    }
    for (int j = 0; j < 4; j++) {
        e[0][j] = g[0][j] + g[1][j];
        e[1][j] = g[3][j] + g[2][j];
        e[2][j] = g[0][j] - g[1][j];
        e[3][j] = g[3][j] - g[2][j];
    }
    for (int i = 0; i < 4; i++) {
        d[i][0] = e[i][0] + e[i][3];
        d[i][1] = e[i][1] + e[i][2];
        d[i][2] = e[i][1] - e[i][2];
        d[i][3] = e[i][0] - e[i][3];
    }
    for (int i = 0; i < 4; i++) {
        t[i][0] = (d[i][0] + d[i][1] + 8) >> 4;
        t[i][1] = (d[i][3] + d[i][2] + 8) >> 4;
        t[i][2] = (d[i][0] - d[i][1] + 8) >> 4;
        t[i][3] = (d[i][3] - d[i][2] + 8) >> 4;
    }
    int q6 = qP / 6, ql = fo_level_quantize(qP % 6, 0, 0);
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            int v;
            if (qP >= 36)
                v = (t[i][j] >> (q6 - 6)) * ql;
            else
                v = ((t[i][j] * (1 << (6 - q6))) - (1 << (5 - q6))) * ql;
            c[i][j] = (v + 16384) >> 15;
        }
}

/* F/quantizationTransform.cpp:157-178 + :264-282 */
void fo_forwardDCChroma(int qP, const int f[2][2], int c[2][2])
{
    int d00 = f[0][0] + f[0][1], d01 = f[0][0] - f[0][1];
    int d10 = f[1][0] + f[1][1], d11 = f[1][0] - f[1][1];
    int t[2][2];
    t[0][0] = (d00 + d10 + 2) >> 2;
    t[0][1] = (d01 + d11 + 2) >> 2;
    t[1][0] = (d00 - d10 + 2) >> 2;
    t[1][1] = (d01 - d11 + 2) >> 2;
    int q6 = qP / 6, ql = fo_level_quantize(qP % 6, 0, 0);
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++) {
            int v = ((t[i][j] * 32) >> q6) * ql;
            c[i][j] = (v + 16384) >> 15;
        }
}

/* F/scaleTransform.cpp:308-340 + :101-150 */
void fo_inverseResidual(int qP, const int c[4][4], int r[4][4], int keepDC)
{
    int d[4][4], e[4][4], f[4][4], g[4][4], h[4][4];
    int q6 = qP / 6, m = qP % 6;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            if (qP >= 24)
                d[i][j] = (c[i][j] * fo_level_scale(m, i, j)) * (1 << (q6 - 4));
            else
                d[i][j] = (c[i][j] * fo_level_scale(m, i, j) + (1 << (3 - q6))) >> (4 - q6);
        }
    if (keepDC) d[0][0] = c[0][0];
    for (int i = 0; i < 4; i++) {
        e[i][0] = d[i][0] + d[i][2];
        e[i][1] = d[i][0] - d[i][2];
        e[i][2] = (d[i][1] >> 1) - d[i][3];
        e[i][3] = d[i][1] + (d[i][3] >> 1);
    }
    for (int i = 0; i < 4; i++) {
        f[i][0] = e[i][0] + e[i][3];
        f[i][1] = e[i][1] + e[i][2];
        f[i][2] = e[i][1] - e[i][2];
        f[i][3] = e[i][0] - e[i][3];
    }
    for (int j = 0; j < 4; j++) {
        g[0][j] = f[0][j] + f[2][j];
        g[1][j] = f[0][j] - f[2][j];
        g[2][j] = (f[1][j] >> 1) - f[3][j];
        g[3][j] = f[1][j] + (f[3][j] >> 1);
    }
    for (int j = 0; j < 4; j++) {
        h[0][j] = g[0][j] + g[3][j];
        h[1][j] = g[1][j] + g[2][j];
        h[2][j] = g[1][j] - g[2][j];
        h[3][j] = g[0][j] - g[3][j];
    }
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) r[i][j] = (h[i][j] + 32) >> 6;
}

/* F/scaleTransform.cpp:154-189 + :344-376 */
void fo_inverseDCLumaIntra(int qP, const int c[4][4], int dcY[4][4])
{
    int d[4][4], e[4][4], g[4][4], f[4][4];
    for (int i = 0; i < 4; i++) {
        d[i][0] = c[i][0] + c[i][2];
        d[i][1] = c[i][0] - c[i][2];
        d[i][2] = c[i][1] - c[i][3];
        d[i][3] = c[i][1] + c[i][3];
    }
    for (int i = 0; i < 4; i++) {
        e[i][0] = d[i][0] + d[i][3];
        e[i][1] = d[i][1] + d[i][2];
        e[i][2] = d[i][1] - d[i][2];
        e[i][3] = d[i][0] - d[i][3];
    }
    for (int j = 0; j < 4; j++) {
        g[0][j] = e[0][j] + e[2][j];
        g[1][j] = e[0][j] - e[2][j];
        g[2][j] = e[1][j] - e[3][j];
        g[3][j] = e[1][j] + e[3][j];
    }
    for (int j = 0; j < 4; j++) {
        f[0][j] = g[0][j] + g[3][j];
        f[1][j] = g[1][j] + g[2][j];
        f[2][j] = g[1][j] - g[2][j];
        f[3][j] = g[0][j] - g[3][j];
    }
    int q6 = qP / 6, ls = fo_level_scale(qP % 6, 0, 0);
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            if (qP >= 36)
                dcY[i][j] = (f[i][j] * ls) * (1 << (q6 - 6));
            else
                dcY[i][j] = (f[i][j] * ls + (1 << (5 - q6))) >> (6 - q6);
        }
}

/* F/scaleTransform.cpp:247-261 + :408-421 */
void fo_inverseDCChroma(int qP, const int c[2][2], int dcC[2][2])
{
    int d00 = c[0][0] + c[1][0], d01 = c[0][1] + c[1][1];
    int d10 = c[0][0] - c[1][0], d11 = c[0][1] - c[1][1];
    int f[2][2];
    f[0][0] = d00 + d01;
    f[0][1] = d00 - d01;
    f[1][0] = d10 + d11;
    f[1][1] = d10 - d11;
    int q6 = qP / 6, ls = fo_level_scale(qP % 6, 0, 0);
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++) dcC[i][j] = ((f[i][j] * ls) * (1 << q6)) >> 5;
}

/* F/quantizationTransform.cpp:310-339 */
void fo_scan(const int c[4][4], int list[16], int ac)
{
    if (ac) {
        for (int i = 1; i < 16; i++) list[i - 1] = c[fo_zigzag[i][0]][fo_zigzag[i][1]];
    } else {
        for (int i = 0; i < 16; i++) list[i] = c[fo_zigzag[i][0]][fo_zigzag[i][1]];
    }
}

/* F/scaleTransform.cpp:454-462 */
void fo_invscan(const int list[16], int c[4][4])
{
    for (int i = 0; i < 16; i++) c[fo_zigzag[i][0]][fo_zigzag[i][1]] = list[i];
}

static int chroma_qp(const fo_ctx *c, int QPy) { return fo_qpc[clip3(0, 51, QPy + c->chroma_qp_offset)]; }

/* F/inttransform.cpp:133-155 */
void fo_transformDecoding4x4Luma(fo_ctx *c, int level[16][16], int predL[16][16], int blk, int QPy)
{
    int cc[4][4], r[4][4];
    fo_invscan(level[blk], cc);
    fo_inverseResidual(QPy, cc, r, 0);
    int x0 = fo_blk_xy[blk][0], y0 = fo_blk_xy[blk][1];
    int xP = (c->cur % c->mbw) << 4, yP = (c->cur / c->mbw) << 4;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            c->L[(yP + y0 + i) * c->W + xP + x0 + j] = (uint8_t)clip255(predL[y0 + i][x0 + j] + r[i][j]);
}

/* F/inttransform.cpp:157-213 */
void fo_transformDecoding16x16Luma(fo_ctx *c, int dc[16], int ac[16][16], int predL[16][16], int QPy)
{
    int cc[4][4], dcY[4][4], r[4][4], list[16];
    fo_invscan(dc, cc);
    fo_inverseDCLumaIntra(QPy, cc, dcY);
    int xP = (c->cur % c->mbw) << 4, yP = (c->cur / c->mbw) << 4;
    for (int blk = 0; blk < 16; blk++) {
        int x0 = fo_blk_xy[blk][0], y0 = fo_blk_xy[blk][1];
        list[0] = dcY[y0 >> 2][x0 >> 2];
        for (int k = 1; k < 16; k++) list[k] = ac[blk][k - 1];
        fo_invscan(list, cc);
        fo_inverseResidual(QPy, cc, r, 1);
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++)
                c->L[(yP + y0 + i) * c->W + xP + x0 + j] = (uint8_t)clip255(predL[y0 + i][x0 + j] + r[i][j]);
    }
}

/* F/inttransform.cpp:237-320 */
void fo_transformDecodingChroma(fo_ctx *c, int dc[4], int ac[4][16], int predC[8][8], int QPy, int cb)
{
    if (c->cur_mb_type == FO_P_SKIP) {
        for (int i = 0; i < 4; i++) {
            dc[i] = 0;
            for (int j = 0; j < 16; j++) ac[i][j] = 0;
        }
    }
    int c2[2][2], dcC[2][2];
    for (int i = 0; i < 4; i++) c2[i >> 1][i & 1] = dc[i];
    int qP = chroma_qp(c, QPy);
    fo_inverseDCChroma(qP, c2, dcC);
    uint8_t *plane = cb ? c->C[0] : c->C[1];
    int xP = (c->cur % c->mbw) << 3, yP = (c->cur / c->mbw) << 3;
    for (int blk = 0; blk < 4; blk++) {
        int list[16], cc[4][4], r[4][4];
        list[0] = dcC[blk / 2][blk % 2];
        for (int k = 1; k < 16; k++) list[k] = ac[blk][k - 1];
        fo_invscan(list, cc);
        fo_inverseResidual(qP, cc, r, 1);
        int x0 = (blk % 2) * 4, y0 = (blk / 2) * 4;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++)
                plane[(yP + y0 + i) * c->Wc + xP + x0 + j] = (uint8_t)clip255(predC[y0 + i][x0 + j] + r[i][j]);
    }
}

/* F/inttransform.cpp:215-231 */
void fo_transformDecodingPSkip(fo_ctx *c, int predL[16][16], int predCb[8][8], int predCr[8][8], int QPy)
{
    int lvl[16][16];
    memset(lvl, 0, sizeof lvl);
    for (int blk = 0; blk < 16; blk++) fo_transformDecoding4x4Luma(c, lvl, predL, blk, QPy);
    int dc[4] = {0, 0, 0, 0}, ac[4][16];
    memset(ac, 0, sizeof ac);
    fo_transformDecodingChroma(c, dc, ac, predCb, QPy, 1);
    fo_transformDecodingChroma(c, dc, ac, predCr, QPy, 0);
}

/* MbPartPredMode(mb_type,0) classes used on this path (F/h264_globals.h:123):
 * returns 0 = Intra_4x4, 1 = Intra_16x16, 2 = inter (Pred_L0 or NA). */
static int pred_class(const fo_ctx *c, int mb_type)
{
    if (c->slice_type == 0) { /* P slice: P_and_SP_macroblock_modes rows */
        if (mb_type == 5) return 0;
        if (mb_type >= 6 && mb_type <= 29) return 1;
        return 2;
    }
    if (mb_type == 0) return 0;
    if (mb_type >= 1 && mb_type <= 24) return 1;
    return 2; /* I_PCM etc.: NA */
}

/* F/quantizationTransform.cpp:349-486 */
void fo_quantizationTransform(fo_ctx *c, int predL[16][16], int predCb[8][8], int predCr[8][8], int reconstruct)
{
    int diff[4][4], rL[4][4], DCL[4][4], rDCL[4][4];
    int xP = (c->cur % c->mbw) << 4, yP = (c->cur / c->mbw) << 4;
    int pc = pred_class(c, c->cur_mb_type);
    int qP = c->QPy;
    if (pc != 0) {
        for (int blk = 0; blk < 16; blk++) {
            int x0 = fo_blk_xy[blk][0], y0 = fo_blk_xy[blk][1];
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 4; j++)
                    diff[i][j] = c->L[(yP + y0 + i) * c->W + xP + x0 + j] - predL[y0 + i][x0 + j];
            if (pc == 1) {
                fo_forwardResidual(qP, diff, rL, 1);
                DCL[y0 / 4][x0 / 4] = rL[0][0];
                fo_scan(rL, c->lv.AC16[blk], 1);
            } else {
                fo_forwardResidual(qP, diff, rL, 0);
                fo_scan(rL, c->lv.Lumalevel[blk], 0);
                if (reconstruct) fo_transformDecoding4x4Luma(c, c->lv.Lumalevel, predL, blk, c->QPy);
            }
        }
        if (pc == 1) {
            fo_forwardDCLumaIntra(c->QPy, DCL, rDCL);
            fo_scan(rDCL, c->lv.DC16, 0);
            if (reconstruct) fo_transformDecoding16x16Luma(c, c->lv.DC16, c->lv.AC16, predL, c->QPy);
        }
    }
    int xPC = xP / 2, yPC = yP / 2;
    qP = chroma_qp(c, c->QPy);
    int DCb[2][2], DCr[2][2], rDCb[2][2], rDCr[2][2], dCb[4][4], dCr[4][4], rCb[4][4], rCr[4][4];
    for (int blk = 0; blk < 4; blk++) {
        int x0 = (blk % 2) * 4, y0 = (blk / 2) * 4;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
                dCb[i][j] = c->C[0][(yPC + y0 + i) * c->Wc + xPC + x0 + j] - predCb[y0 + i][x0 + j];
                dCr[i][j] = c->C[1][(yPC + y0 + i) * c->Wc + xPC + x0 + j] - predCr[y0 + i][x0 + j];
            }
        fo_forwardResidual(qP, dCb, rCb, 1);
        fo_forwardResidual(qP, dCr, rCr, 1);
        DCb[y0 >> 2][x0 >> 2] = rCb[0][0];
        DCr[y0 >> 2][x0 >> 2] = rCr[0][0];
        fo_scan(rCb, c->lv.CAC[0][blk], 1);
        fo_scan(rCr, c->lv.CAC[1][blk], 1);
    }
    fo_forwardDCChroma(qP, DCb, rDCb);
    fo_forwardDCChroma(qP, DCr, rDCr);
    for (int i = 0; i < 4; i++) {
        c->lv.CDC[0][i] = rDCb[i / 2][i % 2];
        c->lv.CDC[1][i] = rDCr[i / 2][i % 2];
    }
    if (reconstruct) {
        fo_transformDecodingChroma(c, c->lv.CDC[0], c->lv.CAC[0], predCb, c->QPy, 1);
        fo_transformDecodingChroma(c, c->lv.CDC[1], c->lv.CAC[1], predCr, c->QPy, 0);
    }
}

/* F/rbsp_encoding.cpp:21-105 */
void fo_setCodedBlockPattern(fo_ctx *c)
{
    int pc = pred_class(c, c->cur_mb_type);
    int l = 0, ch = 0;
    for (int i8 = 0; i8 < 4; i8++) {
        int nz = 0;
        for (int i4 = 0; i4 < 4 && !nz; i4++) {
            const int *p = (pc == 1) ? c->lv.AC16[(i8 << 2) + i4] : c->lv.Lumalevel[(i8 << 2) + i4];
            int n = (pc == 1) ? 15 : 16;
            for (int i = 0; i < n; i++)
                if (p[i] != 0) {
                    nz = 1;
                    break;
                }
        }
        if (nz) l |= 1 << i8;
    }
    for (int i = 0; i < 4; i++)
        if (c->lv.CDC[0][i] != 0 || c->lv.CDC[1][i] != 0) {
            ch |= 1;
            break;
        }
    for (int i4 = 0; i4 < 4; i4++) {
        int nz = 0;
        for (int i = 0; i < 15; i++)
            if (c->lv.CAC[0][i4][i] != 0 || c->lv.CAC[1][i4][i] != 0) {
                nz = 1;
                break;
            }
        if (nz) {
            ch |= 2;
            break;
        }
    }
    if (pc == 1 && l != 0) l = 15;
    if (ch == 3) ch = 2;
    c->cbpL = l;
    c->cbpC = ch;
    c->cbp_l[c->cur] = l;
    c->cbp_c[c->cur] = ch;
}

int fo_pred_class(const fo_ctx *c, int mb_type) { return pred_class(c, mb_type); }
