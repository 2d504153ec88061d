/*
 * fo_inter.c -- ORACLE (test infrastructure): rows a15..a18 of SURVEY.md 8a.
 * Motion-vector prediction (F/mode_pred.cpp), motion compensation
 * (F/mocomp.cpp), interpolated reference planes + box features + counting
 * sort (F/moestimation.cpp:74-173) and the P-macroblock decision
 * interEncoding (F/moestimation.cpp:175-585).  Quirks are kept on purpose and
 * marked QUIRK.
 */
#include "fo.h"
#include <limits.h>
#include <stdlib.h>
#include <string.h>

static inline int iabs(int a) { return a < 0 ? -a : a; }
static inline int clip255(int x) { return x < 0 ? 0 : (x > 255 ? 255 : x); }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int median3(int a, int b, int c) { return imax(imin(a, b), imin(c, imax(a, b))); }

/* P_and_SP_macroblock_modes columns 2,5,6 (F/h264_globals.cpp:25-58) */
static int num_mb_part(int t)
{
    if (t == 0 || t == FO_P_SKIP) return 1;
    if (t == 1 || t == 2) return 2;
    if (t == 3 || t == 4) return 4;
    if (t == 5) return 0;
    return 0xff; /* NA: intra 16x16 rows and I_PCM */
}
static int part_w(int t)
{
    if (t == 0 || t == 1 || t == FO_P_SKIP) return 16;
    if (t == 2 || t == 3 || t == 4) return 8;
    if (t == 5) return 0xff;
    if (t >= 6 && t <= 29) return ((t - 6) / 4) % 3; /* column 5 of the I16x16 rows = cbp chroma */
    return 0xff;
}
static int part_h(int t)
{
    if (t == 0 || t == 2 || t == FO_P_SKIP) return 16;
    if (t == 1 || t == 3 || t == 4) return 8;
    if (t == 5) return 0xff;
    if (t >= 6 && t <= 29) return (t >= 18) ? 15 : 0;
    return 0xff;
}
static int num_sub_part(int s) /* P_sub_macroblock_modes[s+1][2] */
{
    static const int n[4] = {1, 2, 2, 4};
    return (s >= 0 && s < 4) ? n[s] : 0xff;
}

/* F/mode_pred.cpp:61-99 */
static void nbr_loc(const fo_ctx *c, int xN, int yN, int *mbN, int *xW, int *yW, int *valid)
{
    int W = c->mbw, cur = c->cur;
    *xW = xN;
    *yW = yN;
    *valid = 0;
    if (*xW > 15 && *yW >= 0) return;
    if (*yW > 15) return;
    *valid = 1;
    *mbN = cur;
    if (*xW >= 0 && *xW < 16 && *yW >= 0) return;
    *mbN = cur - W;
    if (*xW >= 0 && *xW < 16) {
        if (cur < W) *valid = 0;
        *yW += 16;
        return;
    }
    (*mbN)++;
    if (*xW > 15) {
        if (cur < W) *valid = 0;
        *xW -= 16;
        *yW += 16;
        if ((*mbN) % W == 0) *valid = 0;
        return;
    }
    *xW += 16;
    *mbN -= 2;
    if (*yW < 0) {
        if (cur < W) *valid = 0;
        if (cur % W == 0) *valid = 0;
        *yW += 16;
        return;
    }
    if (cur % W == 0) *valid = 0;
    *mbN = cur - 1;
}

/* F/mode_pred.cpp:102-110: only the mbPartIdx result is ever consumed */
static int part_of(int xP, int yP, int t)
{
    if (num_mb_part(t) == 0xff) return 0;
    /* QUIRK: for I_4x4 inside a P slice (t == 5) the reference divides by NA (255) */
    return ((yP / part_h(t)) << 1) + (xP / part_w(t));
}

typedef struct {
    int mb[4], part[4], valid[4];
} nbrs;

/* F/mode_pred.cpp:113-160 (6.4.10.7) */
static void nbr_parts(const fo_ctx *c, int mbPartIdx, int subIdx, nbrs *n)
{
    int t = c->cur_mb_type;
    int pw = part_w(t), ph = part_h(t);
    int x = (mbPartIdx % (16 / pw)) * pw, y = (mbPartIdx / (16 / pw)) * ph;
    int xS = 0, yS = 0, ppw = 16;
    if (t == FO_P_8x8 || t == FO_P_8x8ref0) {
        xS = fo_blk_xy[subIdx][0];
        yS = fo_blk_xy[subIdx][1];
        ppw = 8;
        if (c->sub_mb_type[mbPartIdx] == 3 || c->sub_mb_type[mbPartIdx] == 2) ppw = 4;
    }
    if (t == FO_P_8x16) ppw = 8;
    int xs[4] = {x + xS - 1, x + xS, x + xS + ppw, x + xS - 1};
    int ys[4] = {y + yS, y + yS - 1, y + yS - 1, y + yS - 1};
    for (int k = 0; k < 4; k++) {
        int xW, yW;
        n->mb[k] = 0;
        n->part[k] = 0;
        nbr_loc(c, xs[k], ys[k], &n->mb[k], &xW, &yW, &n->valid[k]);
        if (n->valid[k]) n->part[k] = part_of(xW, yW, c->mb_type[n->mb[k]]);
    }
}

/* F/mode_pred.cpp:49-58 */
static void nbr_mv(const fo_ctx *c, int mbN, int part, int *mx, int *my, int *ref)
{
    int np = num_mb_part(c->mb_type[mbN]);
    if (np == 0xff || np == 0) {
        *mx = 0;
        *my = 0;
        *ref = -1;
        return;
    }
    *mx = c->mvx[mbN][part][0];
    *my = c->mvy[mbN][part][0];
    *ref = c->refidx[mbN];
}

/* shared tail of PredictMV_Luma / PredictMV_LumaSubMB (F/mode_pred.cpp:203-249, :298-333).
 * Returns 1 when one of the single-reference early returns fired, 0 for the median. */
static int median_tail(int cref, int mx[3], int my[3], int ref[3], int *ox, int *oy)
{
    if (mx[0] == FO_MV_NA && mx[1] == FO_MV_NA) {
        mx[0] = 0;
        my[0] = 0;
        ref[0] = cref;
    }
    if (mx[0] == FO_MV_NA && mx[1] != FO_MV_NA) {
        mx[0] = 0;
        my[0] = 0;
        ref[0] = -1;
    }
    if (mx[1] == FO_MV_NA) {
        mx[1] = mx[0];
        my[1] = my[0];
        ref[1] = ref[0];
    }
    if (mx[2] == FO_MV_NA) {
        mx[2] = mx[0];
        my[2] = my[0];
        ref[2] = ref[0];
    }
    if (ref[0] == cref && ref[1] != cref && ref[2] != cref) {
        *ox = mx[0];
        *oy = my[0];
        return 1;
    }
    if (ref[0] != cref && ref[1] == cref && ref[2] != cref) {
        *ox = mx[1];
        *oy = my[1];
        return 1;
    }
    if (ref[0] != cref && ref[1] != cref && ref[2] == cref) {
        *ox = mx[2];
        *oy = my[2];
        return 1;
    }
    *ox = median3(mx[0], mx[1], mx[2]);
    *oy = median3(my[0], my[1], my[2]);
    return 0;
}

static void gather(fo_ctx *c, int mbPartIdx, int subIdx, int mx[3], int my[3], int ref[3])
{
    nbrs n;
    for (int i = 0; i < 3; i++) {
        mx[i] = my[i] = FO_MV_NA;
        ref[i] = -1;
    }
    nbr_parts(c, mbPartIdx, subIdx, &n);
    if (!n.valid[2]) {
        n.valid[2] = n.valid[3];
        n.mb[2] = n.mb[3];
        n.part[2] = n.part[3];
    }
    for (int i = 0; i < 3; i++)
        if (n.valid[i]) nbr_mv(c, n.mb[i], n.part[i], &mx[i], &my[i], &ref[i]);
}

/* F/mode_pred.cpp:163-249 */
static void predict_sub(fo_ctx *c, int mbPartIdx, int subIdx)
{
    int cur = c->cur, cref = c->refidx[cur];
    int mx[3], my[3], ref[3];
    int st = c->sub_mb_type[subIdx]; /* QUIRK: indexed by subMbPartIdx (F/mode_pred.cpp:168) */
    gather(c, mbPartIdx, subIdx, mx, my, ref);
    int k = -1;
    if (st == 1 && subIdx == 0 && mx[1] != FO_MV_NA && cref == ref[1])
        k = 1;
    else if (st == 1 && subIdx == 1 && mx[0] != FO_MV_NA && cref == ref[0])
        k = 0;
    else if (st == 2 && subIdx == 0 && mx[0] != FO_MV_NA && cref == ref[0])
        k = 0;
    else if (st == 2 && subIdx == 1 && mx[2] != FO_MV_NA && cref == ref[2])
        k = 2;
    if (k >= 0) {
        c->mvx[cur][mbPartIdx][subIdx] = mx[k];
        c->mvy[cur][mbPartIdx][subIdx] = my[k];
        return;
    }
    median_tail(cref, mx, my, ref, &c->mvx[cur][mbPartIdx][subIdx], &c->mvy[cur][mbPartIdx][subIdx]);
}

/* F/mode_pred.cpp:252-371 */
static void predict_luma(fo_ctx *c, int mbPartIdx)
{
    int cur = c->cur, cref = c->refidx[cur], t = c->cur_mb_type;
    int mx[3], my[3], ref[3];
    gather(c, mbPartIdx, 0, mx, my, ref);
    int k = -1;
    if (t == FO_P_16x8 && mbPartIdx == 0 && mx[1] != FO_MV_NA && cref == ref[1])
        k = 1;
    else if (t == FO_P_16x8 && mbPartIdx == 1 && mx[0] != FO_MV_NA && cref == ref[0])
        k = 0;
    else if (t == FO_P_8x16 && mbPartIdx == 0 && mx[0] != FO_MV_NA && cref == ref[0])
        k = 0;
    else if (t == FO_P_8x16 && mbPartIdx == 1 && mx[2] != FO_MV_NA && cref == ref[2])
        k = 2;
    if (k >= 0) {
        c->mvx[cur][mbPartIdx][0] = mx[k];
        c->mvy[cur][mbPartIdx][0] = my[k];
        return;
    }
    /* the three single-reference early returns of :328-341 skip the sub-MB part below */
    if (median_tail(cref, mx, my, ref, &c->mvx[cur][mbPartIdx][0], &c->mvy[cur][mbPartIdx][0])) return;
    if (t == FO_P_8x8 || t == FO_P_8x8ref0) {
        int (*vx)[4] = c->mvx[cur], (*vy)[4] = c->mvy[cur];
        int st = c->sub_mb_type[mbPartIdx];
        predict_sub(c, mbPartIdx, 0);
        if (num_sub_part(st) > 1) {
            predict_sub(c, mbPartIdx, 1);
            if (num_sub_part(st) > 2) {
                predict_sub(c, mbPartIdx, 2);
                predict_sub(c, mbPartIdx, 3);
            } else if (st == 2) {
                vx[mbPartIdx][2] = vx[mbPartIdx][0];
                vy[mbPartIdx][2] = vy[mbPartIdx][0];
                vx[mbPartIdx][3] = vx[mbPartIdx][1];
                vy[mbPartIdx][3] = vy[mbPartIdx][1];
            } else {
                vx[mbPartIdx][2] = vx[mbPartIdx][1];
                vy[mbPartIdx][2] = vy[mbPartIdx][1];
                vx[mbPartIdx][3] = vx[mbPartIdx][1];
                vy[mbPartIdx][3] = vy[mbPartIdx][1];
                vx[mbPartIdx][1] = vx[mbPartIdx][0];
                vy[mbPartIdx][1] = vy[mbPartIdx][0];
            }
        } else {
            for (int i = 1; i < 3; i++) {
                vx[mbPartIdx][i] = vx[mbPartIdx][0];
                vy[mbPartIdx][i] = vy[mbPartIdx][0];
            }
        }
    }
}

/* F/mode_pred.cpp:381-425 */
static void predict_mv(fo_ctx *c)
{
    int cur = c->cur, W = c->mbw, t = c->cur_mb_type;
    if (t == FO_P_SKIP) {
        memset(c->mvd, 0, sizeof c->mvd);
        c->refidx[cur] = 0;
        if (cur < W || cur % W == 0) {
            c->mvx[cur][0][0] = 0;
            c->mvy[cur][0][0] = 0;
        } else {
            int up = cur - W, lf = cur - 1;
            int npu = num_mb_part(c->mb_type[up]), npl = num_mb_part(c->mb_type[lf]);
            int zu = ((npu == 0) | (npu == 0xff) | c->refidx[up] | c->mvx[up][2][0] | c->mvy[up][2][0]) == 0;
            int zl = ((npl == 0) | (npl == 0xff) | c->refidx[lf] | c->mvx[lf][1][0] | c->mvy[lf][1][0]) == 0;
            if (zu || zl) {
                c->mvx[cur][0][0] = 0;
                c->mvy[cur][0][0] = 0;
            } else {
                predict_luma(c, 0);
            }
        }
    } else {
        int np = num_mb_part(t);
        predict_luma(c, 0);
        c->refidx[cur] = 0;
        c->mvx[cur][0][0] += c->mvd[0][0][0];
        c->mvy[cur][0][0] += c->mvd[0][0][1];
        if (np > 1) {
            predict_luma(c, 1);
            c->mvx[cur][1][0] += c->mvd[1][0][0];
            c->mvy[cur][1][0] += c->mvd[1][0][1];
            if (np > 2) {
                predict_luma(c, 2);
                c->mvx[cur][2][0] += c->mvd[2][0][0];
                c->mvy[cur][2][0] += c->mvd[2][0][1];
                predict_luma(c, 3);
                c->mvx[cur][3][0] += c->mvd[3][0][0];
                c->mvy[cur][3][0] += c->mvd[3][0][1];
            }
        }
    }
}

/* F/mode_pred.cpp:428-482 */
void fo_DeriveMVs(fo_ctx *c)
{
    int cur = c->cur, t = c->cur_mb_type;
    int (*vx)[4] = c->mvx[cur], (*vy)[4] = c->mvy[cur];
    predict_mv(c);
    int np = num_mb_part(t);
    if (np == 1) {
        for (int i = 1; i < 4; i++) {
            vx[i][0] = vx[0][0];
            vy[i][0] = vy[0][0];
        }
    }
    if (np == 2) {
        if (t == FO_P_16x8) {
            vx[2][0] = vx[1][0];
            vy[2][0] = vy[1][0];
            vx[1][0] = vx[0][0];
            vy[1][0] = vy[0][0];
            vx[3][0] = vx[2][0];
            vy[3][0] = vy[2][0];
        } else {
            vx[2][0] = vx[0][0];
            vy[2][0] = vy[0][0];
            vx[3][0] = vx[1][0];
            vy[3][0] = vy[1][0];
        }
    }
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            vx[i][j] = vx[i][0];
            vy[i][j] = vy[i][0];
        }
}

/* ------------------------------------------------------------------ MC */

static inline int tap6(int E, int F, int G, int H, int I, int J)
{
    return clip255((E - 5 * F + 20 * G + 20 * H - 5 * I + J + 16) >> 5);
}
#define MID(a, b) (((a) + (b) + 1) >> 1)
#define PX(x, y) d[(y) * 9 + (x)]

/* F/mocomp.cpp:50-78.  QUIRK: the centre sample j filters already clipped
 * and rounded intermediates (cc,dd,h,m,ee,ff). */
static int luma_frac(const int *d, int frac)
{
    int b, cc, dd, ee, ff, h, j, m, s;
    if (frac == 0) return PX(0, 0);
    b = tap6(PX(-2, 0), PX(-1, 0), PX(0, 0), PX(1, 0), PX(2, 0), PX(3, 0));
    if (frac == 1) return MID(PX(0, 0), b);
    if (frac == 2) return b;
    if (frac == 3) return MID(b, PX(1, 0));
    h = tap6(PX(0, -2), PX(0, -1), PX(0, 0), PX(0, 1), PX(0, 2), PX(0, 3));
    if (frac == 4) return MID(PX(0, 0), h);
    if (frac == 8) return h;
    if (frac == 12) return MID(h, PX(0, 1));
    if (frac == 5) return MID(b, h);
    m = tap6(PX(1, -2), PX(1, -1), PX(1, 0), PX(1, 1), PX(1, 2), PX(1, 3));
    if (frac == 7) return MID(b, m);
    s = tap6(PX(-2, 1), PX(-1, 1), PX(0, 1), PX(1, 1), PX(2, 1), PX(3, 1));
    if (frac == 13) return MID(h, s);
    if (frac == 15) return MID(s, m);
    cc = tap6(PX(-2, -2), PX(-2, -1), PX(-2, 0), PX(-2, 1), PX(-2, 2), PX(-2, 3));
    dd = tap6(PX(-1, -2), PX(-1, -1), PX(-1, 0), PX(-1, 1), PX(-1, 2), PX(-1, 3));
    ee = tap6(PX(2, -2), PX(2, -1), PX(2, 0), PX(2, 1), PX(2, 2), PX(2, 3));
    ff = tap6(PX(3, -2), PX(3, -1), PX(3, 0), PX(3, 1), PX(3, 2), PX(3, 3));
    j = tap6(cc, dd, h, m, ee, ff);
    if (frac == 10) return j;
    if (frac == 6) return MID(b, j);
    if (frac == 9) return MID(h, j);
    if (frac == 14) return MID(j, s);
    if (frac == 11) return MID(j, m);
    return 128;
}

/* F/mocomp.cpp:152-195 with the fetch of :11-36 */
void fo_mc_sub(fo_ctx *c, int predL[16][16], int predCr[8][8], int predCb[8][8], const uint8_t *rL,
               const uint8_t *rCb, const uint8_t *rCr, int mb, int sub, int part)
{
    int W = c->W, H = c->H, Wc = c->Wc, Hc = c->Hc;
    int org_y = ((sub & 2) << 2) + ((part & 2) << 1);
    int org_x = ((sub & 1) << 3) + ((part & 1) << 2);
    int mvx = c->mvx[mb][sub][part], mvy = c->mvy[mb][sub][part];
    int xAl = ((mb % c->mbw) << 4) + org_x, yAl = ((mb / c->mbw) << 4) + org_y;
    int Lt[9][9], Ct[2][3][3];
    int ox = xAl + (mvx >> 2) - 2, oy = yAl + (mvy >> 2) - 2;
    for (int y = 0; y < 9; y++) {
        int sy = oy + y;
        if (sy < 0) sy = 0;
        if (sy >= H) sy = H - 1;
        for (int x = 0; x < 9; x++) {
            int sx = ox + x;
            if (sx < 0) sx = 0;
            if (sx >= W) sx = W - 1;
            Lt[y][x] = rL[sy * W + sx];
        }
    }
    int cx = xAl / 2 + (mvx >> 3), cy = yAl / 2 + (mvy >> 3);
    for (int y = 0; y < 3; y++) {
        int sy = cy + y;
        if (sy < 0) sy = 0;
        if (sy >= Hc) sy = Hc - 1;
        for (int x = 0; x < 3; x++) {
            int sx = cx + x;
            if (sx < 0) sx = 0;
            if (sx >= Wc) sx = Wc - 1;
            Ct[0][y][x] = rCb[sy * Wc + sx];
            Ct[1][y][x] = rCr[sy * Wc + sx];
        }
    }
    int frac = (mvy & 3) * 4 + (mvx & 3);
    for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++) predL[org_y + y][org_x + x] = luma_frac(&Lt[y + 2][x + 2], frac);
    org_x /= 2;
    org_y /= 2;
    int xl = mvx & 7, yl = mvy & 7;
    for (int y = 0; y < 2; y++)
        for (int x = 0; x < 2; x++) {
            predCb[org_y + y][org_x + x] = ((8 - xl) * (8 - yl) * Ct[0][y][x] + xl * (8 - yl) * Ct[0][y][x + 1] +
                                            (8 - xl) * yl * Ct[0][y + 1][x] + xl * yl * Ct[0][y + 1][x + 1] + 32) >>
                                           6;
            predCr[org_y + y][org_x + x] = ((8 - xl) * (8 - yl) * Ct[1][y][x] + xl * (8 - yl) * Ct[1][y][x + 1] +
                                            (8 - xl) * yl * Ct[1][y + 1][x] + xl * yl * Ct[1][y + 1][x + 1] + 32) >>
                                           6;
        }
}

/* F/mocomp.cpp:200-208 (every RefPicList0 entry is the single dpb) */
void fo_Decode(fo_ctx *c, int predL[16][16], int predCr[8][8], int predCb[8][8])
{
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) fo_mc_sub(c, predL, predCr, predCb, c->dL, c->dC[0], c->dC[1], c->cur, i, j);
}

/* --------------------------------------------------- FillInterpolatedRefFrame */

#define KAR(k, f, y, x) c->kar[k][f][(size_t)(y) * (size_t)(c->W + 8) + (size_t)(x)]

/* F/moestimation.cpp:74-173 */
void fo_fill_interpolated(fo_ctx *c)
{
    int W = c->W, H = c->H;
    if (!c->interp[0]) {
        for (int i = 0; i < 16; i++) {
            c->interp[i] = (uint8_t *)malloc((size_t)W * H);
            for (int k = 0; k < 5; k++) c->kar[k][i] = (int *)malloc(sizeof(int) * (size_t)(W + 8) * (H + 8));
        }
        for (int k = 0; k < 5; k++) {
            c->sorted[k] = (int *)malloc(sizeof(int) * (size_t)W * H);
            c->sorted_tmp[k] = (int *)malloc(sizeof(int) * (size_t)W * H);
        }
    }
    int predL[16][16], predCr[8][8], predCb[8][8];
    for (int frac = 0; frac < 16; frac++) {
        int mvx = frac & 3, mvy = (frac & 12) / 4;
        for (int mb = 0; mb < c->nmb; mb++) {
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 4; j++) {
                    c->mvx[mb][i][j] = mvx; /* QUIRK: clobbers the MV field */
                    c->mvy[mb][i][j] = mvy;
                    fo_mc_sub(c, predL, predCr, predCb, c->dL, c->dC[0], c->dC[1], mb, i, j);
                }
            int x0 = (mb % c->mbw) << 4, y0 = (mb / c->mbw) << 4;
            for (int i = 0; i < 16; i++)
                for (int j = 0; j < 16; j++) c->interp[frac][(y0 + i) * W + x0 + j] = (uint8_t)predL[i][j];
        }
    }
    for (int f = 0; f < 16; f++) {
        const uint8_t *P = c->interp[f];
        for (int tx = W + 7; tx >= 0; tx--)
            for (int ty = H + 7; ty >= 0; ty--) {
                int sx = tx < W ? tx : W - 1, sy = ty < H ? ty : H - 1;
                int v = P[sy * W + sx];
                if (ty < H + 7) v += KAR(0, f, ty + 1, tx);
                if (tx < W + 7) v += KAR(0, f, ty, tx + 1);
                if (ty < H + 7 && tx < W + 7) v -= KAR(0, f, ty + 1, tx + 1);
                KAR(0, f, ty, tx) = v;
            }
        for (int tx = 0; tx < W; tx++)
            for (int ty = 0; ty < H; ty++) {
#define S(y, x) KAR(0, f, y, x)
                KAR(4, f, ty, tx) = S(ty, tx) - S(ty, tx + 2) - S(ty + 8, tx) + S(ty + 8, tx + 2) + S(ty, tx + 4) -
                                    S(ty, tx + 6) - S(ty + 8, tx + 4) + S(ty + 8, tx + 6);
                KAR(3, f, ty, tx) = S(ty, tx) - S(ty + 2, tx) - S(ty, tx + 8) + S(ty + 2, tx + 8) + S(ty + 4, tx) -
                                    S(ty + 6, tx) - S(ty + 4, tx + 8) + S(ty + 6, tx + 8);
                KAR(2, f, ty, tx) = S(ty, tx) - S(ty + 8, tx) - S(ty, tx + 4) + S(ty + 8, tx + 4);
                KAR(1, f, ty, tx) = S(ty, tx) - S(ty + 4, tx) - S(ty, tx + 8) + S(ty + 4, tx + 8);
                KAR(0, f, ty, tx) = S(ty, tx) - (S(ty + 8, tx) + S(ty, tx + 8) - S(ty + 8, tx + 8));
#undef S
            }
    }
    /* counting sort of all positions of plane 0 by their 8x8 sum, column-major arrival */
    int b = 0;
    for (int i = 0; i < 16385; i++) c->koliko[i] = 0;
    for (int tx = 0; tx < W; tx++)
        for (int ty = 0; ty < H; ty++) {
            c->sorted_tmp[0][b] = KAR(0, 0, ty, tx);
            c->sorted_tmp[3][b] = KAR(1, 0, ty, tx);
            c->sorted_tmp[4][b] = KAR(2, 0, ty, tx);
            c->koliko[c->sorted_tmp[0][b]]++;
            c->sorted_tmp[1][b] = ty;
            c->sorted_tmp[2][b++] = tx;
        }
    int b1;
    b = 0;
    for (int i = 1; i < 16384; i++) { /* QUIRK: bucket 0 is left out of the prefix sum */
        b1 = b + c->koliko[i];
        c->koliko[i] = b;
        b = b1;
    }
    b = 0;
    for (int tx = 0; tx < W; tx++)
        for (int ty = 0; ty < H; ty++) {
            b1 = c->koliko[c->sorted_tmp[0][b]];
            c->koliko[c->sorted_tmp[0][b]]++;
            if (b1 < W * H) {
                c->sorted[0][b1] = c->sorted_tmp[0][b];
                c->sorted[1][b1] = c->sorted_tmp[1][b];
                c->sorted[3][b1] = c->sorted_tmp[3][b];
                c->sorted[4][b1] = c->sorted_tmp[4][b];
                c->sorted[2][b1] = c->sorted_tmp[2][b];
            }
            b++;
        }
    for (int i = 16383; i > 0; i--) c->koliko[i] = c->koliko[i - 1];
    c->koliko[0] = 0;
    c->koliko[16384] = W * H; /* the reference reads one past the array here (only for sums >= 16383) */
    c->me_ready = 1;
}

/* ---------------------------------------------------------------- search */

typedef struct {
    int bmins[85], bxs[85], bys[85];
    int suma[5];
} me_list;

/* F/moestimation.cpp:175-195 ("satd" is a plain SAD against the interpolated planes) */
static int sad8x8(fo_ctx *c, int mvx, int mvy, int blk8)
{
    int W = c->W, H = c->H;
    int xP = ((c->cur % c->mbw) << 4) + (blk8 % 2) * 8, yP = ((c->cur / c->mbw) << 4) + (blk8 / 2) * 8;
    int xPi = xP + (mvx >> 2), yPi = yP + (mvy >> 2);
    if (xPi < 0) xPi = 0;
    if (xPi >= W) xPi = W - 1;
    if (yPi < 0) yPi = 0;
    if (yPi >= H) yPi = H - 1;
    const uint8_t *R = c->interp[(mvx & 3) + (mvy & 3) * 4];
    int s = 0;
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) {
            int px = xPi + j, py = yPi + i;
            if (px >= W) px = W - 1;
            if (py >= H) py = H - 1;
            s += iabs(c->L[(yP + i) * W + xP + j] - (int)R[py * W + px]);
        }
    return s;
}

/* F/moestimation.cpp:246-252 */
static int sad_mvs(fo_ctx *c, int mvx, int mvy, int part)
{
    int t = c->cur_mb_type;
    if (t == FO_P_8x8 || t == FO_P_8x8ref0) return sad8x8(c, mvx, mvy, part);
    if (t == FO_P_16x8) return sad8x8(c, mvx, mvy, part * 2) + sad8x8(c, mvx, mvy, part * 2 + 1);
    if (t == FO_P_8x16) return sad8x8(c, mvx, mvy, part) + sad8x8(c, mvx, mvy, part + 2);
    return sad8x8(c, mvx, mvy, 0) + sad8x8(c, mvx, mvy, 1) + sad8x8(c, mvx, mvy, 2) + sad8x8(c, mvx, mvy, 3);
}

/* F/moestimation.cpp:254-296 */
static void mestimation(fo_ctx *c, me_list *l, int sx, int sy, int granica, int stepMV, int stepFrac, int genx,
                        int geny, int px, int py)
{
    const int *s = l->suma;
    for (int tmpx = px - granica; tmpx <= px + granica; tmpx += stepMV)
        for (int tmpy = py - granica; tmpy <= py + granica; tmpy += stepMV)
            for (int frac = 0; frac < 16; frac += stepFrac) {
                int refx = sx + tmpx, refy = sy + tmpy;
                if (!(refy >= 0 && refy < c->H && refx >= 0 && refx < c->W)) continue;
                int k0 = KAR(0, frac, refy, refx), k1 = KAR(1, frac, refy, refx), k2 = KAR(2, frac, refy, refx),
                    k3 = KAR(3, frac, refy, refx), k4 = KAR(4, frac, refy, refx);
                int d = (iabs(tmpx - genx) + iabs(tmpy - geny) + 4) *
                        (iabs(s[0] - k0) + iabs(s[1] - k1) + iabs(s[0] - s[1] - k0 + k1) + iabs(s[2] - k2) +
                         iabs(s[0] - s[2] - k0 + k2) + iabs(s[3] - k3) + iabs(s[0] - s[3] - k0 + k3) +
                         iabs(s[4] - k4) + iabs(s[0] - s[4] - k0 + k4));
                if (l->bmins[64] < d) continue;
                l->bmins[64] = d;
                l->bxs[64] = (tmpx * 4) | (frac & 3);
                l->bys[64] = (tmpy * 4) | ((frac >> 2) & 3);
                for (int j = 64; j > 0; j--) {
                    if (l->bmins[j] < l->bmins[j - 1]) {
                        int t1 = l->bmins[j];
                        l->bmins[j] = l->bmins[j - 1];
                        l->bmins[j - 1] = t1;
                        t1 = l->bxs[j];
                        l->bxs[j] = l->bxs[j - 1];
                        l->bxs[j - 1] = t1;
                        t1 = l->bys[j];
                        l->bys[j] = l->bys[j - 1];
                        l->bys[j - 1] = t1;
                    } else
                        break;
                }
            }
}

static void eval_list(fo_ctx *c, me_list *l, int n, int need_bmin, int part, int mvpx, int mvpy, int *bmin, int *bx,
                      int *by)
{
    for (int j = 0; j <= n; j++) {
        if (need_bmin && !(l->bmins[j] < 100000000)) continue;
        if (!(l->bxs[j] < 100000000 && l->bys[j] < 100000000)) continue;
        l->bmins[j] = sad_mvs(c, l->bxs[j], l->bys[j], part);
        int cost = l->bmins[j] + iabs(l->bxs[j] - mvpx) + iabs(l->bys[j] - mvpy);
        if (cost < *bmin) {
            *bmin = cost;
            *bx = l->bxs[j];
            *by = l->bys[j];
        }
    }
}

/* the exhaustive search of F/moestimation.cpp:298-390; its result is
 * overwritten by the feature search that always follows (:394-397), but its
 * side effects on frame/mv state and counters are kept. */
static int basic_inter(fo_ctx *c, int predL[16][16], int predCr[8][8], int predCb[8][8]);

/* F/moestimation.cpp:392-585 */
void fo_interEncoding(fo_ctx *c, int predL[16][16], int predCr[8][8], int predCb[8][8])
{
    int W = c->W, cur = c->cur;
    if (c->basic) basic_inter(c, predL, predCr, predCb);
    int xp = (cur % c->mbw) << 4, yp = (cur / c->mbw) << 4;
    int mvx[4], mvy[4];
    memset(c->mvd, 0, sizeof c->mvd);
    c->cur_mb_type = FO_P_SKIP;
    c->mb_type[cur] = FO_P_SKIP;
    fo_DeriveMVs(c);
    fo_Decode(c, predL, predCr, predCb);
    if (c->maxdiff_set == -1) {
        int bla = 0;
        for (int tx = 0; tx < 16; tx++)
            for (int ty = 0; ty < 16; ty++) bla += c->L[(ty + yp) * W + tx + xp];
        int m = bla / 256;
        bla = 0;
        for (int tx = 0; tx < 16; tx++)
            for (int ty = 0; ty < 16; ty++) bla += iabs((int)c->L[(ty + yp) * W + tx + xp] - m);
        c->MAXDIFF = bla / 256;
        if (c->MAXDIFF < 3) c->MAXDIFF = 3;
    } else {
        c->MAXDIFF = c->maxdiff_set;
    }
    int exact = 0;
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) exact += iabs(c->L[(yp + i) * W + xp + j] - predL[i][j]) <= c->MAXDIFF;
    if (exact == 256) {
        c->type_count[0]++;
        for (int i = 0; i < 16; i++)
            for (int j = 0; j < 16; j++) c->L[(yp + i) * W + xp + j] = (uint8_t)predL[i][j];
        return;
    }
    c->cur_mb_type = FO_P_8x8ref0;
    c->mb_type[cur] = FO_P_8x8ref0;
    memset(c->mvd, 0, sizeof c->mvd);
    me_list l;
    for (int i = 0; i < 4; i++) {
        c->mvd[i][0][0] = c->mvd[i][0][1] = 0;
        fo_DeriveMVs(c);
        int mvpx = c->mvx[cur][i][0], mvpy = c->mvy[cur][i][0];
        int genx = mvpx >> 2, geny = mvpy >> 2;
        int relx = (i % 2) * 8, rely = (i / 2) * 8;
        for (int te = 0; te < 5; te++) l.suma[te] = 0;
        for (int tx = 0; tx < 8; tx++)
            for (int ty = 0; ty < 8; ty++) {
                int v = c->L[(ty + rely + yp) * W + tx + relx + xp];
                l.suma[0] += v;
                l.suma[1] += (ty > 3) ? 0 : v;
                l.suma[2] += (tx > 3) ? 0 : v;
                l.suma[3] += ((ty % 4) > 1) ? 0 : v;
                l.suma[4] += ((tx % 4) > 1) ? 0 : v;
            }
        int bx = 0, by = 0, bmin;
        for (int j = 0; j < 85; j++) {
            l.bmins[j] = 1000000000;
            l.bxs[j] = l.bys[j] = 100000000;
        }
        mestimation(c, &l, relx + xp, rely + yp, c->window / 16, 1, 1, genx, geny, genx, geny);
        bmin = 2000000000;
        eval_list(c, &l, 16, 0, i, mvpx, mvpy, &bmin, &bx, &by);
        if (!c->basic) {
            int tren = 0;
            for (int j = 0; j < 85; j++) l.bmins[j] = 1000000000;
            for (int j = 0; j <= 180; j++) {
                for (int side = 0; side < 2; side++) {
                    int a = side == 0 ? l.suma[0] - j : l.suma[0] + j; /* QUIRK: j == 0 visits the bucket twice */
                    if (a >= 0 && a < 16384) {
                        for (int k = c->koliko[a]; k < c->koliko[a + 1]; k++)
                            if (iabs(c->sorted[2][k] - relx - xp) + iabs(c->sorted[1][k] - rely - yp) < 280 &&
                                iabs(c->sorted[3][k] - l.suma[1]) < 100 && iabs(c->sorted[4][k] - l.suma[2]) < 100) {
                                tren++;
                                mestimation(c, &l, relx + xp, rely + yp, 0, 1, 16, genx, geny,
                                            c->sorted[2][k] - relx - xp, c->sorted[1][k] - rely - yp);
                            }
                    }
                }
                if (tren > 128) break;
            }
            eval_list(c, &l, 32, 1, i, mvpx, mvpy, &bmin, &bx, &by);
            for (int j = 0; j < 85; j++) l.bmins[j] = 1000000000;
            mestimation(c, &l, relx + xp, rely + yp, c->window / 2, 1, 16, 0, 0, 0, 0);
            mestimation(c, &l, relx + xp, rely + yp, c->window / 16, 1, 1, 0, 0, 0, 0);
            eval_list(c, &l, 32, 1, i, mvpx, mvpy, &bmin, &bx, &by);
        }
        mvx[i] = bx;
        mvy[i] = by;
        bx -= c->mvx[cur][i][0];
        by -= c->mvy[cur][i][0];
        c->mvd[i][0][0] = bx;
        c->mvd[i][0][1] = by;
    }
    c->type_count[4]++;
    if (mvx[0] == mvx[1] && mvx[0] == mvx[2] && mvx[0] == mvx[3] && mvy[0] == mvy[1] && mvy[0] == mvy[2] &&
        mvy[0] == mvy[3]) {
        c->cur_mb_type = c->mb_type[cur] = FO_P_L0_16x16;
        c->type_count[1]++;
        c->type_count[4]--;
    } else if (mvx[0] == mvx[1] && mvx[2] == mvx[3] && mvy[0] == mvy[1] && mvy[2] == mvy[3]) {
        c->cur_mb_type = c->mb_type[cur] = FO_P_16x8;
        mvx[1] = mvx[2];
        mvy[1] = mvy[2];
        c->type_count[2]++;
        c->type_count[4]--;
    } else if (mvx[0] == mvx[2] && mvx[1] == mvx[3] && mvy[0] == mvy[2] && mvy[1] == mvy[3]) {
        c->cur_mb_type = c->mb_type[cur] = FO_P_8x16;
        c->type_count[3]++;
        c->type_count[4]--;
    }
    int t = c->cur_mb_type;
    int np = (t == FO_P_L0_16x16) ? 1 : ((t == FO_P_16x8 || t == FO_P_8x16) ? 2 : 4);
    for (int i = 0; i < np; i++) {
        c->mvd[i][0][0] = c->mvd[i][0][1] = 0;
        fo_DeriveMVs(c);
        if (i == 1 && t == FO_P_16x8) {
            c->mvd[i][0][0] = mvx[i] - c->mvx[cur][2][0];
            c->mvd[i][0][1] = mvy[i] - c->mvy[cur][2][0];
        } else {
            c->mvd[i][0][0] = mvx[i] - c->mvx[cur][i][0];
            c->mvd[i][0][1] = mvy[i] - c->mvy[cur][i][0];
        }
    }
    fo_DeriveMVs(c);
    fo_Decode(c, predL, predCr, predCb);
    /* snap source samples to the prediction (F/moestimation.cpp:570-584) */
    for (int ty = 0; ty < 16; ty++)
        for (int tx = 0; tx < 16; tx++)
            if (iabs(c->L[(yp + ty) * W + xp + tx] - predL[ty][tx]) < c->MAXDIFF)
                c->L[(yp + ty) * W + xp + tx] = (uint8_t)predL[ty][tx];
    for (int ty = 0; ty < 8; ty++)
        for (int tx = 0; tx < 8; tx++) {
            uint8_t *pb = &c->C[0][(yp / 2 + ty) * c->Wc + xp / 2 + tx];
            uint8_t *pr = &c->C[1][(yp / 2 + ty) * c->Wc + xp / 2 + tx];
            if (iabs(*pb - predCb[ty][tx]) <= c->MAXDIFF) *pb = (uint8_t)predCb[ty][tx];
            if (iabs(*pr - predCr[ty][tx]) <= c->MAXDIFF) *pr = (uint8_t)predCr[ty][tx];
        }
}

/* F/moestimation.cpp:298-390 */
static int basic_inter(fo_ctx *c, int predL[16][16], int predCr[8][8], int predCb[8][8])
{
    int W = c->W, cur = c->cur;
    int xp = (cur % c->mbw) << 4, yp = (cur / c->mbw) << 4;
    int cx[4], cy[4];
    memset(c->mvd, 0, sizeof c->mvd);
    c->cur_mb_type = FO_P_SKIP;
    c->mb_type[cur] = FO_P_SKIP;
    fo_DeriveMVs(c);
    fo_Decode(c, predL, predCr, predCb);
    if (c->maxdiff_set == -1) {
        int bla = 0;
        for (int tx = 0; tx < 16; tx++)
            for (int ty = 0; ty < 16; ty++) bla += c->L[(ty + yp) * W + tx + xp];
        int m = bla / 256;
        bla = 0;
        for (int tx = 0; tx < 16; tx++)
            for (int ty = 0; ty < 16; ty++) bla += iabs((int)c->L[(ty + yp) * W + tx + xp] - m);
        c->MAXDIFF = bla / 256;
        if (c->MAXDIFF < 3) c->MAXDIFF = 3;
    } else {
        c->MAXDIFF = c->maxdiff_set;
    }
    int exact = 0;
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) exact += iabs(c->L[(yp + i) * W + xp + j] - predL[i][j]) <= c->MAXDIFF;
    if (exact == 256) {
        c->type_count[0]++;
        for (int i = 0; i < 16; i++)
            for (int j = 0; j < 16; j++) c->L[(yp + i) * W + xp + j] = (uint8_t)predL[i][j];
        return 1;
    }
    c->cur_mb_type = FO_P_8x8ref0;
    c->mb_type[cur] = FO_P_8x8ref0;
    memset(c->mvd, 0, sizeof c->mvd);
    for (int i = 0; i < 4; i++) {
        int minBlock = INT_MAX;
        cx[i] = cy[i] = -255;
        for (int tmvx = -c->window / 2; tmvx <= c->window / 2; tmvx++)
            for (int tmvy = -c->window / 2; tmvy <= c->window / 2; tmvy++) {
                c->mvx[cur][i][0] = tmvx;
                c->mvy[cur][i][0] = tmvy;
                fo_Decode(c, predL, predCr, predCb);
                /* QUIRK sadLuma8x8 (:197-212): block i of the source against the TOP-LEFT 8x8 of predL */
                int sad = 0;
                int bxp = xp + (i % 2) * 8, byp = yp + (i / 2) * 8;
                for (int a = 0; a < 8; a++)
                    for (int b = 0; b < 8; b++) sad += iabs(c->L[(byp + a) * W + bxp + b] - predL[a][b]);
                if (sad < minBlock) {
                    minBlock = sad;
                    cx[i] = tmvx;
                    cy[i] = tmvy;
                }
            }
    }
    c->type_count[4]++;
    if (cx[0] == cx[1] && cx[0] == cx[2] && cx[0] == cx[3] && cy[0] == cy[1] && cy[0] == cy[2] && cy[0] == cy[3]) {
        c->cur_mb_type = c->mb_type[cur] = FO_P_L0_16x16;
        c->type_count[1]++;
        c->type_count[4]--;
    } else if (cx[0] == cx[1] && cx[2] == cx[3] && cy[0] == cy[1] && cy[2] == cy[3]) {
        c->cur_mb_type = c->mb_type[cur] = FO_P_16x8;
        c->type_count[2]++;
        c->type_count[4]--;
    } else if (cx[0] == cx[2] && cx[1] == cx[3] && cy[0] == cy[2] && cy[1] == cy[3]) {
        c->cur_mb_type = c->mb_type[cur] = FO_P_8x16;
        c->type_count[3]++;
        c->type_count[4]--;
    }
    int t = c->cur_mb_type;
    int np = (t == FO_P_L0_16x16) ? 1 : ((t == FO_P_16x8 || t == FO_P_8x16) ? 2 : 4);
    for (int i = 0; i < np; i++) c->mvd[i][0][0] = c->mvd[i][0][1] = 0;
    fo_DeriveMVs(c);
    fo_Decode(c, predL, predCr, predCb);
    return 0;
}
