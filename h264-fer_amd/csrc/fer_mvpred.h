// fer_mvpred.h -- motion-vector prediction (row a18, F/mode_pred.cpp) shared by the encoder's
// resolve kernel and the decoder's parse kernel.
#pragma once
#include "fer_dev.h"

// ------------------------------------------------------------------ MV prediction (a18)
// Every macroblock of a P picture keeps one vector per 8x8 quadrant in d.mv (for 16x16, 16x8,
// 8x16 and P_Skip the quadrants simply repeat the partition vector), so the neighbour partition
// lookup of F/mode_pred.cpp:102-110 reduces to "quadrant that contains the neighbour sample".
struct MvCtx {
    const short *mv;      // stream base [nmb][4][2]
    const int *mb_type;   // stream base, or nullptr when every macroblock is known to be inter (encoder)
    int mbw, cur, type;
};

__device__ __forceinline__ int p_part_w(int t) { return (t == 0 || t == 1 || t == FER_P_SKIP) ? 16 : 8; }
__device__ __forceinline__ int p_part_h(int t) { return (t == 0 || t == 2 || t == FER_P_SKIP) ? 16 : 8; }

// neighbour location + motion vector, F/mode_pred.cpp:49-110 (all MBs of a P picture are inter here)
__device__ void nbr_fetch(const MvCtx &c, int xN, int yN, bool &valid, int &mx, int &my, int &ref)
{
    int W = c.mbw, cur = c.cur;
    int xW = xN, yW = yN, mbN = cur;
    valid = false;
    if (xW > 15 && yW >= 0) return;
    if (yW > 15) return;
    valid = true;
    if (!(xW >= 0 && xW < 16 && yW >= 0)) {
        mbN = cur - W;
        if (xW >= 0 && xW < 16) {
            if (cur < W) valid = false;
            yW += 16;
        } else {
            mbN++;
            if (xW > 15) {
                if (cur < W) valid = false;
                xW -= 16;
                yW += 16;
                if (mbN % W == 0) valid = false;
            } else {
                xW += 16;
                mbN -= 2;
                if (yW < 0) {
                    if (cur < W) valid = false;
                    if (cur % W == 0) valid = false;
                    yW += 16;
                } else {
                    if (cur % W == 0) valid = false;
                    mbN = cur - 1;
                }
            }
        }
    }
    if (!valid) return;
    if (c.mb_type && mbN != cur) {  // get_neighbour_mv (F/mode_pred.cpp:49-58): intra neighbour -> (0,0), refIdx -1
        int t = c.mb_type[mbN];
        if (t >= 5 && t <= 30) {
            mx = 0;
            my = 0;
            ref = -1;
            return;
        }
    }
    int q = ((yW >> 3) << 1) + (xW >> 3);
    mx = c.mv[((size_t)mbN * 4 + q) * 2];
    my = c.mv[((size_t)mbN * 4 + q) * 2 + 1];
    ref = 0;
}

__device__ __forceinline__ int med3(int a, int b, int c) { return max(min(a, b), min(c, max(a, b))); }

// PredictMV_Luma, F/mode_pred.cpp:252-371, for reference index 0 everywhere
__device__ void predict_luma(const MvCtx &c, int part, int &ox, int &oy)
{
    int t = c.type;
    int pw = p_part_w(t), ph = p_part_h(t);
    int x = (part % (16 / pw)) * pw, y = (part / (16 / pw)) * ph;
    int ppw = (t == FER_P_8x8ref0 || t == FER_P_8x16) ? 8 : 16;
    int mx[3], my[3], ref[3];
    bool val[4];
    int dxm = FER_MV_NA, dym = FER_MV_NA, dref = -1;
    for (int i = 0; i < 3; i++) {
        mx[i] = my[i] = FER_MV_NA;
        ref[i] = -1;
    }
    nbr_fetch(c, x - 1, y, val[0], mx[0], my[0], ref[0]);
    nbr_fetch(c, x, y - 1, val[1], mx[1], my[1], ref[1]);
    nbr_fetch(c, x + ppw, y - 1, val[2], mx[2], my[2], ref[2]);
    if (!val[2]) {
        nbr_fetch(c, x - 1, y - 1, val[3], dxm, dym, dref);
        val[2] = val[3];
        mx[2] = dxm;
        my[2] = dym;
        ref[2] = dref;
    }
    if (t == FER_P_16x8 && part == 0 && mx[1] != FER_MV_NA && ref[1] == 0) {
        ox = mx[1];
        oy = my[1];
        return;
    }
    if (t == FER_P_16x8 && part == 1 && mx[0] != FER_MV_NA && ref[0] == 0) {
        ox = mx[0];
        oy = my[0];
        return;
    }
    if (t == FER_P_8x16 && part == 0 && mx[0] != FER_MV_NA && ref[0] == 0) {
        ox = mx[0];
        oy = my[0];
        return;
    }
    if (t == FER_P_8x16 && part == 1 && mx[2] != FER_MV_NA && ref[2] == 0) {
        ox = mx[2];
        oy = my[2];
        return;
    }
    if (mx[0] == FER_MV_NA && mx[1] == FER_MV_NA) {
        mx[0] = 0;
        my[0] = 0;
        ref[0] = 0;
    }
    if (mx[0] == FER_MV_NA && mx[1] != FER_MV_NA) {
        mx[0] = 0;
        my[0] = 0;
        ref[0] = -1;
    }
    if (mx[1] == FER_MV_NA) {
        mx[1] = mx[0];
        my[1] = my[0];
        ref[1] = ref[0];
    }
    if (mx[2] == FER_MV_NA) {
        mx[2] = mx[0];
        my[2] = my[0];
        ref[2] = ref[0];
    }
    if (ref[0] == 0 && ref[1] != 0 && ref[2] != 0) {
        ox = mx[0];
        oy = my[0];
        return;
    }
    if (ref[0] != 0 && ref[1] == 0 && ref[2] != 0) {
        ox = mx[1];
        oy = my[1];
        return;
    }
    if (ref[0] != 0 && ref[1] != 0 && ref[2] == 0) {
        ox = mx[2];
        oy = my[2];
        return;
    }
    ox = med3(mx[0], mx[1], mx[2]);
    oy = med3(my[0], my[1], my[2]);
}

