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
    bool coh;             // vectors are read while other wavefronts of the same launch publish them: agent-scope loads
};

// one quadrant's vector as a packed (x | y << 16) word; `coh` = coherent across the XCDs' L2s
__device__ __forceinline__ int ld_mv(const short *mv, size_t q, bool coh)
{
    const int *p = (const int *)(mv + q * 2);
    return coh ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
__device__ __forceinline__ void st_mv_coh(short *mv, size_t q, int x, int y)
{
    __hip_atomic_store((int *)(mv + q * 2), (x & 0xffff) | (y << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int p_part_w(int t) { return (t == 0 || t == 1 || t == FER_P_SKIP) ? 16 : 8; }
__device__ __forceinline__ int p_part_h(int t) { return (t == 0 || t == 2 || t == FER_P_SKIP) ? 16 : 8; }

// neighbour location, F/mode_pred.cpp:60-110: macroblock and 8x8 quadrant holding sample (xN, yN)
// relative to the current macroblock; valid = inside the picture and already decoded
__device__ __forceinline__ void nbr_locate(int W, int cur, int xN, int yN, bool &valid, int &mbN, int &q)
{
    int xW = xN, yW = yN;
    mbN = cur;
    q = 0;
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
    q = ((yW >> 3) << 1) + (xW >> 3);
}

// neighbour lookup of partition `part` of macroblock (mbx, mby): like nbr_locate, without divisions
__device__ __forceinline__ void nbr_locate_xy(int mbw, int mbx, int mby, int xN, int yN, bool &valid, int &mbN, int &q)
{
    const int cur = mby * mbw + mbx;
    int xW = xN, yW = yN;
    mbN = cur;
    q = 0;
    valid = false;
    if (xW > 15 && yW >= 0) return;
    if (yW > 15) return;
    valid = true;
    if (!(xW >= 0 && xW < 16 && yW >= 0)) {
        if (xW >= 0 && xW < 16) {  // above
            mbN = cur - mbw;
            valid = mby > 0;
            yW += 16;
        } else if (xW > 15) {      // above right
            mbN = cur - mbw + 1;
            valid = mby > 0 && mbx + 1 < mbw;
            xW -= 16;
            yW += 16;
        } else if (yW < 0) {       // above left
            mbN = cur - mbw - 1;
            valid = mby > 0 && mbx > 0;
            xW += 16;
            yW += 16;
        } else {                   // left
            mbN = cur - 1;
            valid = mbx > 0;
            xW += 16;
        }
    }
    q = ((yW >> 3) << 1) + (xW >> 3);
}

// neighbour motion vector (all MBs of a P picture are inter in the encoder)
__device__ void nbr_fetch(const MvCtx &c, int xN, int yN, bool &valid, int &mx, int &my, int &ref)
{
    int mbN, q;
    nbr_locate(c.mbw, c.cur, xN, yN, valid, mbN, q);
    if (!valid) return;
    if (c.mb_type && mbN != c.cur) {  // get_neighbour_mv (F/mode_pred.cpp:49-58): intra neighbour -> (0,0), refIdx -1
        int t = c.mb_type[mbN];
        if (t >= 5 && t <= 30) {
            mx = 0;
            my = 0;
            ref = -1;
            return;
        }
    }
    int w = ld_mv(c.mv, (size_t)mbN * 4 + q, c.coh);
    mx = (int)(short)(w & 0xffff);
    my = w >> 16;
    ref = 0;
}

__device__ __forceinline__ int med3(int a, int b, int c) { return max(min(a, b), min(c, max(a, b))); }

// median rule of PredictMV_Luma (F/mode_pred.cpp:322-371) once A, B, C (C already replaced by D when
// unavailable) are known: mx == FER_MV_NA marks an unavailable neighbour, ref is 0 or -1
// Returns true when one of the three single-reference rules fired (the reference returns from PredictMV_Luma there,
// before its sub-macroblock step, F/mode_pred.cpp:328-341).
__device__ __forceinline__ bool predict_core(int mx[3], int my[3], int ref[3], int &ox, int &oy)
{
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
        return true;
    }
    if (ref[0] != 0 && ref[1] == 0 && ref[2] != 0) {
        ox = mx[1];
        oy = my[1];
        return true;
    }
    if (ref[0] != 0 && ref[1] != 0 && ref[2] == 0) {
        ox = mx[2];
        oy = my[2];
        return true;
    }
    ox = med3(mx[0], mx[1], mx[2]);
    oy = med3(my[0], my[1], my[2]);
    return false;
}

// neighbours A, B, C (D in place of an unavailable C) of the block whose top-left sample is (x, y) and whose
// C neighbour sits at x + ppw (F/mode_pred.cpp:113-160, :258-296)
__device__ __forceinline__ void gather_abc(const MvCtx &c, int x, int y, int ppw, int mx[3], int my[3], int ref[3])
{
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
        mx[2] = dxm;
        my[2] = dym;
        ref[2] = dref;
    }
}

// PredictMV_Luma, F/mode_pred.cpp:252-371, for reference index 0 everywhere
__device__ void predict_luma(const MvCtx &c, int part, int &ox, int &oy)
{
    int t = c.type;
    int pw = p_part_w(t), ph = p_part_h(t);
    int x = (part % (16 / pw)) * pw, y = (part / (16 / pw)) * ph;
    int ppw = (t == FER_P_8x8 || t == FER_P_8x8ref0 || t == FER_P_8x16) ? 8 : 16;
    int mx[3], my[3], ref[3];
    gather_abc(c, x, y, ppw, mx, my, ref);
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
    predict_core(mx, my, ref, ox, oy);
}

// PredictMV_Luma for a quadrant of P_8x8 / P_8x8ref0 whose sub-macroblock types are sub[0..3], as far as the
// reference's DeriveMVs keeps it (F/mode_pred.cpp:163-249, :343-369, :476-481).  Every sub-partition vector ends up
// overwritten by the quadrant's FIRST one, so only that one is derived:
//   * the C neighbour sits 4 samples to the right for the 4-wide sub types (2, 3), 8 otherwise;
//   * when one of the single-reference rules fires the reference has already returned;
//   * otherwise PredictMV_LumaSubMB(part, 0) runs on the same neighbours with the directional rules keyed on
//     sub_mb_type[subMbPartIdx] = sub[0] (quirk: not sub[part]): 8x4 takes B, 4x8 takes A, else the median again.
__device__ void predict_luma_quadrant(const MvCtx &c, int part, const int sub[4], int &ox, int &oy)
{
    const int x = (part & 1) * 8, y = (part >> 1) * 8;
    const int ppw = (sub[part] == 2 || sub[part] == 3) ? 4 : 8;
    int mx[3], my[3], ref[3];
    gather_abc(c, x, y, ppw, mx, my, ref);
    const bool a_ok = mx[0] != FER_MV_NA && ref[0] == 0, b_ok = mx[1] != FER_MV_NA && ref[1] == 0;
    const int ax = mx[0], ay = my[0], bx = mx[1], by = my[1];
    if (predict_core(mx, my, ref, ox, oy)) return;
    if (sub[0] == 1 && b_ok) {
        ox = bx;
        oy = by;
    } else if (sub[0] == 2 && a_ok) {
        ox = ax;
        oy = ay;
    }
}

// PredictMV_Luma (F/mode_pred.cpp:252-371) of partition `part` of an inter macroblock of type `type` in the encoder
// (every macroblock of the picture is inter, reference index 0), with the neighbours' vectors already in registers:
// lane j < 20 of `tbl` holds the packed vector of quadrant j & 3 of macroblock {cur, left, above, above right, above
// left}[j >> 2].  All lanes must call this together (ds_bpermute); partition and type may differ per lane.
__device__ __forceinline__ void predict_luma_tbl(int tbl, int mbw, int mbx, int mby, int type, int part, int &ox, int &oy)
{
    const int cur = mby * mbw + mbx;
    const int pw = p_part_w(type), ph = p_part_h(type);
    const int x = pw == 16 ? 0 : (part & 1) * 8, y = ph == 16 ? 0 : (pw == 16 ? part : (part >> 1)) * 8;
    const int ppw = (type == FER_P_8x8 || type == FER_P_8x8ref0 || type == FER_P_8x16) ? 8 : 16;
    auto fetch = [&](int xN, int yN, int &mx, int &my, int &ref) {
        bool valid;
        int mbN, q;
        nbr_locate_xy(mbw, mbx, mby, xN, yN, valid, mbN, q);
        const int which = mbN == cur ? 0 : (mbN == cur - 1 ? 1 : (mbN == cur - mbw ? 2 : (mbN == cur - mbw + 1 ? 3 : 4)));
        const int w = __shfl(tbl, which * 4 + q);
        mx = valid ? (int)(short)(w & 0xffff) : FER_MV_NA;
        my = valid ? (w >> 16) : FER_MV_NA;
        ref = valid ? 0 : -1;
    };
    int mx[3], my[3], ref[3], dx, dy, dr;
    fetch(x - 1, y, mx[0], my[0], ref[0]);
    fetch(x, y - 1, mx[1], my[1], ref[1]);
    fetch(x + ppw, y - 1, mx[2], my[2], ref[2]);
    fetch(x - 1, y - 1, dx, dy, dr);
    if (ref[2] != 0) {  // C unavailable: D takes its place
        mx[2] = dx;
        my[2] = dy;
        ref[2] = dr;
    }
    const bool useB = type == FER_P_16x8 && part == 0 && ref[1] == 0;
    const bool useA = ((type == FER_P_16x8 && part == 1) || (type == FER_P_8x16 && part == 0)) && ref[0] == 0;
    const bool useC = type == FER_P_8x16 && part == 1 && ref[2] == 0;
    const int ax = mx[0], ay = my[0], bx = mx[1], by = my[1], cx = mx[2], cy = my[2];
    predict_core(mx, my, ref, ox, oy);
    if (useB) {
        ox = bx;
        oy = by;
    } else if (useA) {
        ox = ax;
        oy = ay;
    } else if (useC) {
        ox = cx;
        oy = cy;
    }
}
