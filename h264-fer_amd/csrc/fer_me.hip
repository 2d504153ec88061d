// fer_me.hip -- P-macroblock decision (rows a15, a18 of SURVEY.md 8a) on CDNA4.
//
// The reference's interEncoding (F/moestimation.cpp:392-585) is serial over macroblocks
// because every search is centred on, and costed against, the predicted motion vector of
// the partition (F/mode_pred.cpp:252-371).  The work is split here into
//   k_me_pre      neighbour-independent: per 8x8 partition the five box sums of the source
//                 block, the complete stage-3 search (wide integer + local quarter-pel window
//                 around 0, top-33 list, SAD of each survivor) and the stage-2 candidate set
//                 (bucket walk over the sum-sorted positions) -- one wavefront per partition;
//   k_me_resolve  neighbour-dependent: P_Skip test, stage-1 search around the predictor,
//                 re-ranking of the stage-2 candidates, final costs, partition merge, mvd,
//                 final motion compensation and source snapping -- one wavefront per
//                 macroblock, launched once per anti-diagonal x + 2y of the MB grid.
// One wavefront owns one macroblock; the candidate list of MEstimation
// (F/moestimation.cpp:254-296) lives one slot per lane and is updated by ballot-ordered
// insertion, which reproduces the reference's arrival-order tie breaking exactly.
#include "fer_internal.h"

#define INF_M 1000000000

struct WList {
    int m;   // metric of slot == lane
    int xy;  // (bx & 0xffff) | by << 16
};

__device__ __forceinline__ int pack_xy(int x, int y) { return (x & 0xffff) | (y << 16); }
__device__ __forceinline__ int unp_x(int xy) { return (int)(short)(xy & 0xffff); }
__device__ __forceinline__ int unp_y(int xy) { return xy >> 16; }

// insert the valid candidates of this batch in lane (= arrival) order; list keeps K best
__device__ __forceinline__ void wl_insert(WList &L, int K, int lane, bool valid, int m, int xy)
{
    int thr = lane_bcast(L.m, K - 1);
    unsigned long long mask = __ballot(valid && m < thr);
    while (mask) {
        int src = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        int cm = lane_bcast(m, src), cxy = lane_bcast(xy, src);
        int pos = __popcll(__ballot(lane < K && L.m <= cm));
        if (pos < K) {
            int um = FER_DPP(L.m, DPP_WAVE_SHR1), uxy = FER_DPP(L.xy, DPP_WAVE_SHR1);
            if (lane > pos && lane < K) {
                L.m = um;
                L.xy = uxy;
            } else if (lane == pos) {
                L.m = cm;
                L.xy = cxy;
            }
        }
    }
}

// the 9-term feature distance of F/moestimation.cpp:267-276 at (frac, refy, refx)
__device__ __forceinline__ int feat_dist(const uint16_t *__restrict__ Fs, size_t ysz, int W, int frac, int refy,
                                         int refx, const int s[5])
{
    const uint16_t *F = Fs + (size_t)frac * 5 * ysz + (size_t)refy * W + refx;
    int k0 = F[0], k1 = F[ysz], k2 = F[2 * ysz], k3 = F[3 * ysz], k4 = F[4 * ysz];
    return iabs(s[0] - k0) + iabs(s[1] - k1) + iabs(s[0] - s[1] - k0 + k1) + iabs(s[2] - k2) +
           iabs(s[0] - s[2] - k0 + k2) + iabs(s[3] - k3) + iabs(s[0] - s[3] - k0 + k3) + iabs(s[4] - k4) +
           iabs(s[0] - s[4] - k0 + k4);
}

// SAD of the 8x8 source block against interpolated plane (F/moestimation.cpp:175-195).
// 8 lanes (rows) per candidate, 8 candidates per call: lane = cand*8 + row.  Returns the
// full SAD in every lane of the group.
__device__ __forceinline__ int sad8_rows(const uint8_t *__restrict__ Ps, size_t ysz, int W, int H, int xP, int yP,
                                         int mvx, int mvy, int row, const int src[8])
{
    int xPi = iclamp(xP + (mvx >> 2), 0, W - 1), yPi = iclamp(yP + (mvy >> 2), 0, H - 1);
    const uint8_t *R = Ps + (size_t)((mvx & 3) + (mvy & 3) * 4) * ysz + (size_t)min(yPi + row, H - 1) * W;
    int s = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) s += iabs(src[j] - (int)R[min(xPi + j, W - 1)]);
    return oct_sum(s);
}

// ------------------------------------------------------------------ k_me_pre
__global__ __launch_bounds__(64) void k_me_pre(FerDev d)
{
    const int lane = threadIdx.x;
    const int s = blockIdx.y;
    if (d.hdr[s * 4 + 3] != 0) return;
    const int mb = blockIdx.x >> 2, part = blockIdx.x & 3;
    const int W = d.W, H = d.H;
    const size_t ysz = d.ysz;
    const uint8_t *Y = d.curY + (size_t)s * ysz;
    const uint8_t *Ps = d.interp + (size_t)s * 16 * ysz;
    const uint16_t *Fs = d.feat + (size_t)s * 80 * ysz;
    const int sx = ((mb % d.mbw) << 4) + (part & 1) * 8, sy = ((mb / d.mbw) << 4) + (part >> 1) * 8;
    const size_t pidx = ((size_t)s * d.nmb + mb) * 4 + part;

    // box sums of the source block, F/moestimation.cpp:440-451
    int px = lane & 7, py = lane >> 3;
    int v = Y[(size_t)(sy + py) * W + sx + px];
    int su[5];
    su[0] = wave_sum(v);
    su[1] = wave_sum(py > 3 ? 0 : v);
    su[2] = wave_sum(px > 3 ? 0 : v);
    su[3] = wave_sum((py & 3) > 1 ? 0 : v);
    su[4] = wave_sum((px & 3) > 1 ? 0 : v);
    if (lane < 5) d.suma[pidx * 5 + lane] = su[lane];

    // source rows for the SAD groups
    int row = lane & 7, src[8];
#pragma unroll
    for (int j = 0; j < 8; j++) src[j] = Y[(size_t)(sy + row) * W + sx + j];

    // ---- stage 3: MEstimation(+-W/2, frac 0, centre 0) then MEstimation(+-W/16, 16 fracs, centre 0)
    WList L;
    L.m = INF_M;
    L.xy = 0;
    const int R = d.window / 2, n = 2 * R + 1;
    for (int base = 0; base < n * n; base += 64) {
        int c = base + lane;
        int tx = c / n - R, ty = c % n - R;
        int rx = sx + tx, ry = sy + ty;
        bool ok = c < n * n && rx >= 0 && rx < W && ry >= 0 && ry < H;
        int m = 0;
        if (ok) m = (iabs(tx) + iabs(ty) + 4) * feat_dist(Fs, ysz, W, 0, ry, rx, su);
        wl_insert(L, 33, lane, ok, m, pack_xy(tx * 4, ty * 4));
    }
    const int r2 = d.window / 16, n2w = 2 * r2 + 1;
    for (int base = 0; base < n2w * n2w * 16; base += 64) {
        int c = base + lane;
        int frac = c & 15, pos = c >> 4;
        int tx = pos / n2w - r2, ty = pos % n2w - r2;
        int rx = sx + tx, ry = sy + ty;
        bool ok = c < n2w * n2w * 16 && rx >= 0 && rx < W && ry >= 0 && ry < H;
        int m = 0;
        if (ok) m = (iabs(tx) + iabs(ty) + 4) * feat_dist(Fs, ysz, W, frac, ry, rx, su);
        wl_insert(L, 33, lane, ok, m, pack_xy(tx * 4 + (frac & 3), ty * 4 + (frac >> 2)));
    }
    int n3 = __popcll(__ballot(lane < 33 && L.m < 100000000));
    for (int base = 0; base < n3; base += 8) {
        int j = base + (lane >> 3);
        int xy = __shfl(L.xy, j < 33 ? j : 0);
        int bx = unp_x(xy), by = unp_y(xy);
        int sad = sad8_rows(Ps, ysz, W, H, sx, sy, bx, by, row, src);
        if (j < n3 && row == 0) {
            int *o = d.st3 + (pidx * 33 + j) * 3;
            o[0] = bx;
            o[1] = by;
            o[2] = sad;
        }
    }
    if (lane == 0) d.st3n[pidx] = n3;

    // ---- stage 2 candidate set: bucket walk of F/moestimation.cpp:470-496 (weight applied later)
    const int *kol = d.koliko + (size_t)s * 16385;
    const uint32_t *spos = d.sort_pos + (size_t)s * ysz;
    const uint32_t *sk12 = d.sort_k12 + (size_t)s * ysz;
    const uint32_t *sk34 = d.sort_k34 + (size_t)s * ysz;
    int tren = 0;
    if (!d.basic) {
        for (int j = 0; j <= 180; j++) {
            for (int side = 0; side < 2; side++) {
                int a = side ? su[0] + j : su[0] - j;
                if (a < 0 || a >= 16384) continue;
                int k0 = kol[a], k1 = kol[a + 1];
                for (int base = k0; base < k1; base += 64) {
                    int k = base + lane;
                    bool ok = false;
                    int tx = 0, ty = 0, D = 0;
                    if (k < k1) {
                        uint32_t p = spos[k], q = sk12[k];
                        int ax = (int)(p >> 16), ay = (int)(p & 0xffff);
                        int q1 = (int)(q & 0xffff), q2 = (int)(q >> 16);
                        tx = ax - sx;
                        ty = ay - sy;
                        ok = iabs(tx) + iabs(ty) < 280 && iabs(q1 - su[1]) < 100 && iabs(q2 - su[2]) < 100;
                        if (ok) {  // feature distance from the sorted payload: k0 == a, no scattered reads
                            uint32_t r = sk34[k];
                            int q3 = (int)(r & 0xffff), q4 = (int)(r >> 16);
                            D = iabs(su[0] - a) + iabs(su[1] - q1) + iabs(su[0] - su[1] - a + q1) + iabs(su[2] - q2) +
                                iabs(su[0] - su[2] - a + q2) + iabs(su[3] - q3) + iabs(su[0] - su[3] - a + q3) +
                                iabs(su[4] - q4) + iabs(su[0] - su[4] - a + q4);
                        }
                    }
                    unsigned long long mk = __ballot(ok);
                    int rank = tren + __popcll(mk & ((1ull << lane) - 1));
                    if (ok && rank < FER_ST2_CAP) {
                        int *o = d.st2 + (pidx * FER_ST2_CAP + rank) * 2;
                        o[0] = pack_xy(tx, ty);
                        o[1] = D;
                    }
                    tren += __popcll(mk);
                }
            }
            if (tren > 128) break;
        }
    }
    if (lane == 0) {
        d.st2n[pidx] = tren;
        if (tren > FER_ST2_CAP) atomicOr(&d.status[s], FER_ERR_ST2_OVERFLOW);
    }
}

// ------------------------------------------------------------------ MV prediction (a18)
struct MvCtx {
    const int *mb_type;  // stream base
    const short *mv;     // stream base [nmb][4][2]
    int mbw, cur, type;  // type = current mb_type
    int cx[4], cy[4];    // quadrant MVs of the current MB as far as known
};

__device__ __forceinline__ int p_part_w(int t) { return (t == 0 || t == 1 || t == FER_P_SKIP) ? 16 : 8; }
__device__ __forceinline__ int p_part_h(int t) { return (t == 0 || t == 2 || t == FER_P_SKIP) ? 16 : 8; }

// neighbour location + motion vector, F/mode_pred.cpp:49-110 (all MBs of a P picture are inter here)
__device__ void nbr_fetch(const MvCtx &c, int xN, int yN, bool &valid, int &mx, int &my)
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
    int t = (mbN == cur) ? c.type : c.mb_type[mbN];
    int part = ((yW / p_part_h(t)) << 1) + (xW / p_part_w(t));
    if (mbN == cur) {
        mx = c.cx[part];
        my = c.cy[part];
    } else {
        mx = c.mv[((size_t)mbN * 4 + part) * 2];
        my = c.mv[((size_t)mbN * 4 + part) * 2 + 1];
    }
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
    int dxm = FER_MV_NA, dym = FER_MV_NA;
    for (int i = 0; i < 3; i++) {
        mx[i] = my[i] = FER_MV_NA;
        ref[i] = -1;
    }
    nbr_fetch(c, x - 1, y, val[0], mx[0], my[0]);
    nbr_fetch(c, x, y - 1, val[1], mx[1], my[1]);
    nbr_fetch(c, x + ppw, y - 1, val[2], mx[2], my[2]);
    if (!val[2]) {
        nbr_fetch(c, x - 1, y - 1, val[3], dxm, dym);
        val[2] = val[3];
        mx[2] = dxm;
        my[2] = dym;
    }
    for (int i = 0; i < 3; i++)
        if (val[i]) ref[i] = 0;
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

// ------------------------------------------------------------------ k_me_resolve
__global__ __launch_bounds__(64) void k_me_resolve(FerDev d, int diag)
{
    const int lane = threadIdx.x;
    const int s = blockIdx.y;
    if (d.hdr[s * 4 + 3] != 0) return;
    // k-th macroblock on the anti-diagonal mbx + 2*mby == diag
    int y_lo = diag - (d.mbw - 1);
    y_lo = y_lo > 0 ? (y_lo + 1) >> 1 : 0;
    int mby = y_lo + blockIdx.x, mbx = diag - 2 * mby;
    if (mby >= d.mbh || mbx < 0 || mbx >= d.mbw) return;
    const int mb = mby * d.mbw + mbx;
    const int W = d.W, H = d.H, Wc = d.Wc, Hc = d.Hc;
    const size_t ysz = d.ysz, csz = d.csz;
    uint8_t *Y = d.curY + (size_t)s * ysz;
    uint8_t *Cb = d.curCb + (size_t)s * csz, *Cr = d.curCr + (size_t)s * csz;
    const uint8_t *RY = d.refY + (size_t)s * ysz;
    const uint8_t *RCb = d.refCb + (size_t)s * csz, *RCr = d.refCr + (size_t)s * csz;
    const uint8_t *Ps = d.interp + (size_t)s * 16 * ysz;
    const uint16_t *Fs = d.feat + (size_t)s * 80 * ysz;
    int *mbt = d.mb_type + (size_t)s * d.nmb;
    short *mvs = d.mv + (size_t)s * d.nmb * 8;
    const int xp = mbx << 4, yp = mby << 4;

    MvCtx c;
    c.mb_type = mbt;
    c.mv = mvs;
    c.mbw = d.mbw;
    c.cur = mb;
    c.type = FER_P_SKIP;
    for (int i = 0; i < 4; i++) c.cx[i] = c.cy[i] = 0;

    // ---- P_Skip candidate, F/mode_pred.cpp:381-402 + F/moestimation.cpp:402-425
    int smx = 0, smy = 0;
    if (!(mb < d.mbw || mbx == 0)) {
        int up = mb - d.mbw, lf = mb - 1;
        bool zu = (mvs[(up * 4 + 2) * 2] | mvs[(up * 4 + 2) * 2 + 1]) == 0;
        bool zl = (mvs[(lf * 4 + 1) * 2] | mvs[(lf * 4 + 1) * 2 + 1]) == 0;
        if (!(zu || zl)) predict_luma(c, 0, smx, smy);
    }
    // each lane owns 4 luma samples: x = (lane&3)*4.., y = lane>>2 .. wait 16 rows: lane>>2 in 0..15
    const int lx = (lane & 3) * 4, ly = lane >> 2;
    int srcv[4], pred[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        srcv[k] = Y[(size_t)(yp + ly) * W + xp + lx + k];
        pred[k] = mc_luma(RY, W, H, xp, yp, lx + k, ly, smx, smy);
    }
    int MAXDIFF = d.maxdiff_set;
    if (d.maxdiff_set == -1) {  // adaptive tolerance, F/moestimation.cpp:407-419
        int mean = wave_sum(srcv[0] + srcv[1] + srcv[2] + srcv[3]) / 256;
        int dev = wave_sum(iabs(srcv[0] - mean) + iabs(srcv[1] - mean) + iabs(srcv[2] - mean) + iabs(srcv[3] - mean));
        MAXDIFF = dev / 256;
        if (MAXDIFF < 3) MAXDIFF = 3;
    }
    bool exact = true;
#pragma unroll
    for (int k = 0; k < 4; k++) exact = exact && iabs(srcv[k] - pred[k]) <= MAXDIFF;
    // chroma sample owned by the lane: (lane&7, lane>>3) of both planes
    const int cxl = lane & 7, cyl = lane >> 3;
    if (__all(exact)) {
        // P_Skip: reconstruction == prediction (F/inttransform.cpp:215-231)
#pragma unroll
        for (int k = 0; k < 4; k++) Y[(size_t)(yp + ly) * W + xp + lx + k] = (uint8_t)pred[k];
        Cb[(size_t)(yp / 2 + cyl) * Wc + xp / 2 + cxl] = (uint8_t)mc_chroma(RCb, Wc, Hc, xp / 2, yp / 2, cxl, cyl, smx, smy);
        Cr[(size_t)(yp / 2 + cyl) * Wc + xp / 2 + cxl] = (uint8_t)mc_chroma(RCr, Wc, Hc, xp / 2, yp / 2, cxl, cyl, smx, smy);
        if (lane < 4) {
            mvs[(mb * 4 + lane) * 2] = (short)smx;
            mvs[(mb * 4 + lane) * 2 + 1] = (short)smy;
        }
        if (lane == 0) {
            mbt[mb] = FER_P_SKIP;
            atomicAdd(&d.stats[s * 5 + 0], 1);
        }
        return;
    }

    // ---- four 8x8 partitions as P_8x8ref0
    c.type = FER_P_8x8ref0;
    int mvx[4], mvy[4];
    const int row = lane & 7;
    for (int i = 0; i < 4; i++) {
        int mvpx, mvpy;
        predict_luma(c, i, mvpx, mvpy);
        const int genx = mvpx >> 2, geny = mvpy >> 2;
        const int sx = xp + (i & 1) * 8, sy = yp + (i >> 1) * 8;
        const size_t pidx = ((size_t)s * d.nmb + mb) * 4 + i;
        int su[5];
#pragma unroll
        for (int k = 0; k < 5; k++) su[k] = d.suma[pidx * 5 + k];
        int src[8];
#pragma unroll
        for (int j = 0; j < 8; j++) src[j] = Y[(size_t)(sy + row) * W + sx + j];
        int bx = 0, by = 0, bmin = 2000000000;

        // stage 1: +-W/16 around the predictor, all 16 fractional planes (K = 17)
        WList L;
        L.m = INF_M;
        L.xy = 0;
        const int r1 = d.window / 16, n1 = 2 * r1 + 1;
        for (int base = 0; base < n1 * n1 * 16; base += 64) {
            int cc = base + lane;
            int frac = cc & 15, pos = cc >> 4;
            int tx = genx - r1 + pos / n1, ty = geny - r1 + pos % n1;
            int rx = sx + tx, ry = sy + ty;
            bool ok = cc < n1 * n1 * 16 && rx >= 0 && rx < W && ry >= 0 && ry < H;
            int m = 0;
            if (ok) m = (iabs(tx - genx) + iabs(ty - geny) + 4) * feat_dist(Fs, ysz, W, frac, ry, rx, su);
            wl_insert(L, 17, lane, ok, m, pack_xy(tx * 4 + (frac & 3), ty * 4 + (frac >> 2)));
        }
        int cnt = __popcll(__ballot(lane < 17 && L.m < 100000000));
        for (int base = 0; base < cnt; base += 8) {
            int j = base + (lane >> 3);
            int xy = __shfl(L.xy, j < 17 ? j : 0);
            int cxv = unp_x(xy), cyv = unp_y(xy);
            int sad = sad8_rows(Ps, ysz, W, H, sx, sy, cxv, cyv, row, src);
            int cost = j < cnt ? sad + iabs(cxv - mvpx) + iabs(cyv - mvpy) : 2000000000;
            // ordered first-minimum over the 8 candidates of this round
            for (int g = 0; g < 8; g++) {
                int cg = lane_bcast(cost, g * 8), xg = lane_bcast(cxv, g * 8), yg = lane_bcast(cyv, g * 8);
                if (cg < bmin) {
                    bmin = cg;
                    bx = xg;
                    by = yg;
                }
            }
        }
        if (!d.basic) {
            // stage 2: re-rank the precomputed candidate set with the predictor weight (K = 33)
            L.m = INF_M;
            int n2 = min(d.st2n[pidx], FER_ST2_CAP);
            const int *c2 = d.st2 + pidx * FER_ST2_CAP * 2;
            for (int base = 0; base < n2; base += 64) {
                int cc = base + lane;
                bool ok = cc < n2;
                int m = 0, xy = 0;
                if (ok) {
                    int pxy = c2[cc * 2], D = c2[cc * 2 + 1];
                    int tx = unp_x(pxy), ty = unp_y(pxy);
                    m = (iabs(tx - genx) + iabs(ty - geny) + 4) * D;
                    xy = pack_xy(tx * 4, ty * 4);
                }
                wl_insert(L, 33, lane, ok, m, xy);
            }
            cnt = __popcll(__ballot(lane < 33 && L.m < 100000000));
            for (int base = 0; base < cnt; base += 8) {
                int j = base + (lane >> 3);
                int xy = __shfl(L.xy, j < 33 ? j : 0);
                int cxv = unp_x(xy), cyv = unp_y(xy);
                int sad = sad8_rows(Ps, ysz, W, H, sx, sy, cxv, cyv, row, src);
                int cost = j < cnt ? sad + iabs(cxv - mvpx) + iabs(cyv - mvpy) : 2000000000;
                for (int g = 0; g < 8; g++) {
                    int cg = lane_bcast(cost, g * 8), xg = lane_bcast(cxv, g * 8), yg = lane_bcast(cyv, g * 8);
                    if (cg < bmin) {
                        bmin = cg;
                        bx = xg;
                        by = yg;
                    }
                }
            }
            // stage 3: precomputed survivors of the centre-0 searches
            int n3 = d.st3n[pidx];
            const int *c3 = d.st3 + pidx * 33 * 3;
            int cost = 2000000000, cxv = 0, cyv = 0;
            if (lane < n3) {
                cxv = c3[lane * 3];
                cyv = c3[lane * 3 + 1];
                cost = c3[lane * 3 + 2] + iabs(cxv - mvpx) + iabs(cyv - mvpy);
            }
            for (int g = 0; g < n3; g++) {
                int cg = lane_bcast(cost, g), xg = lane_bcast(cxv, g), yg = lane_bcast(cyv, g);
                if (cg < bmin) {
                    bmin = cg;
                    bx = xg;
                    by = yg;
                }
            }
        }
        mvx[i] = bx;
        mvy[i] = by;
        c.cx[i] = bx;
        c.cy[i] = by;
    }

    // ---- partition merge, F/moestimation.cpp:529-551
    int type = FER_P_8x8ref0, stat = 4;
    if (mvx[0] == mvx[1] && mvx[0] == mvx[2] && mvx[0] == mvx[3] && mvy[0] == mvy[1] && mvy[0] == mvy[2] &&
        mvy[0] == mvy[3]) {
        type = FER_P_L0_16x16;
        stat = 1;
    } else if (mvx[0] == mvx[1] && mvx[2] == mvx[3] && mvy[0] == mvy[1] && mvy[2] == mvy[3]) {
        type = FER_P_16x8;
        stat = 2;
    } else if (mvx[0] == mvx[2] && mvx[1] == mvx[3] && mvy[0] == mvy[2] && mvy[1] == mvy[3]) {
        type = FER_P_8x16;
        stat = 3;
    }
    // mvd under the final type, F/moestimation.cpp:552-564
    c.type = type;
    int np = type == FER_P_L0_16x16 ? 1 : (type == FER_P_8x8ref0 ? 4 : 2);
    int dvx[4] = {0, 0, 0, 0}, dvy[4] = {0, 0, 0, 0};
    for (int i = 0; i < np; i++) {
        int q = i;  // quadrant that carries partition i's vector
        if (type == FER_P_16x8 && i == 1) q = 2;
        int px_, py_;
        predict_luma(c, i, px_, py_);
        dvx[i] = mvx[q] - px_;
        dvy[i] = mvy[q] - py_;
    }
    if (lane < 4) {
        mvs[(mb * 4 + lane) * 2] = (short)mvx[lane];
        mvs[(mb * 4 + lane) * 2 + 1] = (short)mvy[lane];
        short *o = d.mvd + ((size_t)s * d.nmb + mb) * 8;
        o[lane * 2] = (short)dvx[lane];
        o[lane * 2 + 1] = (short)dvy[lane];
    }
    if (lane == 0) {
        mbt[mb] = type;
        atomicAdd(&d.stats[s * 5 + stat], 1);
    }

    // ---- final prediction and source snapping, F/moestimation.cpp:565-584
    {
        int q = (ly >> 3) * 2 + (lx >> 3);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int p = mc_luma(RY, W, H, xp, yp, lx + k, ly, mvx[q], mvy[q]);
            if (iabs(srcv[k] - p) < MAXDIFF) Y[(size_t)(yp + ly) * W + xp + lx + k] = (uint8_t)p;
        }
        int qc = (cyl >> 2) * 2 + (cxl >> 2);
        size_t co = (size_t)(yp / 2 + cyl) * Wc + xp / 2 + cxl;
        int pb = mc_chroma(RCb, Wc, Hc, xp / 2, yp / 2, cxl, cyl, mvx[qc], mvy[qc]);
        int pr = mc_chroma(RCr, Wc, Hc, xp / 2, yp / 2, cxl, cyl, mvx[qc], mvy[qc]);
        if (iabs((int)Cb[co] - pb) <= MAXDIFF) Cb[co] = (uint8_t)pb;
        if (iabs((int)Cr[co] - pr) <= MAXDIFF) Cr[co] = (uint8_t)pr;
    }
}

void fer_launch_me_pre(const FerDev &d, hipStream_t st)
{
    hipLaunchKernelGGL(k_me_pre, dim3(d.nmb * 4, d.S), dim3(64), 0, st, d);
}

void fer_launch_me_resolve(const FerDev &d, hipStream_t st)
{
    int ndiag = d.mbw + 2 * (d.mbh - 1);
    int maxk = min(d.mbh, (d.mbw + 1) / 2);
    for (int dg = 0; dg < ndiag; dg++) hipLaunchKernelGGL(k_me_resolve, dim3(maxk, d.S), dim3(64), 0, st, d, dg);
}
