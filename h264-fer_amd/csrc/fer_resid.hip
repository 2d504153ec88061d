// fer_resid.hip -- residual path of inter macroblocks (rows a1, a2, a4-a9 of SURVEY.md 8a):
// quantizationTransform(predL, predCb, predCr, reconstruct = true) for P macroblocks,
// F/quantizationTransform.cpp:349-486, with the picture construction of
// F/inttransform.cpp:133-155,237-320 and setCodedBlockPattern of F/rbsp_encoding.cpp:21-105.
//
// The same wavefront first finishes the motion decision of its macroblock -- partition merge, mvd and the
// source snapping of F/moestimation.cpp:529-584 -- because that needs exactly the prediction the residual
// needs; the snapped source never goes back to memory (the reconstruction overwrites it anyway).
//
// One wavefront per macroblock.  All 64 lanes rebuild the prediction (motion compensation,
// 4 luma + 2 chroma samples each) into LDS; then lane b < 16 owns luma block b, lanes 16..19
// the Cb blocks and 20..23 the Cr blocks: difference, forward core, quantiser, zig-zag,
// dequantiser, inverse core and clipped reconstruction all stay in that lane's registers.
// The 2x2 chroma DC Hadamard crosses lanes with shuffles.
#include "fer_internal.h"
#include "fer_mvpred.h"

__global__ __launch_bounds__(64) void k_p_resid(FerDev d)
{
    __shared__ uint8_t pL[16][16], pC[2][8][8];  // prediction
    __shared__ uint8_t sL[16][16], sC[2][8][8];  // source after snapping
    const int lane = threadIdx.x;
    const int s = blockIdx.y, mb = blockIdx.x;
    if (d.hdr[s * 4 + 3] != 0) return;
    int *mbt = d.mb_type + (size_t)s * d.nmb;
    if (mbt[mb] == FER_P_SKIP) return;  // reconstructed by k_me_resolve
    const int W = d.W, H = d.H, Wc = d.Wc, Hc = d.Hc;
    uint8_t *Y = d.curY + (size_t)s * d.ysz;
    uint8_t *C0 = d.curCb + (size_t)s * d.csz, *C1 = d.curCr + (size_t)s * d.csz;
    const uint8_t *RY = d.refY + (size_t)s * d.ysz;
    const uint8_t *RC0 = d.refCb + (size_t)s * d.csz, *RC1 = d.refCr + (size_t)s * d.csz;
    const short *mvs = d.mv + (size_t)s * d.nmb * 8;
    const int xp = (mb % d.mbw) << 4, yp = (mb / d.mbw) << 4;

    // ---- partition merge and mvd under the final type (F/moestimation.cpp:529-560); every vector of the
    // picture is final here, so nothing below is read by another macroblock's decision
    int mvx[4], mvy[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        mvx[i] = mvs[(mb * 4 + i) * 2];
        mvy[i] = mvs[(mb * 4 + i) * 2 + 1];
    }
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
    {
        MvCtx c;
        c.mv = mvs;
        c.mb_type = nullptr;
        c.mbw = d.mbw;
        c.cur = mb;
        c.type = type;
        c.coh = false;
        int np = type == FER_P_L0_16x16 ? 1 : (type == FER_P_8x8ref0 ? 4 : 2);
        int dvx[4] = {0, 0, 0, 0}, dvy[4] = {0, 0, 0, 0};
        for (int i = 0; i < np; i++) {
            int q = (type == FER_P_16x8 && i == 1) ? 2 : i;  // quadrant that carries partition i's vector
            int px_, py_;
            predict_luma(c, i, px_, py_);
            dvx[i] = mvx[q] - px_;
            dvy[i] = mvy[q] - py_;
        }
        if (lane < 4) {
            short *o = d.mvd + ((size_t)s * d.nmb + mb) * 8;
            o[lane * 2] = (short)dvx[lane];
            o[lane * 2 + 1] = (short)dvy[lane];
        }
        if (lane == 0) {
            mbt[mb] = type;
            atomicAdd(&d.stats[s * 5 + stat], 1);
        }
    }
    // ---- prediction and source snapping (F/moestimation.cpp:561-584): a source sample within MAXDIFF of the
    // prediction is replaced by it
    {
        int lx = (lane & 3) * 4, ly = lane >> 2;
        int srcv[4];
        uint32_t sv = *(const uint32_t *)(Y + (size_t)(yp + ly) * W + xp + lx);
#pragma unroll
        for (int k = 0; k < 4; k++) srcv[k] = (sv >> (8 * k)) & 0xff;
        int MAXDIFF = d.maxdiff_set;
        if (d.maxdiff_set == -1) {
            int mean = wave_sum(srcv[0] + srcv[1] + srcv[2] + srcv[3]) / 256;
            int dev = wave_sum(iabs(srcv[0] - mean) + iabs(srcv[1] - mean) + iabs(srcv[2] - mean) + iabs(srcv[3] - mean));
            MAXDIFF = dev / 256;
            if (MAXDIFF < 3) MAXDIFF = 3;
        }
        int q = (ly >> 3) * 2 + (lx >> 3);
        int pf[4];
        mc_luma4(RY, ip_stream(d, s), W, H, xp, yp, lx, ly, mvx[q], mvy[q], pf);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            pL[ly][lx + k] = (uint8_t)pf[k];
            sL[ly][lx + k] = (uint8_t)(iabs(srcv[k] - pf[k]) < MAXDIFF ? pf[k] : srcv[k]);
        }
        int cx = lane & 7, cy = lane >> 3;
        int qc = (cy >> 2) * 2 + (cx >> 2);
        size_t co = (size_t)(yp / 2 + cy) * Wc + xp / 2 + cx;
        int pb = mc_chroma(RC0, Wc, Hc, xp / 2, yp / 2, cx, cy, mvx[qc], mvy[qc]);
        int pr = mc_chroma(RC1, Wc, Hc, xp / 2, yp / 2, cx, cy, mvx[qc], mvy[qc]);
        int sb = C0[co], sr = C1[co];
        pC[0][cy][cx] = (uint8_t)pb;
        pC[1][cy][cx] = (uint8_t)pr;
        sC[0][cy][cx] = (uint8_t)(iabs(sb - pb) <= MAXDIFF ? pb : sb);
        sC[1][cy][cx] = (uint8_t)(iabs(sr - pr) <= MAXDIFF ? pr : sr);
    }
    __syncthreads();

    int16_t *lv = d.levels + ((size_t)s * d.nmb + mb) * FER_LEVELS;
    const bool isL = lane < 16, isC = lane >= 16 && lane < 24;
    const int pl = (lane - 16) >> 2, cb = (lane - 16) & 3;  // chroma plane / block
    int x0 = 0, y0 = 0, stride = W;
    uint8_t *dst = Y;
    const uint8_t *prd = &pL[0][0], *srd = &sL[0][0];
    int pstride = 16;
    if (isL) {
        x0 = c_bx[lane];
        y0 = c_by[lane];
        dst = Y + (size_t)(yp + y0) * W + xp + x0;
        prd = &pL[y0][x0];
        srd = &sL[y0][x0];
    } else if (isC) {
        x0 = (cb & 1) * 4;
        y0 = (cb >> 1) * 4;
        stride = Wc;
        dst = (pl ? C1 : C0) + (size_t)(yp / 2 + y0) * Wc + xp / 2 + x0;
        prd = &pC[pl][y0][x0];
        srd = &sC[pl][y0][x0];
        pstride = 8;
    }
    int r[16], t[16], q[16], p[16];
    int nz = 0, dcraw = 0;
    const int qP = isL ? d.qp : d.qpc;
    if (isL || isC) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            p[i] = prd[(i >> 2) * pstride + (i & 3)];
            r[i] = (int)srd[(i >> 2) * pstride + (i & 3)] - p[i];
        }
        fwd4x4(r, t);
        quant4x4(t, q, qP, isC);
        dcraw = q[0];
    }
    // chroma DC: 2x2 Hadamard + quantiser on lanes 16 and 20, then back to the block lanes
    int dcq = 0, dcdeq = 0;
    {
        int base = 16 + (lane >= 20 ? 4 : 0);
        int f[4], cq[4], dq[4];
#pragma unroll
        for (int i = 0; i < 4; i++) f[i] = __shfl(dcraw, base + i);
        fwd_dc_chroma(f, cq, d.qpc);
        inv_dc_chroma(cq, dq, d.qpc);
        if (isC) {
            dcq = cq[cb];
            dcdeq = dq[cb];
        }
    }
    int cnt = 0;
    if (isL) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            int v = q[c_zz[k]];
            lv[lane * 16 + k] = (int16_t)v;
            cnt += v != 0;
        }
        nz = cnt != 0;
        inv4x4(q, r, qP, false);
    } else if (isC) {
        lv[FER_LV_CDC + pl * 4 + cb] = (int16_t)dcq;
#pragma unroll
        for (int k = 1; k < 16; k++) {
            int v = q[c_zz[k]];
            lv[FER_LV_CAC + (pl * 4 + cb) * 15 + k - 1] = (int16_t)v;
            cnt += v != 0;
        }
        nz = cnt != 0;
        q[0] = dcdeq;
        inv4x4(q, r, qP, true);
    }
    if (isL || isC) {
#pragma unroll
        for (int i = 0; i < 16; i++) dst[(size_t)(i >> 2) * stride + (i & 3)] = (uint8_t)clip255(p[i] + r[i]);
        d.tc[((size_t)s * d.nmb + mb) * 24 + lane] = (uint8_t)cnt;
    }
    // coded block pattern
    unsigned long long nzm = __ballot(nz != 0);
    unsigned long long dcm = __ballot(isC && dcq != 0);
    if (lane == 0) {
        int l = 0;
        for (int i8 = 0; i8 < 4; i8++)
            if ((nzm >> (i8 * 4)) & 0xf) l |= 1 << i8;
        int ch = 0;
        if (dcm) ch |= 1;
        if ((nzm >> 16) & 0xff) ch |= 2;
        if (ch == 3) ch = 2;
        d.cbp[((size_t)s * d.nmb + mb) * 2] = (uint8_t)l;
        d.cbp[((size_t)s * d.nmb + mb) * 2 + 1] = (uint8_t)ch;
    }
}

void fer_launch_p_resid(const FerDev &d, hipStream_t st)
{
    hipLaunchKernelGGL(k_p_resid, dim3(d.nmb, d.S), dim3(64), 0, st, d);
}
