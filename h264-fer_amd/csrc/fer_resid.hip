// fer_resid.hip -- residual path of inter macroblocks (rows a1, a2, a4-a9 of SURVEY.md 8a):
// quantizationTransform(predL, predCb, predCr, reconstruct = true) for P macroblocks,
// F/quantizationTransform.cpp:349-486, with the picture construction of
// F/inttransform.cpp:133-155,237-320 and setCodedBlockPattern of F/rbsp_encoding.cpp:21-105.
//
// The same wavefront first finishes the motion decision of its macroblock -- partition merge, mvd and the
// source snapping of F/moestimation.cpp:529-584 -- because that needs exactly the prediction the residual
// needs; the snapped source never goes back to memory (the reconstruction overwrites it anyway).
//
// One wavefront per macroblock, four lanes per 4x4 block: lane = block * 4 + row holds one ROW of its block (four
// samples in registers) from the prediction to the reconstruction.  The vertical half of every transform crosses the
// four lanes of a quad with DPP quad_perm broadcasts, the horizontal half stays in the lane; all 64 lanes carry the 16
// luma blocks at once, a second pass carries the 8 chroma blocks on lanes 0..31.  Prediction, source and
// reconstruction move as one dword per lane; the levels are put in scan order in LDS and leave as one coalesced
// 800-byte record.  The 2x2 chroma DC Hadamard runs on values read out of the lanes with v_readlane.
#include "fer_internal.h"
#include "fer_mvpred.h"

// A workgroup = four wavefronts = four macroblocks side by side (64-byte picture segments), and the workgroups are
// dealt so that each XCD works on one contiguous band of the picture: a picture line is then fetched into ONE L2, once,
// instead of by every XCD that happens to own one of its 16-sample pieces.  The wavefronts share nothing: their LDS
// is private and ordered with wave-level fences, no s_barrier.
#define RES_SYNC()                                             \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
    } while (0)
#ifndef RESID_WAVES
#define RESID_WAVES 6  // 74 registers, no scratch; the kernel is a chain of memory round trips and gains from the sixth wavefront
#endif
#ifndef RES_MBW
#define RES_MBW 4  // macroblocks (wavefronts) of a workgroup, side by side
#endif
__global__ __launch_bounds__(64 * RES_MBW) __attribute__((amdgpu_waves_per_eu(RESID_WAVES, 8))) void k_p_resid(FerDev d)
{
    __shared__ __align__(16) int16_t lvs_[RES_MBW][FER_LEVELS];
    __shared__ __align__(4) uint8_t tcs_[RES_MBW][24];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int16_t *lvs = lvs_[wv];
    uint8_t *tcs = tcs_[wv];
    const int s = blockIdx.y, mb = (int)xcd_swizzle(blockIdx.x, gridDim.x) * RES_MBW + wv;
    if (mb >= d.nmb) return;
    const int W = d.W, H = d.H, Wc = d.Wc, Hc = d.Hc;
    int *mbt = d.mb_type + (size_t)s * d.nmb;
    uint8_t *Y = d.curY + (size_t)s * d.ysz;
    const uint8_t *RY = d.refY + (size_t)s * d.ysz;
    const short *mvs = d.mv + (size_t)s * d.nmb * 8;
    const int mbx = mb % d.mbw, mby = mb / d.mbw;
    const int xp = mbx << 4, yp = mby << 4;
    const int row = lane & 3;
    // Everything that does not depend on the vectors is requested before the first test: the kernel is a chain of
    // memory round trips (picture type -> macroblock type -> vectors -> reference samples -> stores), not a stream.
    const int blk = lane >> 2;
    const int lx = c_bx[blk], ly = c_by[blk] + row;
    uint8_t *dstL = Y + (size_t)(yp + ly) * W + xp + lx;
    const int l5 = lane & 31, cblk = l5 >> 2, pl = cblk >> 2, cb = cblk & 3;  // chroma: plane, block
    const int cx0 = (cb & 1) * 4, cy0 = (cb >> 1) * 4 + row;
    uint8_t *Cp = (pl ? d.curCr : d.curCb) + (size_t)s * d.csz;
    const uint8_t *Rp = (pl ? d.refCr : d.refCb) + (size_t)s * d.csz;
    uint8_t *dstC = Cp + (size_t)(yp / 2 + cy0) * Wc + xp / 2 + cx0;
    const int ptype = (int)d.hdr[s * 4 + 3], mtype = mbt[mb];
    const uint32_t svL = *(const uint32_t *)dstL, svC = *(const uint32_t *)dstC;
    // the vectors of this macroblock and of its left / above / above-right / above-left neighbours, one per lane
    int tbl;
    {
        const int which = min(lane >> 2, 4), q = lane & 3;
        const int off = which == 0 ? 0 : (which == 1 ? -1 : (which == 2 ? -d.mbw : (which == 3 ? 1 - d.mbw : -1 - d.mbw)));
        tbl = ((const int *)mvs)[(size_t)iclamp(mb + off, 0, d.nmb - 1) * 4 + q];
    }
    // every vector of the macroblock must carry this picture's serial: a row the motion chain never reached (it cannot
    // happen by construction; ferhip_status bit 6 if it ever does) would otherwise go on with stale vectors silently.
    // Requested here with everything else, tested when the macroblock is done.
    const unsigned chw_hi = (unsigned)(d.chain64[((size_t)s * d.nmb + mb) * 4 + (lane & 3)] >> 32);
    auto chain_check = [&]() {
        if (lane < 4 && (chw_hi & 0x7fffffffu) != ((unsigned)d.serial & 0x7fffffffu)) atomicOr(&d.status[s], FER_ERR_CHAIN_UNRESOLVED);
    };
    if (ptype != 0) return;

    // ---- partition merge and mvd under the final type (F/moestimation.cpp:529-560); every vector of the
    // picture is final here, so nothing below is read by another macroblock's decision
    int mvx[4], mvy[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int w = __builtin_amdgcn_readlane(tbl, i);
        mvx[i] = (int)(short)(w & 0xffff);
        mvy[i] = w >> 16;
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
    auto mv_of = [&](int q, int &vx, int &vy) {  // vector of quadrant q (q differs per lane)
        const int w = __shfl(tbl, q);
        vx = (int)(short)(w & 0xffff);
        vy = w >> 16;
    };
    // the reference samples: luma row of four from the quarter-sample planes, chroma row of four (bilinear)
    int pfL[4], pfC[4];
    {
        int qx, qy;
        mv_of((ly >> 3) * 2 + (lx >> 3), qx, qy);
        mc_luma4(RY, ip_stream(d, s), W, H, xp, yp, lx, ly, qx, qy, pfL);
        mv_of(cb, qx, qy);
        mc_chroma_row4(Rp, Wc, Hc, xp / 2, yp / 2, cx0, cy0, qx, qy, pfC);
    }
    if (mtype == FER_P_SKIP) {
        // P_Skip (decided by k_me_resolve, which left the P_Skip vector in all four quadrants): the reconstruction is
        // the prediction (F/moestimation.cpp:421-425, F/inttransform.cpp:215-231)
        *(uint32_t *)dstL = (uint32_t)pfL[0] | ((uint32_t)pfL[1] << 8) | ((uint32_t)pfL[2] << 16) | ((uint32_t)pfL[3] << 24);
        if (lane < 32) *(uint32_t *)dstC = (uint32_t)pfC[0] | ((uint32_t)pfC[1] << 8) | ((uint32_t)pfC[2] << 16) | ((uint32_t)pfC[3] << 24);
        chain_check();
        return;
    }
    {
        const int np = type == FER_P_L0_16x16 ? 1 : (type == FER_P_8x8ref0 ? 4 : 2);
        const int i = lane & 3;  // lane i < 4 derives partition i: one pass of the predictor for all of them
        const int pi = i < np ? i : 0;
        const int q = (type == FER_P_16x8 && pi == 1) ? 2 : pi;  // quadrant that carries partition i's vector
        int px_, py_, qx, qy;
        predict_luma_tbl(tbl, d.mbw, mbx, mby, type, pi, px_, py_);
        mv_of(q, qx, qy);
        if (lane < 4) {
            const int dvx = i < np ? qx - px_ : 0, dvy = i < np ? qy - py_ : 0;
            *(uint32_t *)(d.mvd + ((size_t)s * d.nmb + mb) * 8 + lane * 2) = ((uint32_t)dvx & 0xffffu) | ((uint32_t)dvy << 16);
        }
        if (lane == 0) {
            mbt[mb] = type;
            atomicAdd(&d.stats[s * 5 + stat], 1);
        }
    }
    for (int i = lane; i < FER_LEVELS / 2; i += 64) ((uint32_t *)lvs)[i] = 0u;
    RES_SYNC();
    // ---- luma: source snapping (F/moestimation.cpp:561-584: a source sample within MAXDIFF of the prediction is
    // replaced by it), residual, transform, levels, reconstruction
    int MAXDIFF = d.maxdiff_set;
    unsigned long long nz_l;
    {
        int srcv[4];
#pragma unroll
        for (int k = 0; k < 4; k++) srcv[k] = (svL >> (8 * k)) & 0xff;
        if (d.maxdiff_set == -1) {
            int mean = wave_sum(srcv[0] + srcv[1] + srcv[2] + srcv[3]) / 256;
            int dev = wave_sum(iabs(srcv[0] - mean) + iabs(srcv[1] - mean) + iabs(srcv[2] - mean) + iabs(srcv[3] - mean));
            MAXDIFF = dev / 256;
            if (MAXDIFF < 3) MAXDIFF = 3;
        }
        int r[4], c[4], dc0;
#pragma unroll
        for (int k = 0; k < 4; k++) r[k] = iabs(srcv[k] - pfL[k]) < MAXDIFF ? 0 : srcv[k] - pfL[k];
        const RowQ q = rowq_make(row, d.lsq[0]);
        fwd_row(q, r, d.qp, false, c, dc0);
        const uint32_t z = ((const uint32_t *)c_izz)[row];
        int cnt = 0;
#pragma unroll
        for (int x = 0; x < 4; x++) {
            lvs[blk * 16 + ((z >> (8 * x)) & 15u)] = (int16_t)c[x];
            cnt += c[x] != 0;
        }
        cnt = quad_sum(cnt);
        if (row == 0) tcs[blk] = (uint8_t)cnt;
        nz_l = __ballot(cnt != 0);
        inv_row(q, c, d.qp, false, r);
        uint32_t o = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) o |= (uint32_t)clip255(pfL[k] + r[k]) << (8 * k);
        *(uint32_t *)dstL = o;
    }
    // ---- chroma on lanes 0..31 (32..63 shadow them and store nothing): plane = lane >> 4, block = (lane >> 2) & 3
    unsigned long long nz_c, dcm;
    {
        const bool act = lane < 32;
        uint8_t *dst = dstC;
        const int *pf = pfC;
        int r[4], c[4], dcraw;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int sk = (svC >> (8 * k)) & 0xff;
            r[k] = iabs(sk - pf[k]) <= MAXDIFF ? 0 : sk - pf[k];
        }
        const RowQ q = rowq_make(row, d.lsq[1]);
        fwd_row(q, r, d.qpc, true, c, dcraw);
        // chroma DC: 2x2 Hadamard + quantiser on the four DC values of each plane (row-0 lanes of its blocks)
        int dcq, dcdeq;
        {
            int f0[4], f1[4], cq0[4], cq1[4], dq0[4], dq1[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                f0[i] = __builtin_amdgcn_readlane(dcraw, i * 4);
                f1[i] = __builtin_amdgcn_readlane(dcraw, 16 + i * 4);
            }
            fwd_dc_chroma(f0, cq0, d.qpc);
            inv_dc_chroma(cq0, dq0, d.qpc);
            fwd_dc_chroma(f1, cq1, d.qpc);
            inv_dc_chroma(cq1, dq1, d.qpc);
            const int a = cb == 0 ? cq0[0] : (cb == 1 ? cq0[1] : (cb == 2 ? cq0[2] : cq0[3]));
            const int b = cb == 0 ? cq1[0] : (cb == 1 ? cq1[1] : (cb == 2 ? cq1[2] : cq1[3]));
            const int e = cb == 0 ? dq0[0] : (cb == 1 ? dq0[1] : (cb == 2 ? dq0[2] : dq0[3]));
            const int g = cb == 0 ? dq1[0] : (cb == 1 ? dq1[1] : (cb == 2 ? dq1[2] : dq1[3]));
            dcq = pl ? b : a;
            dcdeq = pl ? g : e;
        }
        const uint32_t z = ((const uint32_t *)c_izz)[row];
        int cnt = 0;
#pragma unroll
        for (int x = 0; x < 4; x++) {
            const int k = (int)((z >> (8 * x)) & 15u);
            if (k > 0) {  // (row 0, x 0) is the DC
                if (act) lvs[FER_LV_CAC + cblk * 15 + k - 1] = (int16_t)c[x];
                cnt += c[x] != 0;
            }
        }
        cnt = quad_sum(cnt);
        if (act && row == 0) {
            tcs[16 + cblk] = (uint8_t)cnt;
            lvs[FER_LV_CDC + cblk] = (int16_t)dcq;
        }
        nz_c = __ballot(act && cnt != 0);
        dcm = __ballot(act && dcq != 0);
        if (row == 0) c[0] = dcdeq;
        inv_row(q, c, d.qpc, true, r);
        uint32_t o = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) o |= (uint32_t)clip255(pf[k] + r[k]) << (8 * k);
        if (act) *(uint32_t *)dst = o;
    }
    RES_SYNC();
    {
        uint32_t *g = (uint32_t *)(d.levels + ((size_t)s * d.nmb + mb) * FER_LEVELS);
        for (int i = lane; i < FER_LEVELS / 2; i += 64) g[i] = ((const uint32_t *)lvs)[i];
        if (lane < 6) ((uint32_t *)(d.tc + ((size_t)s * d.nmb + mb) * 24))[lane] = ((const uint32_t *)tcs)[lane];
    }
    // coded block pattern (F/rbsp_encoding.cpp:21-105)
    if (lane == 0) {
        int l = 0;
        for (int i8 = 0; i8 < 4; i8++)
            if ((nz_l >> (i8 * 16)) & 0xffffu) l |= 1 << i8;
        int ch = 0;
        if (dcm) ch |= 1;
        if (nz_c) ch |= 2;
        if (ch == 3) ch = 2;
        d.cbp[((size_t)s * d.nmb + mb) * 2] = (uint8_t)l;
        d.cbp[((size_t)s * d.nmb + mb) * 2 + 1] = (uint8_t)ch;
    }
    chain_check();
}

void fer_launch_p_resid(const FerDev &d, hipStream_t st)
{
    hipLaunchKernelGGL(k_p_resid, dim3((d.nmb + RES_MBW - 1) / RES_MBW, d.S), dim3(64 * RES_MBW), 0, st, d);
}
