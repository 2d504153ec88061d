// fer_intra.hip -- I-macroblock decision and reconstruction (rows a10, a11, a12 of SURVEY.md 8a):
// intraPredictionEncoding (F/intra.cpp:949-1110, CPU path) followed by the final
// quantizationTransform / setCodedBlockPattern of RBSP_encode (F/rbsp_encoding.cpp:194-219).
//
// One wavefront owns one macroblock; macroblocks on the anti-diagonal x + 2y are independent
// (they need the reconstructed left, up, up-left and up-right neighbours), so the host
// launches this kernel once per diagonal over all streams.  Inside the wavefront:
//   phase 1  64 lanes = 4 Intra16x16 modes x 16 blocks: cost = sum |quantised coefficients|
//   phase 2  chroma prediction + residual (lanes 16..23), Intra16x16 trial levels (lanes 0..15),
//            exact bit count of the trial (27 CAVLC blocks sized one per lane)
//   phase 3  144 (block, mode) Intra4x4 costs in three rounds of 64 lanes
//   phase 4  the 16-step reconstruction chain of Intra4x4 (serial by construction)
//   phase 5  bit count of the Intra4x4 alternative, decision (strictly fewer bits wins)
//   phase 6  reconstruction + side information of the winner
#include "fer_cavlc_dev.h"
#include "fer_internal.h"
#include "fer_intra_dev.h"

// bits of one residual block for the size estimates; internal neighbours come from LDS
__device__ int nC_local(const FerDev &d, int s, int mb, bool luma, int blk, int plane, const uint8_t *tcl,
                        const uint8_t tcc[2][4], int cbpL, int cbpC, bool stale_skip)
{
    const int *mbt = d.mb_type + (size_t)s * d.nmb;
    const uint8_t *cbp = d.cbp + (size_t)s * d.nmb * 2;
    const uint8_t *tc = d.tc + (size_t)s * d.nmb * 24;
    bool edgeA, edgeB;
    int bA, bB;
    if (luma) {
        edgeA = blk == 0 || blk == 2 || blk == 8 || blk == 10;
        edgeB = blk == 0 || blk == 1 || blk == 4 || blk == 5;
        bA = c_nbA[blk];
        bB = c_nbB[blk];
    } else {
        edgeA = blk == 0 || blk == 2;
        edgeB = blk < 2;
        bA = c_nbcA[blk];
        bB = c_nbcB[blk];
    }
    bool availA = true, availB = true;
    int nA = 0, nB = 0;
    if (edgeA) {
        if (mb % d.mbw == 0)
            availA = false;
        else {
            int m = mb - 1;
            bool zero = luma ? ((cbp[m * 2] & (1 << (bA / 4))) == 0) : ((cbp[m * 2 + 1] & 2) == 0);
            if (!(mbt[m] == FER_P_SKIP || zero)) nA = luma ? tc[m * 24 + bA] : tc[m * 24 + 16 + plane * 4 + bA];
        }
    } else {
        bool zero = luma ? ((cbpL & (1 << (bA / 4))) == 0) : ((cbpC & 2) == 0);
        if (!(stale_skip || zero)) nA = luma ? tcl[bA] : tcc[plane][bA];
    }
    if (edgeB) {
        if (mb < d.mbw)
            availB = false;
        else {
            int m = mb - d.mbw;
            bool zero = luma ? ((cbp[m * 2] & (1 << (bB / 4))) == 0) : ((cbp[m * 2 + 1] & 2) == 0);
            if (!(mbt[m] == FER_P_SKIP || zero)) nB = luma ? tc[m * 24 + bB] : tc[m * 24 + 16 + plane * 4 + bB];
        }
    } else {
        bool zero = luma ? ((cbpL & (1 << (bB / 4))) == 0) : ((cbpC & 2) == 0);
        if (!(stale_skip || zero)) nB = luma ? tcl[bB] : tcc[plane][bB];
    }
    if (availA && availB) return (nA + nB + 1) >> 1;
    if (availA) return nA;
    if (availB) return nB;
    return 0;
}

__device__ __forceinline__ int ue_len(unsigned v) { return 2 * (31 - __clz((int)(v + 1))) + 1; }

__global__ __launch_bounds__(64) void k_intra_mb(FerDev d, int diag)
{
    __shared__ IntraLds L;
    const int lane = threadIdx.x;
    const int s = blockIdx.y;
    if (d.hdr[s * 4 + 3] != 2) return;
    int y_lo = diag - (d.mbw - 1);
    y_lo = y_lo > 0 ? (y_lo + 1) >> 1 : 0;
    const int mby = y_lo + blockIdx.x, mbx = diag - 2 * mby;
    if (mby >= d.mbh || mbx < 0 || mbx >= d.mbw) return;
    const int mb = mby * d.mbw + mbx;
    const int W = d.W, Wc = d.Wc;
    uint8_t *Y = d.curY + (size_t)s * d.ysz;
    uint8_t *Cp[2] = {d.curCb + (size_t)s * d.csz, d.curCr + (size_t)s * d.csz};
    const int xp = mbx << 4, yp = mby << 4;
    const bool availL = mbx > 0, availT = mby > 0, lastcol = mbx == d.mbw - 1;
    const int QPy = d.qp, QPc = d.qpc;
    const size_t mbi = (size_t)s * d.nmb + mb;
    const bool stale_skip = d.mb_type[mbi] == FER_P_SKIP;  // mb_type_array entry left by the previous picture

    // ---- load the window
    for (int i = lane; i < 17 * 21; i += 64) {
        int r = i / 21, c = i % 21;
        int gx = xp + c - 1, gy = yp + r - 1;
        int v = -1;
        if (gx >= 0 && gy >= 0 && gx < W && (r > 0 ? c <= 16 : true)) v = Y[(size_t)gy * W + gx];
        if (r > 0 && c > 16) v = -1;
        L.fr[r][c] = (int16_t)v;
    }
    for (int i = lane; i < 2 * 9 * 9; i += 64) {
        int pl = i / 81, r = (i % 81) / 9, c = i % 9;
        int gx = xp / 2 + c - 1, gy = yp / 2 + r - 1;
        L.cfr[pl][r][c] = (int16_t)((gx >= 0 && gy >= 0) ? Cp[pl][(size_t)gy * Wc + gx] : -1);
    }
    __syncthreads();

    // ---- Intra16x16 parameters (DC, plane)
    P16 q16;
    pred16_params(L, availL, availT, q16);

    // ---- phase 1: Intra16x16 mode costs (F/intra.cpp:980-1001)
    int mode16;
    {
        int mode = lane >> 4, blk = lane & 15;
        int x0 = c_bx[blk], y0 = c_by[blk];
        int r[16], t[16], qv[16];
        for (int i = 0; i < 16; i++)
            r[i] = L.fr[1 + y0 + (i >> 2)][1 + x0 + (i & 3)] - pred16_px(L, q16, mode, x0 + (i & 3), y0 + (i >> 2));
        fwd4x4(r, t);
        quant4x4(t, qv, QPy, false);
        int c = 0;
        for (int i = 0; i < 16; i++) c += iabs(qv[i]);
        c = row16_sum(c);
        int best = 0x7fffffff;
        mode16 = 0;
        for (int m = 0; m < 4; m++) {
            int cm = lane_bcast(c, m * 16);
            bool ok = !((m == 0 && !availT) || (m == 1 && !availL) || (m == 3 && !(availL && availT)));
            if (ok && cm < best) {
                best = cm;
                mode16 = m;
            }
        }
    }
    const int chroma_mode = mode16 == 0 ? 2 : (mode16 == 1 ? 1 : (mode16 == 2 ? 0 : 3));

    // ---- phase 2a: chroma prediction (F/intra.cpp:568-767), two samples per lane
    {
        int cx = lane & 7, cy = lane >> 3;
        for (int pl = 0; pl < 2; pl++)
            L.predC[pl][cy][cx] = (uint8_t)pred_chroma_px(L.cfr[pl], chroma_mode, cx, cy, availL, availT);
    }
    __syncthreads();

    // ---- phase 2b: chroma residual (lanes 16..23) and the Intra16x16 trial levels (lanes 0..15)
    const bool isL = lane < 16, isC = lane >= 16 && lane < 24;
    const int pl = (lane - 16) >> 2, cb = (lane - 16) & 3;
    int q16v[16];       // quantised Intra16x16 block of this lane (raster), DC slot filled later
    int crec[16];       // reconstructed chroma block of this lane
    int cbpC = 0;
    {
        int r[16], t[16], qv[16], p[16];
        int dcr = 0;
        if (isL) {
            int x0 = c_bx[lane], y0 = c_by[lane];
            for (int i = 0; i < 16; i++)
                r[i] = L.fr[1 + y0 + (i >> 2)][1 + x0 + (i & 3)] - pred16_px(L, q16, mode16, x0 + (i & 3), y0 + (i >> 2));
            fwd4x4(r, t);
            quant4x4(t, q16v, QPy, true);
            L.dcraw[(y0 >> 2) * 4 + (x0 >> 2)] = q16v[0];
            int cnt = 0;
            for (int k = 1; k < 16; k++) {
                int v = q16v[c_zz[k]];
                L.lv16[lane][k - 1] = (int16_t)v;
                cnt += v != 0;
            }
            L.lv16[lane][15] = 0;
            L.tc16[lane] = (uint8_t)cnt;
        } else if (isC) {
            int x0 = (cb & 1) * 4, y0 = (cb >> 1) * 4;
            for (int i = 0; i < 16; i++) {
                p[i] = L.predC[pl][y0 + (i >> 2)][x0 + (i & 3)];
                r[i] = (int)Cp[pl][(size_t)(yp / 2 + y0 + (i >> 2)) * Wc + xp / 2 + x0 + (i & 3)] - p[i];
            }
            fwd4x4(r, t);
            quant4x4(t, qv, QPc, true);
            dcr = qv[0];
        }
        int base = 16 + (lane >= 20 ? 4 : 0);
        int f[4], cq[4], dq[4];
        for (int i = 0; i < 4; i++) f[i] = __shfl(dcr, base + i);
        fwd_dc_chroma(f, cq, QPc);
        inv_dc_chroma(cq, dq, QPc);
        int cnt = 0;
        if (isC) {
            L.cdc[pl][cb] = (int16_t)cq[cb];
            for (int k = 1; k < 16; k++) {
                int v = qv[c_zz[k]];
                L.cac[pl][cb][k - 1] = (int16_t)v;
                cnt += v != 0;
            }
            L.cac[pl][cb][15] = 0;
            L.tcc[pl][cb] = (uint8_t)cnt;
            qv[0] = dq[cb];
            inv4x4(qv, r, QPc, true);
            for (int i = 0; i < 16; i++) crec[i] = clip255(p[i] + r[i]);
        }
        unsigned long long acm = __ballot(isC && cnt != 0);
        unsigned long long dcm = __ballot(isC && cq[cb] != 0);
        if (dcm) cbpC |= 1;
        if (acm) cbpC |= 2;
        if (cbpC == 3) cbpC = 2;
    }
    __syncthreads();
    if (lane == 0) {  // Intra16x16 DC: Hadamard + quantiser (F/quantizationTransform.cpp:105-152,227-260)
        int f[16], c[16], dq[16];
        for (int i = 0; i < 16; i++) f[i] = L.dcraw[i];
        fwd_dc_luma(f, c, QPy);
        for (int k = 0; k < 16; k++) L.dc16[k] = (int16_t)c[c_zz[k]];
        inv_dc_luma(c, dq, QPy);
        for (int i = 0; i < 16; i++) L.dcdeq[i] = dq[i];
    }
    __syncthreads();
    int cbpL16 = 0;
    for (int b = 0; b < 16; b++)
        if (L.tc16[b]) cbpL16 = 15;

    // ---- phase 2c: coded_mb_size(mode16) (F/rbsp_encoding.cpp:330-488)
    int bits16;
    {
        int t16 = 1 + mode16 + (cbpC << 2) + (cbpL16 == 15 ? 12 : 0);
        int hb = ue_len((unsigned)t16) + ue_len((unsigned)chroma_mode) + 1;
        BitW w;
        bw_init<false>(w, nullptr, 0, 0);
        // one residual block per lane (0: the DC block, 1..16: luma AC, 17..18: chroma DC, 19..26: chroma AC), sized by ONE
        // pass through the block coder with per-lane arguments -- the four kinds as four branches would run it four times
        {
            const int16_t *cp = L.dc16;
            int maxn = 0, blk = 0, plane = 0;
            bool luma = true;
            if (lane == 0) {
                maxn = 16;
            } else if (lane <= 16) {
                if (cbpL16) {
                    cp = L.lv16[lane - 1];
                    maxn = 15;
                    blk = lane - 1;
                }
            } else if (lane <= 18) {
                if (cbpC & 3) {
                    cp = L.cdc[lane - 17];
                    maxn = 4;
                }
            } else if (lane < 27) {
                if (cbpC & 2) {
                    plane = (lane - 19) >> 2;
                    blk = (lane - 19) & 3;
                    cp = L.cac[plane][blk];
                    maxn = 15;
                    luma = false;
                }
            }
            if (maxn) {
                const int nC = maxn == 4 ? -1 : nC_local(d, s, mb, luma, blk, plane, L.tc16, L.tcc, cbpL16, cbpC, stale_skip);
                cavlc_block<false>(w, cp, maxn, nC);
            }
        }
        bits16 = hb + wave_sum((int)w.bits);
    }

    // ---- phase 3: Intra4x4 mode costs on source samples (F/intra.cpp:1011-1048)
    // 144 (block, mode) tasks in three rounds; a round holds four modes (16 lanes = 16 blocks each), so the predictor's
    // switch runs 4 + 4 + 1 bodies instead of all nine in every round
    for (int rnd = 0; rnd < 3; rnd++) {
        const int blk = lane & 15, mode = rnd * 4 + (lane >> 4);
        const int task = blk * 9 + mode;
        if (mode < 9) {
            int p[13], o[16], r[16], t[16], qv[16];
            fetch4(L, blk, lastcol, p);
            int key = 0x7fffffff;
            if (mode4_avail(mode, p)) {
                pred4x4(mode, p, o);
                int x0 = c_bx[blk], y0 = c_by[blk];
                for (int i = 0; i < 16; i++) r[i] = L.fr[1 + y0 + (i >> 2)][1 + x0 + (i & 3)] - o[i];
                fwd4x4(r, t);
                quant4x4(t, qv, QPy, false);
                int c = 0;
                for (int i = 0; i < 16; i++) c += iabs(qv[i]);
                key = (c << 4) | mode;
            }
            L.key4[task] = key;
        }
    }
    __syncthreads();
    if (lane < 16) {
        int best = 0x7fffffff;
        for (int m = 0; m < 9; m++) best = min(best, L.key4[lane * 9 + m]);
        L.mode4[lane] = (uint8_t)(best & 15);
    }
    __syncthreads();

    // ---- phase 4: Intra4x4 reconstruction chain (F/intra.cpp:1062-1086), serial over the 16 blocks
    const uint8_t *nmode = d.i4mode + (size_t)s * d.nmb * 16;
    const int *mbt = d.mb_type + (size_t)s * d.nmb;
    // Four lanes per block: lane r < 4 owns row r of the block being reconstructed (fwd_row / inv_row, the vertical
    // transform halves cross the quad with DPP); the other lanes shadow them and write nothing.  The chain itself is
    // serial by construction: a block predicts from the reconstruction of the blocks before it.
    {
        const int row = lane & 3;
        const RowQ rq = rowq_make(row, d.lsq[0]);
        const uint32_t zrow = ((const uint32_t *)c_izz)[row];
        for (int blk = 0; blk < 16; blk++) {
            // setIntra4x4PredMode (F/intra.cpp:878-942)
            bool edgeA = blk == 0 || blk == 2 || blk == 8 || blk == 10;
            bool edgeB = blk == 0 || blk == 1 || blk == 4 || blk == 5;
            bool okA = !(edgeA && !availL), okB = !(edgeB && !availT);
            int mA = 2, mB = 2;
            if (okA && okB) {
                if (edgeA) {
                    int m = mb - 1;
                    mA = mbt[m] == 0 ? nmode[m * 16 + c_nbA[blk]] : 2;
                } else
                    mA = L.mode4[c_nbA[blk]];
                if (edgeB) {
                    int m = mb - d.mbw;
                    mB = mbt[m] == 0 ? nmode[m * 16 + c_nbB[blk]] : 2;
                } else
                    mB = L.mode4[c_nbB[blk]];
            }
            const int pm = mA <= mB ? mA : mB, mode = L.mode4[blk];
            int p[13], o[16], r[4], c[4], dc0;
            fetch4(L, blk, lastcol, p);
            pred4x4(mode, p, o);
            const int x0 = c_bx[blk], y0 = c_by[blk];
            int po[4];  // the lane's row of the prediction
#pragma unroll
            for (int k = 0; k < 4; k++) {
                po[k] = row == 0 ? o[k] : (row == 1 ? o[4 + k] : (row == 2 ? o[8 + k] : o[12 + k]));
                r[k] = L.fr[1 + y0 + row][1 + x0 + k] - po[k];
            }
            fwd_row(rq, r, QPy, false, c, dc0);
            int cnt = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (lane < 4) L.lv4[blk][(zrow >> (8 * k)) & 15u] = (int16_t)c[k];
                cnt += c[k] != 0;
            }
            cnt = quad_sum(cnt);
            inv_row(rq, c, QPy, false, r);
            if (lane < 4) {
#pragma unroll
                for (int k = 0; k < 4; k++) L.fr[1 + y0 + row][1 + x0 + k] = (int16_t)clip255(po[k] + r[k]);
            }
            if (lane == 0) {
                L.flag4[blk] = (uint8_t)(mode == pm ? 8 : (mode < pm ? mode : mode - 1));
                L.tc4[blk] = (uint8_t)cnt;
            }
            __syncthreads();  // (one wavefront: orders the LDS traffic of this block before the next one's)
        }
    }
    __syncthreads();

    // ---- phase 5: coded_mb_size(-1)
    int cbpL4 = 0;
    for (int i8 = 0; i8 < 4; i8++)
        if (L.tc4[i8 * 4] | L.tc4[i8 * 4 + 1] | L.tc4[i8 * 4 + 2] | L.tc4[i8 * 4 + 3]) cbpL4 |= 1 << i8;
    int bits4;
    {
        int hb = 1;  // ue(I_4x4 = 0)
        for (int b = 0; b < 16; b++) hb += (L.flag4[b] & 8) ? 1 : 4;
        hb += ue_len((unsigned)chroma_mode) + ue_len(c_cbp_intra_code[(cbpC << 4) | cbpL4]);
        BitW w;
        bw_init<false>(w, nullptr, 0, 0);
        if (cbpL4 > 0 || cbpC > 0) {
            hb += 1;
            const int16_t *cp = L.lv4[0];
            int maxn = 0, blk = 0, plane = 0;
            bool luma = true;
            if (lane < 16) {
                if (cbpL4 & (1 << (lane >> 2))) {
                    cp = L.lv4[lane];
                    maxn = 16;
                    blk = lane;
                }
            } else if (lane == 17 || lane == 18) {
                if (cbpC & 3) {
                    cp = L.cdc[lane - 17];
                    maxn = 4;
                }
            } else if (lane >= 19 && lane < 27) {
                if (cbpC & 2) {
                    plane = (lane - 19) >> 2;
                    blk = (lane - 19) & 3;
                    cp = L.cac[plane][blk];
                    maxn = 15;
                    luma = false;
                }
            }
            if (maxn) {
                const int nC = maxn == 4 ? -1 : nC_local(d, s, mb, luma, blk, plane, L.tc4, L.tcc, cbpL4, cbpC, false);
                cavlc_block<false>(w, cp, maxn, nC);
            }
        }
        bits4 = hb + wave_sum((int)w.bits);
    }
    const bool use4 = (unsigned)bits4 < (unsigned)bits16;
    if (lane == 0) {  // coded_mb_size of either alternative (F/rbsp_encoding.cpp:330): FERHIP_BUF_MBSIZE
        d.mbsize[mbi * 2] = bits16;
        d.mbsize[mbi * 2 + 1] = bits4;
    }

    // ---- phase 6: reconstruction and side information of the winner
    int16_t *lv = d.levels + mbi * FER_LEVELS;
    if (isL) {
        int x0 = c_bx[lane], y0 = c_by[lane];
        uint8_t *dst = Y + (size_t)(yp + y0) * W + xp + x0;
        if (use4) {
            for (int ry = 0; ry < 4; ry++) {  // one dword per block row
                uint32_t o = 0;
                for (int k = 0; k < 4; k++) o |= (uint32_t)(uint8_t)L.fr[1 + y0 + ry][1 + x0 + k] << (8 * k);
                *(uint32_t *)(dst + (size_t)ry * W) = o;
            }
            for (int k = 0; k < 16; k++) lv[lane * 16 + k] = L.lv4[lane][k];
            d.tc[mbi * 24 + lane] = L.tc4[lane];
        } else {
            int r[16];
            q16v[0] = L.dcdeq[(y0 >> 2) * 4 + (x0 >> 2)];
            inv4x4(q16v, r, QPy, true);
            for (int ry = 0; ry < 4; ry++) {
                uint32_t o = 0;
                for (int k = 0; k < 4; k++)
                    o |= (uint32_t)clip255(pred16_px(L, q16, mode16, x0 + k, y0 + ry) + r[ry * 4 + k]) << (8 * k);
                *(uint32_t *)(dst + (size_t)ry * W) = o;
            }
            for (int k = 0; k < 16; k++) lv[lane * 16 + k] = L.lv16[lane][k];
            lv[FER_LV_DC16 + lane] = L.dc16[lane];
            d.tc[mbi * 24 + lane] = L.tc16[lane];
        }
        d.i4mode[mbi * 16 + lane] = L.mode4[lane];
        d.i4flag[mbi * 16 + lane] = L.flag4[lane];
    } else if (isC) {
        int x0 = (cb & 1) * 4, y0 = (cb >> 1) * 4;
        uint8_t *dst = Cp[pl] + (size_t)(yp / 2 + y0) * Wc + xp / 2 + x0;
        for (int ry = 0; ry < 4; ry++)
            *(uint32_t *)(dst + (size_t)ry * Wc) = (uint32_t)crec[ry * 4] | ((uint32_t)crec[ry * 4 + 1] << 8) |
                                                  ((uint32_t)crec[ry * 4 + 2] << 16) | ((uint32_t)crec[ry * 4 + 3] << 24);
        lv[FER_LV_CDC + pl * 4 + cb] = L.cdc[pl][cb];
        for (int k = 0; k < 15; k++) lv[FER_LV_CAC + (pl * 4 + cb) * 15 + k] = L.cac[pl][cb][k];
        d.tc[mbi * 24 + lane] = L.tcc[pl][cb];
    }
    if (lane == 0) {
        int cl = use4 ? cbpL4 : cbpL16;
        d.mb_type[mbi] = use4 ? 0 : 1 + mode16 + (cbpC << 2) + (cl == 15 ? 12 : 0);
        d.cbp[mbi * 2] = (uint8_t)cl;
        d.cbp[mbi * 2 + 1] = (uint8_t)cbpC;
        d.chroma_mode[mbi] = (uint8_t)chroma_mode;
    }
}

void fer_launch_intra(const FerDev &d, hipStream_t st)
{
    int ndiag = d.mbw + 2 * (d.mbh - 1);
    int maxk = min(d.mbh, (d.mbw + 1) / 2);
    for (int dg = 0; dg < ndiag; dg++) hipLaunchKernelGGL(k_intra_mb, dim3(maxk, d.S), dim3(64), 0, st, d, dg);
}
