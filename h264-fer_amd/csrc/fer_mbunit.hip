// fer_mbunit.hip -- the per-macroblock unit-parity surface of SURVEY.md 8b: the reference's macroblock-level functions
// as batched device calls over the SAME device functions the encode / decode kernels are built from (fer_dev.h,
// fer_cavlc_dev.h), so that a known-answer test pins them one function at a time:
//   quantizationTransform                       F/quantizationTransform.cpp:349-486   k_mb_unit, MBU_QT
//   transformDecoding4x4LumaResidual            F/inttransform.cpp:133-155            MBU_DEC4
//   transformDecodingIntra_16x16Luma            F/inttransform.cpp:157-213            MBU_DEC16
//   transformDecodingChroma                     F/inttransform.cpp:237-320            MBU_DECC
//   transformDecodingP_Skip                     F/inttransform.cpp:215-231            MBU_SKIP
//   residual_block_cavlc_write / _size          F/residual.cpp:374-666 / :673-957     k_cavlc_blocks
//   MotionCompensateSubMBPart                   F/mocomp.cpp:152-195                  k_mc_parts
// Not a fast path: one wavefront per job, records of plain int32.  The legacy-name shims are in fer_legacy.hip.
#include "../../include/ferhip.h"
#include "fer_cavlc_dev.h"
#include "fer_internal.h"
#include <stdio.h>
#include <vector>

#define CKU(x)                                                                                         \
    do {                                                                                               \
        hipError_t e_ = (x);                                                                           \
        if (e_ != hipSuccess) {                                                                        \
            fprintf(stderr, "ferhip: %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            rc = FERHIP_E_HIP;                                                                         \
            goto done;                                                                                 \
        }                                                                                              \
    } while (0)

// list[k] = c[zig-zag k] (transformScan, F/quantizationTransform.cpp:310-325); the Intra16x16AC / chroma AC form drops the DC
__device__ __forceinline__ void mbu_scan(const int c[16], int32_t *list, bool ac)
{
#pragma unroll
    for (int k = ac ? 1 : 0; k < 16; k++) list[ac ? k - 1 : k] = c[c_zz[k]];
    if (ac) list[15] = 0;
}
__device__ __forceinline__ void mbu_invscan(const int32_t *list, int c[16], bool ac, int dc)
{
#pragma unroll
    for (int k = 0; k < 16; k++) c[c_zz[k]] = ac ? (k == 0 ? dc : list[k - 1]) : list[k];
}

// One job = one macroblock.  lanes 0..15 own the luma 4x4 blocks (luma4x4BlkIdx order), lanes 16..23 the chroma blocks
// (Cb 0..3, Cr 0..3); the DC transforms run on the lane of block 0 of their plane.
__global__ __launch_bounds__(64) void k_mb_unit(const ferhip_mb_job *jobs, ferhip_mb_result *res, int n)
{
    __shared__ int dcl[16], dcc[2][4], dcdeq[2][4], dcy[16];
    const int j = blockIdx.x, lane = threadIdx.x;
    if (j >= n) return;
    const ferhip_mb_job &J = jobs[j];
    ferhip_mb_result &R = res[j];
    const int op = J.op, qp = J.qp, qpc = J.qpc;
    const bool luma = lane < 16, chroma = lane >= 16 && lane < 24;
    const int blk = lane & 15, cpl = (lane - 16) >> 2, cblk = (lane - 16) & 3;
    const int x0 = luma ? c_bx[blk] : (cblk & 1) * 4, y0 = luma ? c_by[blk] : (cblk >> 1) * 4;
    int src[16], pred[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int yy = y0 + (i >> 2), xx = x0 + (i & 3);
        if (luma) {
            src[i] = J.srcY[yy * 16 + xx];
            pred[i] = J.predY[yy * 16 + xx];
        } else if (chroma) {
            src[i] = (cpl ? J.srcCr : J.srcCb)[yy * 8 + xx];
            pred[i] = (cpl ? J.predCr : J.predCb)[yy * 8 + xx];
        } else {
            src[i] = pred[i] = 0;
        }
    }
    auto store_rec = [&](const int r[16]) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int yy = y0 + (i >> 2), xx = x0 + (i & 3);
            const int v = clip255(pred[i] + r[i]);
            if (luma)
                R.recY[yy * 16 + xx] = v;
            else if (chroma)
                (cpl ? R.recCr : R.recCb)[yy * 8 + xx] = v;
        }
    };
    if (op == FERHIP_MBU_QT) {
        // ---- quantizationTransform(predL, predCb, predCr, reconstruct); cls = MbPartPredMode(mb_type, 0): 0 Intra_4x4
        // (luma is left to the prediction loop), 1 Intra_16x16, 2 inter
        const int cls = J.cls;
        int diff[16], d[16], c[16], r[16];
#pragma unroll
        for (int i = 0; i < 16; i++) diff[i] = src[i] - pred[i];
        fwd4x4(diff, d);
        if (luma && cls != 0) {
            quant4x4(d, c, qp, cls == 1);
            if (cls == 1) {
                dcl[(y0 >> 2) * 4 + (x0 >> 2)] = c[0];
                mbu_scan(c, R.ac16[blk], true);
            } else {
                mbu_scan(c, R.lumaLevel[blk], false);
                if (J.reconstruct) {
                    int cc[16];
                    mbu_invscan(R.lumaLevel[blk], cc, false, 0);
                    inv4x4(cc, r, qp, false);
                    store_rec(r);
                }
            }
        }
        if (chroma) {
            quant4x4(d, c, qpc, true);
            dcc[cpl][cblk] = c[0];
            mbu_scan(c, R.cac[cpl][cblk], true);
        }
        __syncthreads();
        if (lane == 0 && cls == 1) {
            int t[16], q[16];
#pragma unroll
            for (int i = 0; i < 16; i++) t[i] = dcl[i];
            fwd_dc_luma(t, q, qp);
            mbu_scan(q, R.dc16, false);
            if (J.reconstruct) {
                int cc[16], dq[16];
                mbu_invscan(R.dc16, cc, false, 0);
                inv_dc_luma(cc, dq, qp);
#pragma unroll
                for (int i = 0; i < 16; i++) dcy[i] = dq[i];
            }
        }
        if ((lane == 16 || lane == 20)) {
            int f[4], q[4], dq[4];
#pragma unroll
            for (int i = 0; i < 4; i++) f[i] = dcc[cpl][i];
            fwd_dc_chroma(f, q, qpc);
#pragma unroll
            for (int i = 0; i < 4; i++) R.cdc[cpl][i] = q[i];
            inv_dc_chroma(q, dq, qpc);
#pragma unroll
            for (int i = 0; i < 4; i++) dcdeq[cpl][i] = dq[i];
        }
        __syncthreads();
        if (J.reconstruct) {
            if (luma && cls == 1) {
                int cc[16];
                mbu_invscan(R.ac16[blk], cc, true, dcy[(y0 >> 2) * 4 + (x0 >> 2)]);
                inv4x4(cc, r, qp, true);
                store_rec(r);
            }
            if (chroma) {
                int cc[16];
                mbu_invscan(R.cac[cpl][cblk], cc, true, dcdeq[cpl][cblk]);
                inv4x4(cc, r, qpc, true);
                store_rec(r);
            }
        }
        return;
    }
    // ---- the decode-side drivers: levels in, samples out
    int r[16], cc[16];
    if (op == FERHIP_MBU_DEC4) {  // transformDecoding4x4LumaResidual(LumaLevel, predL, luma4x4BlkIdx = J.blk, QPy)
        if (luma && blk == J.blk) {
            mbu_invscan(J.lumaLevel[blk], cc, false, 0);
            inv4x4(cc, r, qp, false);
            store_rec(r);
        }
    } else if (op == FERHIP_MBU_DEC16) {  // transformDecodingIntra_16x16Luma(Intra16x16DCLevel, Intra16x16ACLevel, predL, QPy)
        if (lane == 0) {
            int dq[16];
            mbu_invscan(J.dc16, cc, false, 0);
            inv_dc_luma(cc, dq, qp);
#pragma unroll
            for (int i = 0; i < 16; i++) dcy[i] = dq[i];
        }
        __syncthreads();
        if (luma) {
            mbu_invscan(J.ac16[blk], cc, true, dcy[(y0 >> 2) * 4 + (x0 >> 2)]);
            inv4x4(cc, r, qp, true);
            store_rec(r);
        }
    } else if (op == FERHIP_MBU_DECC) {  // transformDecodingChroma(ChromaDCLevel, ChromaACLevel, predC, QPy, Cb) for both planes
        if (lane == 16 || lane == 20) {
            int q[4], dq[4];
#pragma unroll
            for (int i = 0; i < 4; i++) q[i] = J.cdc[cpl][i];
            inv_dc_chroma(q, dq, qpc);
#pragma unroll
            for (int i = 0; i < 4; i++) dcdeq[cpl][i] = dq[i];
        }
        __syncthreads();
        if (chroma) {
            mbu_invscan(J.cac[cpl][cblk], cc, true, dcdeq[cpl][cblk]);
            inv4x4(cc, r, qpc, true);
            store_rec(r);
        }
    } else if (op == FERHIP_MBU_SKIP) {  // transformDecodingP_Skip: all levels zero
#pragma unroll
        for (int i = 0; i < 16; i++) cc[i] = 0;
        inv4x4(cc, r, luma ? qp : qpc, !luma);
        if (luma || chroma) store_rec(r);
    }
}

extern "C" int ferhip_mb_unit(const ferhip_mb_job *jobs, ferhip_mb_result *results, size_t n)
{
    if (!jobs || !results || n == 0) return FERHIP_E_ARG;
    int rc = 0;
    ferhip_mb_job *dj = nullptr;
    ferhip_mb_result *dr = nullptr;
    CKU(hipMalloc((void **)&dj, n * sizeof *dj));
    CKU(hipMalloc((void **)&dr, n * sizeof *dr));
    CKU(hipMemcpy(dj, jobs, n * sizeof *dj, hipMemcpyHostToDevice));
    CKU(hipMemset(dr, 0, n * sizeof *dr));
    hipLaunchKernelGGL(k_mb_unit, dim3((unsigned)n), dim3(64), 0, 0, dj, dr, (int)n);
    CKU(hipGetLastError());
    CKU(hipMemcpy(results, dr, n * sizeof *dr, hipMemcpyDeviceToHost));
done:
    if (dj) hipFree(dj);
    if (dr) hipFree(dr);
    return rc;
}

// ---- residual_block_cavlc_write / _size for n blocks: the device block coder of the entropy pass and of coded_mb_size
__global__ __launch_bounds__(64) void k_cavlc_blocks(const int32_t *coef, const int32_t *nC, const int32_t *maxn, int n, uint32_t *bits,
                                                     uint32_t *nbits, int32_t *tc, uint32_t *nbits_size)
{
    __shared__ int16_t stage[16 * 64];
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    int16_t *st = stage + threadIdx.x;
    const int m = maxn[i];
    unsigned nz = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int v = k < m ? coef[(size_t)i * 16 + k] : 0;
        st[k * 64] = (int16_t)v;
        nz |= (v != 0 ? 1u : 0u) << k;
    }
    BitW w;
    bw_init<true>(w, bits + (size_t)i * 16, 16, 0);
    cavlc_block_core<true>(w, nz, m, nC[i], st, 64);
    bw_flush<true>(w);
    nbits[i] = w.bits;
    tc[i] = __popc(nz);
    BitW ws;  // the counting form (residual_block_cavlc_size) must agree with the writer
    bw_init<false>(ws, nullptr, 0, 0);
    cavlc_block_core<false>(ws, nz, m, nC[i], st, 64);
    nbits_size[i] = ws.bits;
}

extern "C" int ferhip_cavlc_blocks(const int32_t *coef, const int32_t *nC, const int32_t *max_num_coeff, size_t n, uint8_t *bits,
                                   uint32_t *nbits, int32_t *total_coeff)
{
    if (!coef || !nC || !max_num_coeff || !bits || !nbits || !total_coeff || n == 0) return FERHIP_E_ARG;
    int rc = 0;
    int32_t *dc = nullptr, *dn = nullptr, *dm = nullptr, *dt = nullptr;
    uint32_t *db = nullptr, *dl = nullptr, *ds = nullptr;
    std::vector<uint32_t> sz(n);
    CKU(hipMalloc((void **)&dc, n * 64));
    CKU(hipMalloc((void **)&dn, n * 4));
    CKU(hipMalloc((void **)&dm, n * 4));
    CKU(hipMalloc((void **)&dt, n * 4));
    CKU(hipMalloc((void **)&db, n * 64));
    CKU(hipMalloc((void **)&dl, n * 4));
    CKU(hipMalloc((void **)&ds, n * 4));
    CKU(hipMemcpy(dc, coef, n * 64, hipMemcpyHostToDevice));
    CKU(hipMemcpy(dn, nC, n * 4, hipMemcpyHostToDevice));
    CKU(hipMemcpy(dm, max_num_coeff, n * 4, hipMemcpyHostToDevice));
    CKU(hipMemset(db, 0, n * 64));
    hipLaunchKernelGGL(k_cavlc_blocks, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, dc, dn, dm, (int)n, db, dl, dt, ds);
    CKU(hipGetLastError());
    CKU(hipMemcpy(bits, db, n * 64, hipMemcpyDeviceToHost));
    CKU(hipMemcpy(nbits, dl, n * 4, hipMemcpyDeviceToHost));
    CKU(hipMemcpy(total_coeff, dt, n * 4, hipMemcpyDeviceToHost));
    CKU(hipMemcpy(sz.data(), ds, n * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++)
        if (sz[i] != nbits[i]) {
            fprintf(stderr, "ferhip_cavlc_blocks: block %zu: the counting form gives %u bits, the writer %u\n", i, sz[i], nbits[i]);
            rc = FERHIP_E_DEVICE;
            break;
        }
done:
    hipFree(dc);
    hipFree(dn);
    hipFree(dm);
    hipFree(dt);
    hipFree(db);
    hipFree(dl);
    hipFree(ds);
    return rc;
}

// ---- MotionCompensateSubMBPart for n (macroblock, 4x4 sub-block, vector) triples on one reference picture:
// thread = one luma sample of the 4x4 block (lanes 0..15), one chroma sample of either plane's 2x2 block (16..23),
// and the four-sample row form the residual kernel uses for the same chroma samples (24..27: Cb rows, 28..31: Cr rows)
__global__ __launch_bounds__(64) void k_mc_parts(const uint8_t *ref, int W, int H, const int32_t *desc, int n, int32_t *predL, int32_t *predCb,
                                                 int32_t *predCr, int32_t *rowCb, int32_t *rowCr)
{
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= n) return;
    const int mbw = W >> 4, Wc = W >> 1, Hc = H >> 1;
    const uint8_t *RY = ref, *RCb = ref + (size_t)W * H, *RCr = RCb + (size_t)Wc * Hc;
    const int mb = desc[i * 5], sub = desc[i * 5 + 1], part = desc[i * 5 + 2], mvx = desc[i * 5 + 3], mvy = desc[i * 5 + 4];
    const int org_y = ((sub & 2) << 2) + ((part & 2) << 1), org_x = ((sub & 1) << 3) + ((part & 1) << 2);
    const int xP = (mb % mbw) << 4, yP = (mb / mbw) << 4;
    if (lane < 16) {
        predL[i * 16 + lane] = mc_luma(RY, W, H, xP, yP, org_x + (lane & 3), org_y + (lane >> 2), mvx, mvy);
    } else if (lane < 24) {
        const int k = lane & 3, pl = (lane >> 2) & 1;
        const int v = mc_chroma(pl ? RCr : RCb, Wc, Hc, xP >> 1, yP >> 1, (org_x >> 1) + (k & 1), (org_y >> 1) + (k >> 1), mvx, mvy);
        (pl ? predCr : predCb)[i * 4 + k] = v;
    } else if (lane < 32) {
        // mc_chroma_row4 computes the four samples x0 .. x0 + 3 of a chroma row with the vector of the first one; the
        // 2x2 block of this sub-block lies in the row's first or second half
        const int pl = (lane >> 2) & 1, ry = (lane & 3) >> 1, half = lane & 1;
        const int cy = (org_y >> 1) + ry, cx0 = ((org_x >> 1) & ~3);
        int o[4];
        mc_chroma_row4(pl ? RCr : RCb, Wc, Hc, xP >> 1, yP >> 1, cx0, cy, mvx, mvy, o);
        const int within = (org_x >> 1) & 3;  // 0 or 2: where the 2x2 block sits in the row of four
        if (half == 0) {
            (pl ? rowCr : rowCb)[i * 4 + ry * 2 + 0] = o[within];
            (pl ? rowCr : rowCb)[i * 4 + ry * 2 + 1] = o[within + 1];
        }
    }
}

extern "C" int ferhip_mc_sub_mb_parts(const uint8_t *ref_i420, int width, int height, const int32_t *desc, size_t n, int32_t *predL,
                                      int32_t *predCb, int32_t *predCr)
{
    if (!ref_i420 || !desc || !predL || !predCb || !predCr || n == 0 || width <= 0 || height <= 0 || (width & 15) || (height & 15))
        return FERHIP_E_ARG;
    int rc = 0;
    const size_t fsz = (size_t)width * height * 3 / 2;
    uint8_t *dref = nullptr;
    int32_t *dd = nullptr, *dl = nullptr, *db = nullptr, *dr = nullptr, *rb = nullptr, *rr = nullptr;
    std::vector<int32_t> hb(n * 4), hr(n * 4);
    CKU(hipMalloc((void **)&dref, fsz + 256));
    CKU(hipMalloc((void **)&dd, n * 20));
    CKU(hipMalloc((void **)&dl, n * 64));
    CKU(hipMalloc((void **)&db, n * 16));
    CKU(hipMalloc((void **)&dr, n * 16));
    CKU(hipMalloc((void **)&rb, n * 16));
    CKU(hipMalloc((void **)&rr, n * 16));
    CKU(hipMemcpy(dref, ref_i420, fsz, hipMemcpyHostToDevice));
    CKU(hipMemcpy(dd, desc, n * 20, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_mc_parts, dim3((unsigned)n), dim3(64), 0, 0, dref, width, height, dd, (int)n, dl, db, dr, rb, rr);
    CKU(hipGetLastError());
    CKU(hipMemcpy(predL, dl, n * 64, hipMemcpyDeviceToHost));
    CKU(hipMemcpy(predCb, db, n * 16, hipMemcpyDeviceToHost));
    CKU(hipMemcpy(predCr, dr, n * 16, hipMemcpyDeviceToHost));
    CKU(hipMemcpy(hb.data(), rb, n * 16, hipMemcpyDeviceToHost));
    CKU(hipMemcpy(hr.data(), rr, n * 16, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n * 4; i++)
        if (hb[i] != predCb[i] || hr[i] != predCr[i]) {
            fprintf(stderr, "ferhip_mc_sub_mb_parts: the row form of the chroma interpolation differs at sub-block %zu\n", i / 4);
            rc = FERHIP_E_DEVICE;
            break;
        }
done:
    hipFree(dref);
    hipFree(dd);
    hipFree(dl);
    hipFree(db);
    hipFree(dr);
    hipFree(rb);
    hipFree(rr);
    return rc;
}
