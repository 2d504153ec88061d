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
#include <stdlib.h>
#include "fer_internal.h"
#include "fer_mvpred.h"

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

// ---- top-K of a candidate set by (metric, arrival order), without serial insertion ----
// The reference's list (F/moestimation.cpp:277-291) ends up holding the K smallest candidates
// ordered by metric, ties by arrival.  With candidate (u, lane) arriving at index u*64 + lane
// that is a selection problem: binary-search the K-th smallest metric with ballot counts, take
// everything below it plus the earliest arrivals equal to it, then rank the <= K winners.
// v[u] < 0 marks an invalid candidate.  sel = 4*64 ints of LDS scratch owned by the wavefront.
template <int NB>
__device__ void select_topk(const int (&v)[NB], const int (&xy)[NB], int K, int lane, int *sel, WList &L)
{
    int *key = sel, *kxy = sel + 64, *key2 = sel + 128, *kxy2 = sel + 192;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int total = 0, lmin = 0x7fffffff, lmax = -1;
#pragma unroll
    for (int u = 0; u < NB; u++) {
        bool ok = v[u] >= 0;
        total += __popcll(__ballot(ok));
        if (ok) lmin = min(lmin, v[u]);
        lmax = max(lmax, v[u]);
    }
    const int need = min(K, total);
    L.m = INF_M;
    L.xy = 0;
    if (need == 0) return;
    int lo = wave_min(lmin), hi;
    if (__popcll(__ballot(lmin != 0x7fffffff)) >= need)
        hi = wave_max(lmin != 0x7fffffff ? lmin : -1);  // >= need candidates are <= the largest lane minimum
    else
        hi = wave_max(lmax);
    while (lo < hi) {  // smallest T with count(v <= T) >= need
        int mid = lo + ((hi - lo) >> 1);
        int cnt = 0;
#pragma unroll
        for (int u = 0; u < NB; u++) cnt += __popcll(__ballot(v[u] >= 0 && v[u] <= mid));
        if (cnt >= need)
            hi = mid;
        else
            lo = mid + 1;
    }
    const int T = lo;
    int c_less = 0;
#pragma unroll
    for (int u = 0; u < NB; u++) c_less += __popcll(__ballot(v[u] >= 0 && v[u] < T));
    const int take_eq = need - c_less;
    int nsel = 0, eq_seen = 0;
#pragma unroll
    for (int u = 0; u < NB; u++) {
        bool ok = v[u] >= 0;
        bool eq = ok && v[u] == T;
        unsigned long long meq = __ballot(eq);
        bool take = (ok && v[u] < T) || (eq && eq_seen + __popcll(meq & lt) < take_eq);
        unsigned long long mt = __ballot(take);
        if (take) {
            int pos = nsel + __popcll(mt & lt);
            key[pos] = v[u];
            kxy[pos] = xy[u];
        }
        nsel += __popcll(mt);
        eq_seen += __popcll(meq);
    }
    __syncthreads();
    int mk = lane < need ? key[lane] : INF_M, mxy = lane < need ? kxy[lane] : 0;
    int rank = 0;
    for (int i = 0; i < need; i++) {
        int ki = key[i];
        rank += (ki < mk) || (ki == mk && i < lane);
    }
    if (lane < need) {
        key2[rank] = mk;
        kxy2[rank] = mxy;
    }
    __syncthreads();
    if (lane < need) {
        L.m = key2[lane];
        L.xy = kxy2[lane];
    }
    __syncthreads();
}

// the 9-term feature distance of F/moestimation.cpp:267-276 from one 12-byte feature record
__device__ __forceinline__ int feat_dist_rec(const uint16_t *__restrict__ rec, const int s[5])
{
    const uint32_t *r = (const uint32_t *)rec;
    uint32_t a = r[0], b = r[1], c = r[2];
    int k0 = (int)(a & 0xffff), k1 = (int)(a >> 16), k2 = (int)(b & 0xffff), k3 = (int)(b >> 16), k4 = (int)(c & 0xffff);
    return iabs(s[0] - k0) + iabs(s[1] - k1) + iabs(s[0] - s[1] - k0 + k1) + iabs(s[2] - k2) +
           iabs(s[0] - s[2] - k0 + k2) + iabs(s[3] - k3) + iabs(s[0] - s[3] - k0 + k3) + iabs(s[4] - k4) +
           iabs(s[0] - s[4] - k0 + k4);
}
// ... at (frac, refy, refx) of the all-fracs array
__device__ __forceinline__ int feat_dist(const uint16_t *__restrict__ Fs, size_t ysz, int W, int frac, int refy,
                                         int refx, const int s[5])
{
    (void)ysz;
    return feat_dist_rec(Fs + (((size_t)refy * W + refx) * 16 + frac) * 6, s);
}

// SAD of the 8x8 source block against interpolated plane (F/moestimation.cpp:175-195).
// 8 lanes (rows) per candidate, 8 candidates per call: lane = cand*8 + row; src = the lane's
// source row packed in two dwords.  Returns the full SAD in every lane of the group.
__device__ __forceinline__ int sad8_rows(const uint8_t *__restrict__ Ps, size_t ysz, int W, int H, int xP, int yP,
                                         int mvx, int mvy, int row, uint32_t s0, uint32_t s1)
{
    int xPi = iclamp(xP + (mvx >> 2), 0, W - 1), yPi = iclamp(yP + (mvy >> 2), 0, H - 1);
    const uint8_t *R = Ps + (size_t)((mvx & 3) + (mvy & 3) * 4) * ysz + (size_t)min(yPi + row, H - 1) * W;
    uint32_t r0, r1;
    if (xPi + 7 < W) {
        load_u8x8(R + xPi, r0, r1);
    } else {  // right edge: the reference clamps each column (F/moestimation.cpp:189)
        r0 = r1 = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            r0 |= (uint32_t)R[min(xPi + j, W - 1)] << (8 * j);
            r1 |= (uint32_t)R[min(xPi + 4 + j, W - 1)] << (8 * j);
        }
    }
    int s = (int)__builtin_amdgcn_sad_u8(r1, s1, __builtin_amdgcn_sad_u8(r0, s0, 0));
    return oct_sum(s);
}

// ------------------------------------------------------------------ k_me_pre
#define ME_WIDE_LDS 1156  // (2*16+2)^2: WindowSize <= 32 takes the LDS route (4.6 KB per wave)
#define ME_SEL_NB 25      // 64-candidate batches of stage 3 at WindowSize 32: 18 wide + 7 local
template <int WIN>  // WindowSize known at compile time (0 = read it from d): divisions by the window become shifts/muls
__global__ __launch_bounds__(64, 5) void k_me_pre(FerDev d)
{
    const int window = WIN ? WIN : d.window;
    __shared__ int wide_m[ME_WIDE_LDS];
    __shared__ int sel_lds[256];
    const int lane = threadIdx.x;
    const int s = blockIdx.y;
    if (d.hdr[s * 4 + 3] != 0) return;
    const int mb = blockIdx.x >> 2, part = blockIdx.x & 3;
    const int W = d.W, H = d.H;
    const size_t ysz = d.ysz;
    const uint8_t *Y = d.curY + (size_t)s * ysz;
    const uint8_t *Ps = d.interp + (size_t)s * 16 * ysz;
    const uint16_t *Fs = d.feat + (size_t)s * 96 * ysz;
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

    // source rows for the SAD groups (sx is a multiple of 8: aligned dwords)
    const int row = lane & 7;
    const uint32_t src0 = *(const uint32_t *)(Y + (size_t)(sy + row) * W + sx);
    const uint32_t src1 = *(const uint32_t *)(Y + (size_t)(sy + row) * W + sx + 4);

    // ---- stage 3: MEstimation(+-W/2, frac 0, centre 0) then MEstimation(+-W/16, 16 fracs, centre 0)
    WList L;
    L.m = INF_M;
    L.xy = 0;
    const int R = window / 2, n = 2 * R + 1;
    const int r2 = window / 16, n2w = 2 * r2 + 1, nloc = n2w * n2w * 16;
    const uint16_t *F0 = d.feat0 + (size_t)s * 6 * ysz;
    const int wb = (n * n + 63) >> 6;  // batches of the wide search
    if (n * n <= ME_WIDE_LDS && wb + ((nloc + 63) >> 6) <= ME_SEL_NB && !(d.dbg & 3)) {
        // pass A: lanes run along x (contiguous 12-byte records); metrics land in LDS at their
        // arrival index (tx outer, ty inner).
        for (int base = 0; base < n * n; base += 64) {
            int c = base + lane;
            if (c < n * n) {
                int iy = c / n, ix = c % n;
                int tx = ix - R, ty = iy - R;
                int rx = sx + tx, ry = sy + ty;
                int m = -1;
                if (rx >= 0 && rx < W && ry >= 0 && ry < H)
                    m = (iabs(tx) + iabs(ty) + 4) * feat_dist_rec(F0 + ((size_t)ry * W + rx) * 6, su);
                wide_m[ix * n + iy] = m;
            }
        }
        __syncthreads();
        int v[ME_SEL_NB], xy[ME_SEL_NB];
#pragma unroll
        for (int u = 0; u < ME_SEL_NB; u++) {
            v[u] = -1;
            xy[u] = 0;
            if (u < wb) {
                int c = u * 64 + lane;
                if (c < n * n) {
                    v[u] = wide_m[c];
                    xy[u] = pack_xy((c / n - R) * 4, (c % n - R) * 4);
                }
            } else {
                int c = (u - wb) * 64 + lane;
                int frac = c & 15, pos = c >> 4;
                int tx = pos / n2w - r2, ty = pos % n2w - r2;
                int rx = sx + tx, ry = sy + ty;
                if (c < nloc && rx >= 0 && rx < W && ry >= 0 && ry < H) {
                    v[u] = (iabs(tx) + iabs(ty) + 4) * feat_dist(Fs, ysz, W, frac, ry, rx, su);
                    xy[u] = pack_xy(tx * 4 + (frac & 3), ty * 4 + (frac >> 2));
                }
            }
        }
        select_topk<ME_SEL_NB>(v, xy, 33, lane, sel_lds, L);
    } else {
        for (int base = 0; base < n * n && !(d.dbg & 1); base += 64) {
            int c = base + lane;
            int tx = c / n - R, ty = c % n - R;
            int rx = sx + tx, ry = sy + ty;
            bool ok = c < n * n && rx >= 0 && rx < W && ry >= 0 && ry < H;
            int m = 0;
            if (ok) m = (iabs(tx) + iabs(ty) + 4) * feat_dist_rec(F0 + ((size_t)ry * W + rx) * 6, su);
            wl_insert(L, 33, lane, ok, m, pack_xy(tx * 4, ty * 4));
        }
        for (int base = 0; base < nloc && !(d.dbg & 2); base += 64) {
            int c = base + lane;
            int frac = c & 15, pos = c >> 4;
            int tx = pos / n2w - r2, ty = pos % n2w - r2;
            int rx = sx + tx, ry = sy + ty;
            bool ok = c < nloc && rx >= 0 && rx < W && ry >= 0 && ry < H;
            int m = 0;
            if (ok) m = (iabs(tx) + iabs(ty) + 4) * feat_dist(Fs, ysz, W, frac, ry, rx, su);
            wl_insert(L, 33, lane, ok, m, pack_xy(tx * 4 + (frac & 3), ty * 4 + (frac >> 2)));
        }
    }
    int n3 = __popcll(__ballot(lane < 33 && L.m < 100000000));
    for (int base = 0; base < n3 && !(d.dbg & 4); base += 8) {
        int j = base + (lane >> 3);
        int xy = __shfl(L.xy, j < 33 ? j : 0);
        int bx = unp_x(xy), by = unp_y(xy);
        int sad = sad8_rows(Ps, ysz, W, H, sx, sy, bx, by, row, src0, src1);
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
    const uint2 *srec = d.sort_rec + (size_t)s * ysz;
    const uint32_t *sk34 = d.sort_k34 + (size_t)s * ysz;
    int tren = 0;
    if (!d.basic && !(d.dbg & 8)) {
        int kl0 = 0, kl1 = 0, kh0 = 0, kh1 = 0;
        for (int j = 0; j <= 180; j++) {
            if ((j & 63) == 0) {  // bucket bounds of the next 64 steps on both sides, one lane per step
                int al = su[0] - (j + lane), ah = su[0] + (j + lane);
                bool vl = al >= 0 && al < 16384, vh = ah >= 0 && ah < 16384;
                kl0 = vl ? kol[al] : 0;
                kl1 = vl ? kol[al + 1] : 0;
                kh0 = vh ? kol[ah] : 0;
                kh1 = vh ? kol[ah + 1] : 0;
            }
            for (int side = 0; side < 2; side++) {
                int a = side ? su[0] + j : su[0] - j;
                if (a < 0 || a >= 16384) continue;
                int k0 = lane_bcast(side ? kh0 : kl0, j & 63), k1 = lane_bcast(side ? kh1 : kl1, j & 63);
                // A bucket is ordered by (tx, ty) and the filter needs |tx - sx| < 280: probe 64
                // evenly spaced entries once and walk only the slice whose tx can pass (the exact
                // filter below still decides, so the candidate set and its order are unchanged).
                if (k1 - k0 > 128) {
                    int len = k1 - k0;
                    int pk = k0 + (int)(((long long)len * lane) >> 6);
                    int ptx = (int)(srec[pk].x >> 16);
                    int nlo = __popcll(__ballot(ptx <= sx - 280));  // probes certainly left of the window
                    int nhi = __popcll(__ballot(ptx < sx + 280));   // probes not yet right of it
                    int s0 = nlo > 0 ? k0 + (int)(((long long)len * (nlo - 1)) >> 6) : k0;
                    int s1 = nhi < 64 ? k0 + (int)(((long long)len * nhi) >> 6) : k1;
                    k0 = s0;
                    k1 = s1;
                }
                for (int base = k0; base < k1; base += 64) {
                    int k = base + lane;
                    bool ok = false;
                    int tx = 0, ty = 0, q1 = 0, q2 = 0;
                    if (k < k1) {
                        uint2 e = srec[k];
                        int ax = (int)(e.x >> 16), ay = (int)(e.x & 0xffff);
                        q1 = (int)(e.y & 0xffff);
                        q2 = (int)(e.y >> 16);
                        tx = ax - sx;
                        ty = ay - sy;
                        ok = iabs(tx) + iabs(ty) < 280 && iabs(q1 - su[1]) < 100 && iabs(q2 - su[2]) < 100;
                    }
                    unsigned long long mk = __ballot(ok);
                    if (mk) {
                        int rank = tren + __popcll(mk & ((1ull << lane) - 1));
                        if (ok && rank < FER_ST2_CAP) {  // feature distance from the sorted payload: k0 == a, no scattered reads
                            uint32_t r = sk34[k];
                            int q3 = (int)(r & 0xffff), q4 = (int)(r >> 16);
                            int D = iabs(su[0] - a) + iabs(su[1] - q1) + iabs(su[0] - su[1] - a + q1) + iabs(su[2] - q2) +
                                    iabs(su[0] - su[2] - a + q2) + iabs(su[3] - q3) + iabs(su[0] - su[3] - a + q3) +
                                    iabs(su[4] - q4) + iabs(su[0] - su[4] - a + q4);
                            int *o = d.st2 + (pidx * FER_ST2_CAP + rank) * 2;
                            o[0] = pack_xy(tx, ty);
                            o[1] = D;
                        }
                        tren += __popcll(mk);
                    }
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

// SAD of up to K list entries (8 per round, all rounds' loads issued before any reduction) and
// the ordered first-minimum cost update of F/moestimation.cpp:460-468: cost = SAD + |mv - mvp|,
// strict < in list order, expressed as the minimum of (cost << 6 | list index).
template <int K>
__device__ __forceinline__ void eval_list(const WList &L, int cnt, int lane, const uint8_t *__restrict__ Ps,
                                          size_t ysz, int W, int H, int sx, int sy, uint32_t src0, uint32_t src1,
                                          int mvpx, int mvpy, int &bmin, int &bx, int &by)
{
    constexpr int ROUNDS = (K + 7) / 8;
    const int row = lane & 7;
    int best = 0x7fffffff, bestxy = 0;
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        int j = r * 8 + (lane >> 3);
        int xy = __shfl(L.xy, j < K ? j : 0);
        int cxv = unp_x(xy), cyv = unp_y(xy);
        int sad = sad8_rows(Ps, ysz, W, H, sx, sy, cxv, cyv, row, src0, src1);
        int key = j < cnt ? ((sad + iabs(cxv - mvpx) + iabs(cyv - mvpy)) << 6) | j : 0x7fffffff;
        if (key < best) {
            best = key;
            bestxy = xy;
        }
    }
    int wmin = wave_min(best);
    if (wmin != 0x7fffffff && (wmin >> 6) < bmin) {
        int src = __ffsll((long long)__ballot(best == wmin)) - 1;
        int xy = lane_bcast(bestxy, src);
        bmin = wmin >> 6;
        bx = unp_x(xy);
        by = unp_y(xy);
    }
}

// ------------------------------------------------------------------ k_me_resolve
// One wavefront per 8x8 partition, launched per anti-diagonal gx + 3*gy of the PARTITION grid:
// a partition needs the final vectors of its left, up, up-right and up-left neighbours only, so
// the serial chain is one partition long (half of what a macroblock-level wavefront would need)
// and twice as many wavefronts are in flight per launch.  Partition 0 of a macroblock also makes
// the P_Skip decision; partition 3 merges, derives mvd and does the final prediction + snapping.
#define ST1_UNROLL 7
template <int WIN>
__global__ __launch_bounds__(64) void k_me_resolve(FerDev d, int diag)
{
    const int window = WIN ? WIN : d.window;
    __shared__ int sel_lds[256];
    const int lane = threadIdx.x;
    const int s = blockIdx.y;
    if (d.hdr[s * 4 + 3] != 0) return;
    // wavefront index f = gx + 3*gy: besides left / up / up-right / up-left, the P_Skip prediction of
    // partition 0 reads the lower-left quadrant of the up-right MACROBLOCK, i.e. partition (gx+2, gy-1)
    const int gw = d.mbw * 2, gh = d.mbh * 2;
    int y_lo = diag - (gw - 1);
    y_lo = y_lo > 0 ? (y_lo + 2) / 3 : 0;
    const int gy = y_lo + blockIdx.x, gx = diag - 3 * gy;
    if (gy >= gh || gx < 0 || gx >= gw) return;
    const int mbx = gx >> 1, mby = gy >> 1, part = (gy & 1) * 2 + (gx & 1);
    const int mb = mby * d.mbw + mbx;
    const int W = d.W, H = d.H, Wc = d.Wc, Hc = d.Hc;
    const size_t ysz = d.ysz, csz = d.csz;
    uint8_t *Y = d.curY + (size_t)s * ysz;
    uint8_t *Cb = d.curCb + (size_t)s * csz, *Cr = d.curCr + (size_t)s * csz;
    const uint8_t *RY = d.refY + (size_t)s * ysz;
    const uint8_t *RCb = d.refCb + (size_t)s * csz, *RCr = d.refCr + (size_t)s * csz;
    const uint8_t *Ps = d.interp + (size_t)s * 16 * ysz;
    const uint16_t *Fs = d.feat + (size_t)s * 96 * ysz;
    int *mbt = d.mb_type + (size_t)s * d.nmb;
    short *mvs = d.mv + (size_t)s * d.nmb * 8;
    const int xp = mbx << 4, yp = mby << 4;

    MvCtx c;
    c.mv = mvs;
    c.mb_type = nullptr;
    c.mbw = d.mbw;
    c.cur = mb;
    c.type = FER_P_SKIP;

    // each lane owns 4 luma samples (lx..lx+3, ly) and one sample of each chroma plane
    const int lx = (lane & 3) * 4, ly = lane >> 2;
    const int cxl = lane & 7, cyl = lane >> 3;

    if (part == 0) {
        // ---- P_Skip candidate, F/mode_pred.cpp:381-402 + F/moestimation.cpp:402-425
        int smx = 0, smy = 0;
        if (!(mb < d.mbw || mbx == 0)) {
            int up = mb - d.mbw, lf = mb - 1;
            bool zu = (mvs[(up * 4 + 2) * 2] | mvs[(up * 4 + 2) * 2 + 1]) == 0;
            bool zl = (mvs[(lf * 4 + 1) * 2] | mvs[(lf * 4 + 1) * 2 + 1]) == 0;
            if (!(zu || zl)) predict_luma(c, 0, smx, smy);
        }
        int srcv[4], pred[4];
        uint32_t sv = *(const uint32_t *)(Y + (size_t)(yp + ly) * W + xp + lx);
#pragma unroll
        for (int k = 0; k < 4; k++) srcv[k] = (sv >> (8 * k)) & 0xff;
        mc_luma4(RY, Ps, ysz, W, H, xp, yp, lx, ly, smx, smy, pred);
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
        if (__all(exact)) {
            // P_Skip: reconstruction == prediction (F/inttransform.cpp:215-231)
            *(uint32_t *)(Y + (size_t)(yp + ly) * W + xp + lx) =
                (uint32_t)pred[0] | ((uint32_t)pred[1] << 8) | ((uint32_t)pred[2] << 16) | ((uint32_t)pred[3] << 24);
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
        if (lane == 0) mbt[mb] = FER_P_8x8ref0;  // also clears a P_Skip left by the previous picture
    } else if (mbt[mb] == FER_P_SKIP) {
        return;
    }

    // ---- search of this 8x8 partition as part of a P_8x8ref0 macroblock
    c.type = FER_P_8x8ref0;
    int mvpx, mvpy;
    predict_luma(c, part, mvpx, mvpy);
    const int genx = mvpx >> 2, geny = mvpy >> 2;
    const int sx = xp + (part & 1) * 8, sy = yp + (part >> 1) * 8;
    const size_t pidx = ((size_t)s * d.nmb + mb) * 4 + part;
    int su[5];
#pragma unroll
    for (int k = 0; k < 5; k++) su[k] = d.suma[pidx * 5 + k];
    const int row = lane & 7;
    const uint32_t src0 = *(const uint32_t *)(Y + (size_t)(sy + row) * W + sx);
    const uint32_t src1 = *(const uint32_t *)(Y + (size_t)(sy + row) * W + sx + 4);
    int bx = 0, by = 0, bmin = 2000000000;

    // stage 1: +-W/16 around the predictor, all 16 fractional planes (K = 17).  The feature
    // records of ST1_UNROLL batches are fetched before the ordered insertion starts.
    WList L;
    L.m = INF_M;
    L.xy = 0;
    const int r1 = window / 16, n1 = 2 * r1 + 1, tot1 = (d.dbg & 16) ? 0 : n1 * n1 * 16;
    if (tot1 <= 64 * ST1_UNROLL) {
        int m[ST1_UNROLL], xy[ST1_UNROLL];
#pragma unroll
        for (int u = 0; u < ST1_UNROLL; u++) {
            int cc = u * 64 + lane;
            int frac = cc & 15, pos = cc >> 4;
            int tx = genx - r1 + pos / n1, ty = geny - r1 + pos % n1;
            int rx = sx + tx, ry = sy + ty;
            m[u] = -1;
            if (cc < tot1 && rx >= 0 && rx < W && ry >= 0 && ry < H)
                m[u] = (iabs(tx - genx) + iabs(ty - geny) + 4) * feat_dist(Fs, ysz, W, frac, ry, rx, su);
            xy[u] = pack_xy(tx * 4 + (frac & 3), ty * 4 + (frac >> 2));
        }
        select_topk<ST1_UNROLL>(m, xy, 17, lane, sel_lds, L);
    } else {
        for (int base = 0; base < tot1; base += 64) {
            int cc = base + lane;
            int frac = cc & 15, pos = cc >> 4;
            int tx = genx - r1 + pos / n1, ty = geny - r1 + pos % n1;
            int rx = sx + tx, ry = sy + ty;
            bool ok = cc < tot1 && rx >= 0 && rx < W && ry >= 0 && ry < H;
            int m = 0;
            if (ok) m = (iabs(tx - genx) + iabs(ty - geny) + 4) * feat_dist(Fs, ysz, W, frac, ry, rx, su);
            wl_insert(L, 17, lane, ok, m, pack_xy(tx * 4 + (frac & 3), ty * 4 + (frac >> 2)));
        }
    }
    int cnt = __popcll(__ballot(lane < 17 && L.m < 100000000));
    eval_list<17>(L, cnt, lane, Ps, ysz, W, H, sx, sy, src0, src1, mvpx, mvpy, bmin, bx, by);

    if (!d.basic && !(d.dbg & 32)) {
        // stage 2: re-rank the precomputed candidate set with the predictor weight (K = 33)
        int n2 = min(d.st2n[pidx], FER_ST2_CAP);
        const int2 *c2 = (const int2 *)(d.st2 + pidx * FER_ST2_CAP * 2);
        {
            int m[FER_ST2_CAP / 64], xy[FER_ST2_CAP / 64];
#pragma unroll
            for (int u = 0; u < FER_ST2_CAP / 64; u++) {
                int cc = u * 64 + lane;
                m[u] = -1;
                xy[u] = 0;
                if (cc < n2) {
                    int2 e = c2[cc];
                    int tx = unp_x(e.x), ty = unp_y(e.x);
                    m[u] = (iabs(tx - genx) + iabs(ty - geny) + 4) * e.y;
                    xy[u] = pack_xy(tx * 4, ty * 4);
                }
            }
            select_topk<FER_ST2_CAP / 64>(m, xy, 33, lane, sel_lds, L);
        }
        cnt = __popcll(__ballot(lane < 33 && L.m < 100000000));
        eval_list<33>(L, cnt, lane, Ps, ysz, W, H, sx, sy, src0, src1, mvpx, mvpy, bmin, bx, by);
        // stage 3: precomputed survivors of the centre-0 searches
        int n3 = (d.dbg & 64) ? 0 : d.st3n[pidx];
        const int *c3 = d.st3 + pidx * 33 * 3;
        int key = 0x7fffffff, cxy = 0;
        if (lane < n3) {
            int cxv = c3[lane * 3], cyv = c3[lane * 3 + 1];
            key = ((c3[lane * 3 + 2] + iabs(cxv - mvpx) + iabs(cyv - mvpy)) << 6) | lane;
            cxy = pack_xy(cxv, cyv);
        }
        int wmin = wave_min(key);
        if (wmin != 0x7fffffff && (wmin >> 6) < bmin) {
            int xy = lane_bcast(cxy, wmin & 63);
            bmin = wmin >> 6;
            bx = unp_x(xy);
            by = unp_y(xy);
        }
    }
    if (lane == 0) {
        mvs[(mb * 4 + part) * 2] = (short)bx;
        mvs[(mb * 4 + part) * 2 + 1] = (short)by;
    }
    if (part != 3) return;

    // ---- last partition: merge, mvd, final prediction, snapping (F/moestimation.cpp:529-584)
    int mvx[4], mvy[4];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        mvx[i] = mvs[(mb * 4 + i) * 2];
        mvy[i] = mvs[(mb * 4 + i) * 2 + 1];
    }
    mvx[3] = bx;
    mvy[3] = by;
    __threadfence_block();
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
    // mvd under the final type.  Own earlier partitions are read from d.mv: quadrant 3 was just
    // stored by lane 0; no prediction of partition <= 3 ever reads quadrant 3 of its own MB.
    c.type = type;
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
    {
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
        mc_luma4(RY, Ps, ysz, W, H, xp, yp, lx, ly, mvx[q], mvy[q], pf);
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) packed |= (uint32_t)(iabs(srcv[k] - pf[k]) < MAXDIFF ? pf[k] : srcv[k]) << (8 * k);
        *(uint32_t *)(Y + (size_t)(yp + ly) * W + xp + lx) = packed;
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
    dim3 g(d.nmb * 4, d.S);
    if (d.window == 32)
        hipLaunchKernelGGL(k_me_pre<32>, g, dim3(64), 0, st, d);
    else if (d.window == 16)
        hipLaunchKernelGGL(k_me_pre<16>, g, dim3(64), 0, st, d);
    else
        hipLaunchKernelGGL(k_me_pre<0>, g, dim3(64), 0, st, d);
}

int fer_me_resolve_launches(const FerDev &d) { return 2 * d.mbw + 3 * (2 * d.mbh - 1); }

void fer_launch_me_resolve(const FerDev &d, hipStream_t st)
{
    int gw = 2 * d.mbw, gh = 2 * d.mbh;
    int ndiag = gw + 3 * (gh - 1);
    int maxk = min(gh, (gw + 2) / 3);
    for (int dg = 0; dg < ndiag; dg++) {
        if (d.window == 32)
            hipLaunchKernelGGL(k_me_resolve<32>, dim3(maxk, d.S), dim3(64), 0, st, d, dg);
        else if (d.window == 16)
            hipLaunchKernelGGL(k_me_resolve<16>, dim3(maxk, d.S), dim3(64), 0, st, d, dg);
        else
            hipLaunchKernelGGL(k_me_resolve<0>, dim3(maxk, d.S), dim3(64), 0, st, d, dg);
    }
}
