// fer_me.hip -- P-macroblock decision (rows a15, a18 of SURVEY.md 8a) on CDNA4.
//
// The reference's interEncoding (F/moestimation.cpp:392-585) is serial over macroblocks
// because every search is centred on, and costed against, the predicted motion vector of
// the partition (F/mode_pred.cpp:252-371).  The work is split here into
//   k_me_pre      neighbour-independent, one wavefront per 8x8 partition (XCD-swizzled so that the partitions of a
//                 picture region share one L2): the box sums of the source block, the complete stage-3 search
//                 (wide integer window read from the plane-0 feature map + local quarter-pel window whose features
//                 are computed on chip from the padded interpolated planes, local_metrics<R1>), its top-33 list
//                 and the SAD of each survivor (lane per candidate, sad_keys<K>);
//   k_me_walk     neighbour-independent, one wavefront per partition: the stage-2 bucket walk over the sum-sorted
//                 16-byte feature records (walk_buckets_q, lane-held batch descriptor tables); partitions whose
//                 candidate set overflows FER_ST2_CAP leave a summary (stop step, per-bucket lower bound, the
//                 distance-0 list) instead of candidates;
//   k_me_spec     neighbour-independent BY SPECULATION, one wavefront per partition: the predictor of every partition is
//                 guessed from the neighbours' SAD-best stage-3 survivors, and the two searches that depend on it --
//                 stage 1 around the predictor, the re-ranking of the stage-2 candidates (resolve_crowded: predictor-
//                 centred ring scan for overflowed partitions) -- run for the guess and leave their lists with SADs;
//   k_me_resolve  neighbour-dependent: the true predictor, P_Skip, and either the pricing of the guessed lists (the
//                 guess was right: a few dozen instructions) or the chain's own search (it was wrong).  Persistent
//                 single-wavefront workgroups take (stream group, partition row) tickets from one queue per XCD,
//                 stealing from the other queues when theirs is empty, and chain along the row and to the row above
//                 through self-validating 64-bit words (chain64): no launch per anti-diagonal, no grid-wide barrier.
//                 (Partition merge, mvd, final motion compensation and source snapping follow in k_p_resid.)
//   k_basic_stat  BasicInterEncoding = 1 only: the counters the discarded exhaustive pass leaves behind.
// The candidate list of MEstimation (F/moestimation.cpp:254-296) is a selection problem here -- the K smallest by
// (metric, arrival) -- solved without serial insertion (select_topk); the ballot-ordered insertion (wl_insert)
// remains for the short serial tails.
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
// that is a selection problem.  The code below sits on the serial chain of k_me_resolve, where one
// wavefront runs alone on its SIMD and every dependent instruction costs its full latency, so
// it is built from few, wide steps:
//   1. T0 = the need-th smallest per-lane minimum -- an upper bound of the need-th smallest
//      candidate (binary search over the value with ballot counts: scalar work),
//   2. the candidates <= T0 (typically 1.2-1.5 x need of them) are gathered in LDS, in no particular order,
//   3. their exact rank by (metric << 11 | arrival index) places the winners.
// More than 64 survivors of step 2 or a T0 >= 2^21 take the binary-search route instead.
// v[u] < 0 marks an invalid candidate; raw(u) is a cheap tag of the lane's candidate u that travels with it,
// fin(tag) turns the tag of a winner into its packed vector (evaluated once per list slot).  sel = 256 ints of 16-byte aligned LDS owned by the wavefront.
// sel is private to ONE wavefront: its LDS operations execute in program order, so ordering them needs no
// s_barrier (which would also be wrong inside the two-wavefront workgroups of k_me_resolve, whose wavefronts
// call this a different number of times) -- only the compiler must not move them across.
#define WAVE_LDS_SYNC()                                    \
    do {                                                   \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                   \
    } while (0)
__device__ __forceinline__ int lds_rank64(const unsigned *p, unsigned mine, int n4 = 16)
{
    // n4 = how many groups of four slots hold keys (wave-uniform); the slots behind them are 0xffffffff, below nothing
    const uint4 *q = (const uint4 *)p;
    int r = 0;
#pragma unroll 4
    for (int i = 0; i < n4; i++) {
        uint4 k = q[i];
        r += (int)(k.x < mine) + (int)(k.y < mine) + (int)(k.z < mine) + (int)(k.w < mine);
    }
    return r;
}

// ORDERED = the registers hold the candidates in arrival order (candidate (u, lane) arrived as number u * 64 + lane).  A caller whose
// candidates sit in any order passes ORDERED = false and arr(u) = the lane's arrival number of candidate u (unique, < 2048): the
// wide steps only need the number inside the sort key; the binary-search route relies on the register order and is not taken then
// -- the function returns false and leaves the selection to the caller.
template <int NB, bool ORDERED, class RAW, class FIN, class ARR>
__device__ __forceinline__ bool select_topk_ex(const int (&v)[NB], int K, int lane, int *sel, WList &L, RAW raw, FIN fin, ARR arr)
{
    unsigned *A = (unsigned *)sel, *B = A + 64;
    int *Ci = sel + 128, *Di = sel + 192;
    const unsigned long long lt = (1ull << lane) - 1ull;
    // (an invalid candidate is -1 = the largest unsigned value: unsigned minima and compares leave it out by themselves,
    // and the valid ones are counted with one compare and a scalar population count per batch)
    int lmin = 0x7fffffff, lmax = -1, total = 0;
#pragma unroll
    for (int u = 0; u < NB; u++) {
        total += __popcll(__ballot(v[u] >= 0));
        lmin = (int)min((unsigned)lmin, (unsigned)v[u]);
        lmax = max(lmax, v[u]);
    }
    const int need = min(K, total);
    L.m = INF_M;
    L.xy = 0;
    if (need == 0) return true;
    int c = 65;
    if (!__any(lmax >= (1 << 26))) {
        B[lane] = 0xffffffffu;
        // T0 = the need-th smallest per-lane minimum -- an upper bound of the need-th smallest candidate -- by binary search
        // over the value: one compare and one scalar population count per step.  (A rank computation over the 64 minima
        // through LDS cost 130 vector instructions here; these kernels are bound by vector issue, the scalar unit has room.)
        int T0 = 0x7fffffff;  // fewer than `need` lanes hold anything: every candidate is needed
        if (__popcll(__ballot(lmin != 0x7fffffff)) >= need) {
            int lo = 0, hi = (1 << 21) - 1;
            if (__popcll(__ballot(lmin <= hi)) >= need) {
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (__popcll(__ballot(lmin <= mid)) >= need)
                        hi = mid;
                    else
                        lo = mid + 1;
                }
                T0 = lo;
            } else {
                T0 = 1 << 21;  // the general route below
            }
        }
        WAVE_LDS_SYNC();
        // The candidates <= T0 are gathered in NO particular order (an LDS counter hands out the slots, batches
        // without survivors cost a compare and a branch); what orders them is the key metric << 11 | arrival index,
        // exact as long as the survivors' metrics stay below 2^21 (they are the smallest ones; larger takes the
        // general route).
        if (T0 < (1 << 21) && NB * 64 <= 2048) {
            int nsurv = 0;  // (slots from a running scalar count and the lane's rank among the takers of its batch: an LDS
                            // counter costs a wave-aggregated atomic, twenty instructions, per batch with a survivor)
#pragma unroll
            for (int u = 0; u < NB; u++) {
                const bool take = (unsigned)v[u] <= (unsigned)T0;  // (T0 < 2^21: an invalid candidate never passes)
                const unsigned long long mk = __ballot(take);
                if (mk) {
                    const int slot = nsurv + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
                    if (take && slot < 64) {
                        B[slot] = ((unsigned)v[u] << 11) | (unsigned)arr(u);
                        Ci[slot] = raw(u);
                    }
                    nsurv += __popcll(mk);
                }
            }
            WAVE_LDS_SYNC();
            c = nsurv;
        }
    }
    if (c <= 64) {
        WAVE_LDS_SYNC();
        const unsigned mine = B[lane];
        const int mypay = Ci[lane];
        const int rk = lds_rank64(B, mine, (__builtin_amdgcn_readfirstlane(c) + 3) >> 2);  // (only the c survivors are keys)
        WAVE_LDS_SYNC();
        if (lane < c && rk < need) {
            A[rk] = mine >> 11;
            Di[rk] = mypay;
        }
        WAVE_LDS_SYNC();
        if (lane < need) {
            L.m = (int)A[lane];
            L.xy = fin(Di[lane]);
        }
        WAVE_LDS_SYNC();
        return true;
    }
    if (!ORDERED) return false;
    // ---- general route: binary-search the need-th smallest metric with ballot counts, take everything
    // below it plus the earliest arrivals equal to it, then rank the winners
    int *key = sel, *kxy = sel + 64, *key2 = sel + 128, *kxy2 = sel + 192;
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
    WAVE_LDS_SYNC();
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
            kxy[pos] = raw(u);
        }
        nsel += __popcll(mt);
        eq_seen += __popcll(meq);
    }
    WAVE_LDS_SYNC();
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
    WAVE_LDS_SYNC();
    if (lane < need) {
        L.m = key2[lane];
        L.xy = fin(kxy2[lane]);
    }
    WAVE_LDS_SYNC();
    return true;
}
template <int NB, class RAW, class FIN>
__device__ __forceinline__ void select_topk(const int (&v)[NB], int K, int lane, int *sel, WList &L, RAW raw, FIN fin)
{
    select_topk_ex<NB, true>(v, K, lane, sel, L, raw, fin, [&](int u) { return u * 64 + lane; });
}

// ---- the 9-term feature distance of F/moestimation.cpp:267-276 on packed 16-bit pairs ----
// A 12-byte record holds (k0,k1) (k2,k3) (k4,0) as u16 pairs, k0 = sum of the 8x8 box and k1..k4 = sums of half
// of its samples, so every k0 - ki is >= 0 and everything fits unsigned 16 bit.  With the source
// sums arranged the same way the nine terms are six v_sad_u16:
//   |s0-k0| + |s1-k1|, |s2-k2| + |s3-k3|, |s4-k4|, and |(s0-si) - (k0-ki)| for i = 1..4.
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
typedef short i16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_sub16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, (u16x2_t)(__builtin_bit_cast(u16x2_t, a) - __builtin_bit_cast(u16x2_t, b)));
}
__device__ __forceinline__ uint32_t pk_abs16(uint32_t a)
{
    i16x2_t x = __builtin_bit_cast(i16x2_t, a), z = {0, 0};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(x, (i16x2_t)(z - x)));
}
struct SuPk {  // source sums of one partition, wave-uniform
    uint32_t s01, s23, s4;  // (s0,s1) (s2,s3) (s4,0)
    uint32_t d01, d23, d4;  // (0,s0-s1) (s0-s2,s0-s3) (s0-s4,0)
    uint32_t s12, s34;      // (s1,s2) (s3,s4): the pairing of the sorted records
    uint32_t e12, e34;      // (s0-s1,s0-s2) (s0-s3,s0-s4)
};
__device__ __forceinline__ SuPk su_pack(const int s[5])
{
    SuPk p;
    p.s01 = (uint32_t)s[0] | ((uint32_t)s[1] << 16);
    p.s23 = (uint32_t)s[2] | ((uint32_t)s[3] << 16);
    p.s4 = (uint32_t)s[4];
    p.d01 = (uint32_t)(s[0] - s[1]) << 16;
    p.d23 = (uint32_t)(s[0] - s[2]) | ((uint32_t)(s[0] - s[3]) << 16);
    p.d4 = (uint32_t)(s[0] - s[4]);
    p.s12 = (uint32_t)s[1] | ((uint32_t)s[2] << 16);
    p.s34 = (uint32_t)s[3] | ((uint32_t)s[4] << 16);
    p.e12 = (uint32_t)(s[0] - s[1]) | ((uint32_t)(s[0] - s[2]) << 16);
    p.e34 = (uint32_t)(s[0] - s[3]) | ((uint32_t)(s[0] - s[4]) << 16);
    return p;
}
__device__ __forceinline__ int feat_dist_w(uint32_t a, uint32_t b, uint32_t c, const SuPk &p)
{
    uint32_t t = __builtin_amdgcn_sad_u16(a, p.s01, 0);
    t = __builtin_amdgcn_sad_u16(b, p.s23, t);
    t = __builtin_amdgcn_sad_u16(c, p.s4, t);
    uint32_t kk = __builtin_amdgcn_perm(a, a, 0x01000100);  // (k0,k0)
    t = __builtin_amdgcn_sad_u16(pk_sub16(kk, a), p.d01, t);
    t = __builtin_amdgcn_sad_u16(pk_sub16(kk, b), p.d23, t);
    t = __builtin_amdgcn_sad_u16(pk_sub16(kk, c) & 0xffffu, p.d4, t);
    return (int)t;
}
__device__ __forceinline__ int feat_dist_rec(const uint16_t *__restrict__ rec, const SuPk &p)
{
    const uint32_t *r = (const uint32_t *)rec;
    return feat_dist_w(r[0], r[1], r[2], p);
}
// Record fetch without control flow: coordinates are clamped into the picture so that the load
// can always be issued (the caller masks candidates outside the picture afterwards).  Loads
// under a branch are waited for one by one; a wavefront that runs alone on its SIMD cannot
// afford that.
struct FeatRec {
    uint32_t a, b, c;
};

// ---- the five box features of one (position, quarter-pel plane) straight from the interpolated plane ----
// refFrameKar[0..4][frac][refy][refx] of F/moestimation.cpp:105-138 for one candidate per lane: eight rows of eight
// samples (the replicated margin of the planes stands in for the reference's 8-pixel padding).  ~80 instructions and
// 8 loads per candidate: the general route (WindowSize other than 16 / 32); the searches of the specialised kernels
// share the row sums of neighbouring candidates through LDS (local_metrics below).
__device__ __forceinline__ FeatRec feat_rec_direct(const IPlanes &ip, int W, int H, int frac, int refy, int refx)
{
    const int y = iclamp(refy, 0, H - 1), x = iclamp(refx, 0, W - 1);
    const uint8_t *p0 = ip.base + (size_t)frac * ip.plane + (size_t)y * ip.pitch + (x & ~3);
    const uint32_t sh = (uint32_t)(x & 3);
    uint32_t h84[8], hc[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint32_t *w = (const uint32_t *)(p0 + (size_t)r * ip.pitch);
        const uint32_t lo = __builtin_amdgcn_alignbyte(w[1], w[0], sh), hi = __builtin_amdgcn_alignbyte(w[2], w[1], sh);
        const uint32_t a4 = __builtin_amdgcn_sad_u8(lo, 0u, 0u);
        h84[r] = __builtin_amdgcn_sad_u8(hi, 0u, a4) | (a4 << 16);
        hc[r] = __builtin_amdgcn_udot4(hi, 0x00000101u, __builtin_amdgcn_udot4(lo, 0x00000101u, 0u, false), false);
    }
    const uint32_t top = h84[0] + h84[1] + h84[2] + h84[3], all = top + h84[4] + h84[5] + h84[6] + h84[7];
    const uint32_t k3 = (h84[0] + h84[1] + h84[4] + h84[5]) & 0xffffu;
    const uint32_t k4 = hc[0] + hc[1] + hc[2] + hc[3] + hc[4] + hc[5] + hc[6] + hc[7];
    FeatRec f;
    f.a = (all & 0xffffu) | (top << 16);  // k0 | k1 << 16
    f.b = (all >> 16) | (k3 << 16);       // k2 | k3 << 16
    f.c = k4;
    return f;
}

// ---- MEstimation(sx, sy, granica = R1, stepMV 1, stepFrac 1, genx, geny, px, py) without a feature table ----
// The (2 R1 + 1)^2 integer positions around the centre on all 16 quarter-pel planes (F/moestimation.cpp:254-296):
// candidate c = (ix * N1 + iy) * 16 + frac in the reference's loop order, metric (|ix - R1| + |iy - R1| + 4) * D
// (both searches of interEncoding that use this shape weigh from their own centre), -1 outside the picture.
// The box features are built on chip: for 8 planes at a time, lane = (plane, row) takes one 16-byte row of the
// (2 R1 + 8)^2 patch of its plane and leaves the horizontal sums of every column offset in LDS; lane = (plane, column
// offset) then runs down its column with prefix sums and emits the N1 metrics.  16-byte loads cover the patch for
// R1 <= 2 (WindowSize 16 and 32).  htab: 8 * PW * N1 * 2 dwords, mtab: NB * 64 ints, both private to the wavefront.
template <int R1>
struct LocalGeo {
    static constexpr int N1 = 2 * R1 + 1, PW = 2 * R1 + 8, NC = N1 * N1 * 16, NB = (NC + 63) / 64;
    static constexpr int HTAB = 8 * PW * N1 * 2;
};
// The 16-byte patch rows of all 16 planes, requested ahead of their use: row task t = t0 + lane of either group of 8 planes
// (two rounds of 64 tasks per group).  A caller that knows the search centre early (k_me_spec, k_me_pre) issues these loads
// before it waits for anything else, so that their round trip overlaps the ones in front of the search.
template <int R1>
struct LocalLoads {
    static constexpr int ROUNDS = (8 * LocalGeo<R1>::PW + 63) / 64;
    uint4 w[2][ROUNDS];
};
template <int R1>
__device__ __forceinline__ void local_request(const IPlanes &ip, int W, int H, int px0, int py0, int lane, LocalLoads<R1> &ld)
{
    using G = LocalGeo<R1>;
    constexpr int N1 = G::N1, PW = G::PW;
#pragma unroll
    for (int half = 0; half < 2; half++)
#pragma unroll
        for (int k = 0; k < LocalLoads<R1>::ROUNDS; k++) ld.w[half][k] = make_uint4(0, 0, 0, 0);
    if (px0 + N1 - 1 < 0 || px0 >= W || py0 + N1 - 1 < 0 || py0 >= H) return;
    const uint8_t *mbase = ip.base - ((size_t)FER_IP_T * ip.pitch + FER_IP_L);
    const uint32_t obase = __umul24((uint32_t)(py0 + FER_IP_T), (uint32_t)ip.pitch) + (uint32_t)((px0 & ~3) + FER_IP_L);
#pragma unroll
    for (int half = 0; half < 2; half++) {
#pragma unroll
        for (int k = 0; k < LocalLoads<R1>::ROUNDS; k++) {
            const int tt = min(k * 64 + lane, 8 * PW - 1);
            const int fl = tt / PW, r = tt - fl * PW;
            typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
            const u32x4_a4 v = *(const u32x4_a4 *)(mbase + (obase + __umul24((uint32_t)(half * 8 + fl), (uint32_t)ip.plane) + __umul24((uint32_t)r, (uint32_t)ip.pitch)));
            ld.w[half][k] = make_uint4(v.x, v.y, v.z, v.w);
        }
    }
}
// half_lo .. half_hi: which groups of 8 planes this call computes (two wavefronts may split them, each with its own htab; the
// caller orders the shared mtab).  pre: the patch rows requested earlier with local_request (all 16 planes), or null.
template <int R1>
__device__ __forceinline__ void local_metrics(const IPlanes &ip, int W, int H, int px0, int py0, const SuPk &sp, int lane,
                                              uint32_t *htab, int *mtab, int half_lo = 0, int half_hi = 2, const LocalLoads<R1> *pre = nullptr)
{
    using G = LocalGeo<R1>;
    constexpr int N1 = G::N1, PW = G::PW;
    static_assert(PW + 3 <= 16, "the patch row must fit one 16-byte load");
    // px0, py0 = picture position of candidate (ix, iy) = (0, 0); nothing to do when no candidate is inside
    if (px0 + N1 - 1 < 0 || px0 >= W || py0 + N1 - 1 < 0 || py0 >= H) {
#pragma unroll
        for (int u = 0; u < G::NB; u++) mtab[u * 64 + lane] = -1;
        WAVE_LDS_SYNC();
        return;
    }
    const uint32_t sh = (uint32_t)(px0 & 3);
    // (a 32-bit byte offset from the stream's uniform plane base; the top-left of the patch may lie in the margin: ip.base
    // points at sample (0, 0) behind it, so the offset is taken from the start of the margin)
    const uint8_t *mbase = ip.base - ((size_t)FER_IP_T * ip.pitch + FER_IP_L);
    const uint32_t obase = __umul24((uint32_t)(py0 + FER_IP_T), (uint32_t)ip.pitch) + (uint32_t)((px0 & ~3) + FER_IP_L);
    for (int half = half_lo; half < half_hi; half++) {
        // ---- horizontal sums: row tasks (plane, row)
#pragma unroll
        for (int t0 = 0; t0 < 8 * PW; t0 += 64) {
            const int t = t0 + lane;
            const int tt = min(t, 8 * PW - 1);
            const int fl = tt / PW, r = tt - fl * PW;
            typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
            uint4 w;
            if (pre) {
                w = half == 0 ? pre->w[0][t0 / 64] : pre->w[1][t0 / 64];
            } else {
                const u32x4_a4 v = *(const u32x4_a4 *)(mbase + (obase + __umul24((uint32_t)(half * 8 + fl), (uint32_t)ip.plane) + __umul24((uint32_t)r, (uint32_t)ip.pitch)));
                w = make_uint4(v.x, v.y, v.z, v.w);
            }
            const uint32_t b0 = __builtin_amdgcn_alignbyte(w.y, w.x, sh), b1 = __builtin_amdgcn_alignbyte(w.z, w.y, sh),
                           b2 = __builtin_amdgcn_alignbyte(w.w, w.z, sh);
            if (t < 8 * PW) {
                uint32_t *o = htab + (size_t)tt * N1 * 2;
#pragma unroll
                for (int ix = 0; ix < N1; ix++) {
                    const uint32_t lo = ix < 4 ? __builtin_amdgcn_alignbyte(b1, b0, (uint32_t)(ix & 3)) : b1;
                    const uint32_t hi = ix < 4 ? __builtin_amdgcn_alignbyte(b2, b1, (uint32_t)(ix & 3)) : b2;
                    const uint32_t a4 = __builtin_amdgcn_sad_u8(lo, 0u, 0u);
                    o[ix * 2] = __builtin_amdgcn_sad_u8(hi, 0u, a4) | (a4 << 16);
                    o[ix * 2 + 1] = __builtin_amdgcn_udot4(hi, 0x00000101u, __builtin_amdgcn_udot4(lo, 0x00000101u, 0u, false), false);  // samples {0,1,4,5}
                }
            }
        }
        WAVE_LDS_SYNC();
        // ---- column tasks (plane, column offset): prefix sums down the PW rows, N1 metrics
        if (lane < 8 * N1) {
            const int fl = lane / N1, ix = lane - fl * N1;
            uint32_t P84[PW + 1], Pc[PW + 1];
            P84[0] = 0;
            Pc[0] = 0;
#pragma unroll
            for (int r = 0; r < PW; r++) {
                const uint2 h = *(const uint2 *)(htab + ((size_t)(fl * PW + r) * N1 + ix) * 2);
                P84[r + 1] = P84[r] + h.x;  // (sum of h8) | (sum of h4) << 16: neither field can carry
                Pc[r + 1] = Pc[r] + h.y;
            }
            const int refx = px0 + ix, wx = iabs(ix - R1) + 4;
            const bool xok = refx >= 0 && refx < W;
#pragma unroll
            for (int iy = 0; iy < N1; iy++) {
                const uint32_t all = P84[iy + 8] - P84[iy];
                const uint32_t k1 = (P84[iy + 4] - P84[iy]) & 0xffffu;
                const uint32_t k3 = ((P84[iy + 2] - P84[iy]) + (P84[iy + 6] - P84[iy + 4])) & 0xffffu;
                const uint32_t k4 = Pc[iy + 8] - Pc[iy];
                const int D = feat_dist_w((all & 0xffffu) | (k1 << 16), (all >> 16) | (k3 << 16), k4, sp);
                const int refy = py0 + iy;
                const int m = (int)__umul24((uint32_t)(wx + iabs(iy - R1)), (uint32_t)D) | ((xok && refy >= 0 && refy < H) ? 0 : -1);
                mtab[((ix * N1 + iy) << 4) + half * 8 + fl] = m;
            }
        }
        WAVE_LDS_SYNC();
    }
    if (half_lo == 0) {
#pragma unroll
        for (int c = G::NC + lane; c < G::NB * 64; c += 64) mtab[c] = -1;  // the padding of the last batch
    }
    WAVE_LDS_SYNC();
}

// ... from the plane-0 copy [H][W][6]
__device__ __forceinline__ FeatRec feat0_load(const uint16_t *__restrict__ F0, int W, int H, int refy, int refx)
{
    int y = iclamp(refy, 0, H - 1), x = iclamp(refx, 0, W - 1);
    const uint32_t *r = (const uint32_t *)((const char *)F0 + (__umul24((uint32_t)y, (uint32_t)W * 12u) + (uint32_t)x * 12u));  // (a stream's map is < 4 GB, a row < 2^24 bytes)
    FeatRec f;
    f.a = r[0];
    f.b = r[1];
    f.c = r[2];
    return f;
}

// SAD of the 8x8 source block against the interpolated plane at the lane's OWN vector (satdLuma8x8MVs,
// F/moestimation.cpp:175-195), ONE CANDIDATE PER LANE: the lane walks the eight rows of the block; the source block
// is wave-uniform (scalar registers), so nothing crosses lanes, and a list of 17 or 33 candidates is one pass.  The
// block origin is clamped into the picture (the reference shifts the block at the left / top edges, :181-182); rows
// and columns beyond the right / bottom edge come from the planes' replicated margin (= the per-sample clamp :189-190).
struct SrcBlk {
    uint32_t lo[8], hi[8];
};
__device__ __forceinline__ SrcBlk src_block_load(const uint8_t *__restrict__ Y, int W, int sx, int sy)
{
    SrcBlk b;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint32_t *q = (const uint32_t *)(Y + (size_t)(sy + r) * W + sx);  // wave-uniform address: scalar loads
        b.lo[r] = __builtin_amdgcn_readfirstlane((int)q[0]);
        b.hi[r] = __builtin_amdgcn_readfirstlane((int)q[1]);
    }
    return b;
}
__device__ __forceinline__ int sad_lane(const IPlanes &ip, int W, int H, int sx, int sy, int mvx, int mvy, const SrcBlk &S)
{
    const int xPi = iclamp(sx + (mvx >> 2), 0, W - 1), yPi = iclamp(sy + (mvy >> 2), 0, H - 1);
    const uint32_t sh = (uint32_t)(xPi & 3);  // planes and rows start on 16-byte boundaries
    // (32-bit byte offsets from the stream's uniform plane base: the 16 planes of a stream are < 4 GB)
    const uint32_t o0 = __umul24((uint32_t)((mvx & 3) + (mvy & 3) * 4), (uint32_t)ip.plane) + __umul24((uint32_t)yPi, (uint32_t)ip.pitch) + (uint32_t)(xPi & ~3);
    uint32_t w[8][3];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint32_t *a = (const uint32_t *)(ip.base + (o0 + __umul24((uint32_t)r, (uint32_t)ip.pitch)));
        w[r][0] = a[0];
        w[r][1] = a[1];
        w[r][2] = a[2];
    }
    uint32_t sad = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint32_t r0 = __builtin_amdgcn_alignbyte(w[r][1], w[r][0], sh), r1 = __builtin_amdgcn_alignbyte(w[r][2], w[r][1], sh);
        sad = __builtin_amdgcn_sad_u8(r1, S.hi[r], __builtin_amdgcn_sad_u8(r0, S.lo[r], sad));
    }
    return (int)sad;
}

// ------------------------------------------------------------------ k_me_pre
#define ME_WIDE_LDS 1156  // (2*16+2)^2: WindowSize <= 32 takes the LDS route (4.6 KB per wave)
#define ME_SEL_NB 25      // 64-candidate batches of stage 3 at WindowSize 32: 18 wide + 7 local
#ifndef PRE_WAVES
#define PRE_WAVES 5  // wavefronts per SIMD the kernel is compiled for (register budget; flat from 5 up, its LDS allows 6.25)
#endif
template <int WIN>  // WindowSize known at compile time (0 = read it from d): divisions by the window become shifts/muls
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PRE_WAVES, PRE_WAVES))) void k_me_pre(FerDev d)
{
    const int window = WIN ? WIN : d.window;
    // LDS of the wavefront: region A = the row sums of the local search, then the wide search's metrics; region B =
    // the local search's metrics, then the scratch of the selection
    __shared__ __attribute__((aligned(16))) int wide_m[ME_WIDE_LDS];
    __shared__ __attribute__((aligned(16))) int sel_lds[448];
    static_assert(LocalGeo<2>::HTAB <= ME_WIDE_LDS && LocalGeo<2>::NB * 64 <= 448, "LDS regions of k_me_pre");
    const int lane = threadIdx.x;
    const int s = blockIdx.y;
    if (d.hdr[s * 4 + 3] != 0) return;
    // workgroups are dealt round-robin over the 8 XCDs: give each XCD a contiguous eighth of the picture, so that the
    // search windows of the partitions it works on overlap in ITS L2
    const unsigned bx = xcd_swizzle(blockIdx.x, gridDim.x);
    const int mb = (int)(bx >> 2), part = (int)(bx & 3);
    const int W = d.W, H = d.H;
    const size_t ysz = d.ysz;
    const uint8_t *Y = d.curY + (size_t)s * ysz;
    const IPlanes ip = ip_stream(d, s);
    const int sx = ((mb % d.mbw) << 4) + (part & 1) * 8, sy = ((mb / d.mbw) << 4) + (part >> 1) * 8;
    const size_t pidx = ((size_t)s * d.nmb + mb) * 4 + part;

#ifdef FER_PROBE
    const bool probe = FER_DBGF(d, 128) && s == 0 && (bx % 997) == 5;  // a sample of partitions reports its time split
    long long tmark = probe ? wall_clock64() : 0;
#define PP_MARK(k)                                                          \
    if (probe) {                                                            \
        long long now_ = wall_clock64();                                    \
        if (lane == 0) atomicAdd((unsigned long long *)&d.timing[32 + k], (unsigned long long)(now_ - tmark)); \
        tmark = now_;                                                       \
    }
#else
#define PP_MARK(k)
#endif
    // the patch rows of the local search are requested before anything is waited for (their addresses depend on the
    // position only)
    constexpr int RRL = (WIN == 16) ? 1 : 2;
    LocalLoads<RRL> ld;
    const bool fast_path = (WIN == 32 || WIN == 16) && !FER_DBGF(d, 3) && !d.basic;
    if (fast_path) local_request<RRL>(ip, W, H, sx - RRL, sy - RRL, lane, ld);
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
    if (d.basic) {  // BasicInterEncoding: stages 2 and 3 are not run (F/moestimation.cpp:470), only the sums are needed
        if (lane == 0) {
            d.st3n[pidx] = 0;
            d.v0[pidx] = 0;
        }
        return;
    }
    const SuPk sp = su_pack(su);
    PP_MARK(0)

    // ---- stage 3: MEstimation(+-W/2, frac 0, centre 0) then MEstimation(+-W/16, 16 fracs, centre 0)
    WList L;
    L.m = INF_M;
    L.xy = 0;
    const int R = window / 2, n = 2 * R + 1;
    const int r2 = window / 16, n2w = 2 * r2 + 1, nloc = n2w * n2w * 16;
    const uint16_t *F0 = d.feat0 + (size_t)s * 6 * ysz;
    const int wb = (n * n + 63) >> 6;  // batches of the wide search
    if ((WIN == 32 || WIN == 16) && !FER_DBGF(d, 3)) {
        // n = WIN + 1: a batch of the wide search is 64 / WIN rows of the first WIN columns, so the column, its validity and its
        // weight are per-lane constants and a row step is one address increment; the last column follows
        constexpr int WCH = 6;
        constexpr int NN = (WIN ? WIN : 32) + 1;
        constexpr int NBc = WIN ? WIN : 32, RPB = 64 / NBc, NWB = (NN + RPB - 1) / RPB;
        constexpr int LB = LocalGeo<RRL>::NB;  // batches of the local search
        static_assert(LB <= 7, "ME_SEL_NB leaves room for 7 local batches");
        const int ix = lane % NBc, r0 = lane / NBc;
        const int rx = sx - R + ix;
        const bool xok = rx >= 0 && rx < W;
        const int wx = iabs(ix - R) + 4;
        // (32-bit byte offsets from the stream's uniform base: a record load is one instruction with a scalar base,
        // where 64-bit index arithmetic costs nine VALU instructions per load)
        const uint32_t colo = (uint32_t)iclamp(rx, 0, W - 1) * 12u, rowb = (uint32_t)W * 12u;
        auto fin = [&](int idx) {  // arrival index -> the candidate's vector
            if (idx < wb * 64) return pack_xy((idx / n - R) * 4, (idx % n - R) * 4);
            int c = idx - wb * 64;
            int frac = c & 15, pos = c >> 4;
            return pack_xy((pos / n2w - r2) * 4 + (frac & 3), (pos % n2w - r2) * 4 + (frac >> 2));
        };
        // ---- Pruning by a lower bound.  metric = w * D with D = |s0 - k0| + sum over the four half sums of
        // (|si - ki| + |(s0 - si) - (k0 - ki)|) >= 5 |s0 - k0| (each pair is at least |s0 - k0| by the triangle inequality),
        // so w * 5 |s0 - k0| bounds a candidate from below with the FIRST dword of its record.  With T = an upper bound of
        // the 33rd smallest metric (the 33rd smallest per-lane minimum of the local search, which runs first), a wide
        // candidate whose bound exceeds T cannot enter the list (33 candidates are strictly better); the others -- a few
        // dozen of the 1089 -- are evaluated in full.  More than 128 survivors (flat content: every bound is 0) or a
        // selection that cannot take its wide route fall back to the full evaluation below.
        uint32_t k0w[NWB];
#pragma unroll
        for (int b2 = 0; b2 < NWB; b2++) {  // in flight during the local search
            const int ry = iclamp(sy - R + b2 * RPB + r0, 0, H - 1);
            k0w[b2] = *(const uint32_t *)((const char *)F0 + (colo + __umul24((uint32_t)ry, rowb)));
        }
        const uint32_t k0c = feat0_load(F0, W, H, sy - R + min(lane, NN - 1), sx + R).a;  // column n - 1
        // the local search first (its row sums borrow the LDS of the wide search's metrics)
        local_metrics<RRL>(ip, W, H, sx - r2, sy - r2, sp, lane, (uint32_t *)wide_m, sel_lds, 0, 2, &ld);
        int vloc[LB];
        unsigned lminl = 0xffffffffu;
#pragma unroll
        for (int u = 0; u < LB; u++) {
            vloc[u] = sel_lds[u * 64 + lane];
            lminl = min(lminl, (unsigned)vloc[u]);
        }
        int T = 0x7fffffff;
        if (__popcll(__ballot(lminl < (1u << 21))) >= 33) {
            int lo = 0, hi = (1 << 21) - 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (__popcll(__ballot(lminl <= (unsigned)mid)) >= 33)
                    hi = mid;
                else
                    lo = mid + 1;
            }
            T = lo;
        }
        int *surv = wide_m;  // arrival numbers of the surviving wide candidates; the row sums are dead
        int ns = 0;
        const uint32_t s0 = (uint32_t)su[0];
        bool done = false;
        if (T != 0x7fffffff) {
#pragma unroll
            for (int b2 = 0; b2 < NWB; b2++) {
                const int iy = b2 * RPB + r0;
                const int ry = sy - R + iy;
                const uint32_t lb = __umul24(5u * (uint32_t)(wx + iabs(iy - R)), __builtin_amdgcn_sad_u16(k0w[b2] & 0xffffu, s0, 0u));
                const bool take = xok && ry >= 0 && ry < H && iy < NN && lb <= (uint32_t)T;
                const unsigned long long mk = __ballot(take);
                if (mk) {
                    const int slot = ns + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
                    if (take && slot < 128) surv[slot] = ix * NN + iy;
                    ns += __popcll(mk);
                }
            }
            {  // column n - 1
                const int iy = lane, ry = sy - R + iy;
                const uint32_t lb = __umul24(5u * (uint32_t)(R + 4 + iabs(iy - R)), __builtin_amdgcn_sad_u16(k0c & 0xffffu, s0, 0u));
                const bool take = iy < NN && sx + R < W && ry >= 0 && ry < H && lb <= (uint32_t)T;
                const unsigned long long mk = __ballot(take);
                if (mk) {
                    const int slot = ns + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
                    if (take && slot < 128) surv[slot] = (NN - 1) * NN + iy;
                    ns += __popcll(mk);
                }
            }
            WAVE_LDS_SYNC();
            if (ns <= 128) {
                int v[2 + LB], ar[2];
#pragma unroll
                for (int b2 = 0; b2 < 2; b2++) {
                    v[b2] = -1;
                    ar[b2] = 0;
                    if (b2 * 64 < ns) {  // (the second batch is rarely needed)
                        const int idx = b2 * 64 + lane;
                        const bool on = idx < ns;
                        const int ai = on ? surv[idx] : 0;
                        const int cix = ai / NN, ciy = ai - cix * NN;
                        const FeatRec f = feat0_load(F0, W, H, sy - R + ciy, sx - R + cix);
                        const int m = (int)__umul24((uint32_t)(iabs(cix - R) + iabs(ciy - R) + 4), (uint32_t)feat_dist_w(f.a, f.b, f.c, sp));
                        v[b2] = on ? m : -1;
                        ar[b2] = ai;
                    }
                }
#pragma unroll
                for (int u = 0; u < LB; u++) v[2 + u] = vloc[u];
                WAVE_LDS_SYNC();  // the selection reuses sel_lds
                auto arrv = [&](int u) { return u < 2 ? ar[u < 2 ? u : 0] : wb * 64 + (u - 2) * 64 + lane; };
                done = select_topk_ex<2 + LB, false>(v, 33, lane, sel_lds, L, arrv, fin, arrv);
            }
        }
        PP_MARK(1)
#ifdef FER_STATS
        if (lane == 0) {
            atomicAdd((unsigned long long *)&d.timing[48], 1ull);
            atomicAdd((unsigned long long *)&d.timing[49], done ? 0ull : 1ull);
            atomicAdd((unsigned long long *)&d.timing[50], (unsigned long long)ns);
            atomicAdd((unsigned long long *)&d.timing[51], T == 0x7fffffff ? 1ull : 0ull);
            atomicAdd((unsigned long long *)&d.timing[52], (unsigned long long)(T == 0x7fffffff ? 0 : T));
        }
#endif
        if (!done) {
            // ---- the full evaluation: lanes run along x (contiguous 12-byte records); metrics land in LDS at their
            // arrival index (tx outer, ty inner).  Records are fetched six batches at a time with clamped coordinates
            // (no control flow around the loads), then masked; the next six are in flight while six are evaluated.
            WAVE_LDS_SYNC();
            auto wide_load = [&](int iy0, FeatRec (&fr)[WCH]) {
#pragma unroll
                for (int q = 0; q < WCH; q++) {
                    int ry = iclamp(sy - R + iy0 + q * RPB + r0, 0, H - 1);
                    const uint32_t *r = (const uint32_t *)((const char *)F0 + (colo + __umul24((uint32_t)ry, rowb)));
                    fr[q].a = r[0];
                    fr[q].b = r[1];
                    fr[q].c = r[2];
                }
            };
            auto wide_eval = [&](int iy0, const FeatRec (&fr)[WCH]) {
#pragma unroll
                for (int q = 0; q < WCH; q++) {
                    // straight-line code: the validity test is a mask, rows beyond the window land in a spare slot
                    // (the compiler otherwise wraps every candidate in two branches)
                    const int iy = iy0 + q * RPB + r0;
                    const int ry = sy - R + iy;
                    const int m = (int)__umul24((uint32_t)(wx + iabs(iy - R)), (uint32_t)feat_dist_w(fr[q].a, fr[q].b, fr[q].c, sp));
                    const int bad = (xok && ry >= 0 && ry < H) ? 0 : -1;
                    wide_m[iy < n ? ix * n + iy : ME_WIDE_LDS - 1] = m | bad;
                }
            };
            {
                constexpr int STEP = RPB * WCH;
                FeatRec fa[WCH], fb[WCH];
                wide_load(0, fa);
                for (int iy0 = 0; iy0 < n; iy0 += 2 * STEP) {
                    if (iy0 + STEP < n) wide_load(iy0 + STEP, fb);
                    wide_eval(iy0, fa);
                    if (iy0 + STEP < n) {
                        if (iy0 + 2 * STEP < n) wide_load(iy0 + 2 * STEP, fa);
                        wide_eval(iy0 + STEP, fb);
                    }
                }
            }
            {  // column n - 1
                const int iy = min(lane, n - 1);
                const int rxl = sx + R, ry = sy - R + iy;
                FeatRec f = feat0_load(F0, W, H, ry, rxl);
                int m = (R + 4 + iabs(iy - R)) * feat_dist_w(f.a, f.b, f.c, sp);
                if (!(rxl < W && ry >= 0 && ry < H)) m = -1;
                if (lane < n) wide_m[(n - 1) * n + iy] = m;
            }
            __syncthreads();
            int v[ME_SEL_NB];
#pragma unroll
            for (int u = 0; u < ME_SEL_NB; u++) {
                v[u] = -1;
                if (u < wb) {
                    int c = u * 64 + lane;
                    if (c < n * n) v[u] = wide_m[c];
                } else if (u - wb < LB) {
                    v[u] = vloc[u - wb < LB ? u - wb : 0];
                }
            }
            WAVE_LDS_SYNC();  // the selection reuses sel_lds
            auto raw = [&](int u) { return u * 64 + lane; };  // arrival index
            PP_MARK(2)
            select_topk<ME_SEL_NB>(v, 33, lane, sel_lds, L, raw, fin);
        }
        PP_MARK(3)
    } else {
        for (int base = 0; base < n * n && !FER_DBGF(d, 1); base += 64) {
            int c = base + lane;
            int tx = c / n - R, ty = c % n - R;
            int rx = sx + tx, ry = sy + ty;
            bool ok = c < n * n && rx >= 0 && rx < W && ry >= 0 && ry < H;
            int m = 0;
            if (ok) m = (iabs(tx) + iabs(ty) + 4) * feat_dist_rec(F0 + ((size_t)ry * W + rx) * 6, sp);
            wl_insert(L, 33, lane, ok, m, pack_xy(tx * 4, ty * 4));
        }
        for (int base = 0; base < nloc && !FER_DBGF(d, 2); base += 64) {
            int c = base + lane;
            int frac = c & 15, pos = c >> 4;
            int tx = pos / n2w - r2, ty = pos % n2w - r2;
            int rx = sx + tx, ry = sy + ty;
            bool ok = c < nloc && rx >= 0 && rx < W && ry >= 0 && ry < H;
            int m = 0;
            if (ok) {
                const FeatRec f = feat_rec_direct(ip, W, H, frac, ry, rx);
                m = (iabs(tx) + iabs(ty) + 4) * feat_dist_w(f.a, f.b, f.c, sp);
            }
            wl_insert(L, 33, lane, ok, m, pack_xy(tx * 4 + (frac & 3), ty * 4 + (frac >> 2)));
        }
    }
    const int n3 = __popcll(__ballot(lane < 33 && L.m < 100000000));
    int v0 = 0;
    if (!FER_DBGF(d, 4)) {  // SADs of the survivors: list slot j lives in lane j
        const SrcBlk SB = src_block_load(Y, W, sx, sy);
        const int cx = unp_x(L.xy), cy = unp_y(L.xy);
        const int sad = sad_lane(ip, W, H, sx, sy, lane < n3 ? cx : 0, lane < n3 ? cy : 0, SB);
        if (lane < n3) {
            int *o = d.st3 + (pidx * 33 + lane) * 3;
            o[0] = cx;
            o[1] = cy;
            o[2] = sad;
        }
        // the survivor of smallest SAD: a neighbour-independent guess of the partition's final vector (k_me_spec)
        const int wk = wave_min(lane < n3 ? (sad << 6) | lane : 0x7fffffff);
        if (wk != 0x7fffffff) v0 = lane_bcast(L.xy, wk & 63);
    }
    if (lane == 0) {
        d.st3n[pidx] = n3;
        d.v0[pidx] = v0;
    }

}

// ------------------------------------------------------------------ bucket walk (stage-2 candidates)
// The bucket walk of F/moestimation.cpp:470-496.  For j = 0, 1, ... the buckets su[0]-j and su[0]+j are scanned in
// (tx, ty) order; a candidate passes when it is inside the 280-diamond and its two half sums are within 100; the
// walk stops after the j that takes the count past 128 (bucket su[0] is visited twice at j = 0, like the
// reference).  Buckets are entered through the column-tile index, so only the slice whose tx can pass is read (each
// record is one 12-byte load; the exact filter decides, so the candidate set and its order are the reference's).
//
// Control flow is kept off the scalar unit (one per CU: a state machine that produced the next 64-record batch with
// ~30 scalar instructions made this kernel scalar-issue bound).  The slices of 16 steps j (32 slices) are cut into
// batches by the LANES: slice k lives in lane k, an exclusive prefix of the batch counts places its batches in a
// table in LDS, and lane b then holds the descriptor of batch b: first record, record count, bucket, "last batch of
// its step" (where the stop test falls).  The batch loop reads descriptors with v_readlane, two batches in flight.
// Groups of steps with more than 64 batches (flat content: thousands of positions per sum) go slice by slice.
// sink(ok, rank, rel, D) is called for every batch by all lanes: ok = the lane holds a candidate, rank = its
// arrival index, rel = (tx - sx) << 16 | (ty - sy) & 0xffff, D = its feature distance, info = the batch descriptor
// (step and side, see below); it returns true to end the
// walk at once (wave-uniform).  Returns the count; jend = the step j the walk ended in.
// tbl = 128 dwords of LDS private to the calling wavefront.
// max_slice: slices with more records than this are not read (the caller bounds them otherwise) -- those of the steps
// j > 0, and those of step 0 as well when skip0 is set.
// probe(bucket) may return a number of candidates the slice of that bucket is KNOWN to hold (0 = unknown): when that
// alone takes the count past halt_cnt the slice is not read (only walks that ask "is this partition crowded" pass one).
struct NoProbe {
    __device__ __forceinline__ int operator()(int) const { return 0; }
};
template <bool QUIRK, class SINK, class PROBE = NoProbe>
__device__ __forceinline__ int walk_buckets_q(const FerDev &d, int s, const int (&su)[5], const SuPk &sp, int sx, int sy, int lane,
                                              uint32_t *tbl, int halt_cnt, int &jend, SINK sink, uint32_t max_slice = 0xffffffffu,
                                              bool skip0 = false, PROBE probe = PROBE())
{
    int tren = 0;
    jend = 180;  // last step whose buckets belong to the candidate set
    const int kt = d.kt;
    const uint32_t *kol2 = d.kol2 + (size_t)s * 16384 * kt;
    const uint32_t *srec = d.sort_rec;  // indexed by the device-wide positions kol2 holds
    // QUIRK: a stream whose reference picture has n0 > 0 positions of sum 0 carries the reference's mis-filed bucket
    // layout (k_sort_quirk): whole buckets are scanned by its rules -- bucket 0 = [0, 2 n0), bucket 1 from 2 n0, the
    // others n0 places early, the last one up to the end of the array --, and a record may sit in a bucket it does not
    // belong to (or be left over from the previous picture), so its distance is taken from the features at its
    // position, as the reference does, not from the record.
    const int n0 = QUIRK ? d.zero_cnt[s] : 0;
    const uint32_t npos = (uint32_t)(d.W * d.H), g0 = (uint32_t)s * npos;
    const uint16_t *F0 = d.feat0 + (size_t)s * 6 * d.ysz;
    const int t_lo = QUIRK ? 0 : (max(sx - 279, 0) >> d.ktw_shift), t_hi = QUIRK ? kt - 1 : (min(sx + 279, d.W - 1) >> d.ktw_shift);
    auto qstart = [&](int a) -> uint32_t {  // first place of bucket a in the mis-filed layout
        if (a <= 0) return g0;
        if (a == 1) return g0 + min(2u * (uint32_t)n0, npos);
        if (a >= 16384) return g0 + npos;
        return kol2[(size_t)a * kt] - (uint32_t)n0;
    };
    const uint32_t sxy = ((uint32_t)sx << 16) | (uint32_t)sy;  // the records' (tx << 16) | ty pairing
    // Descriptor of a batch, one per lane: first record; count (bits 0-6) | last batch of its step (bit 7) | step j
    // (bits 8-15) | bucket (bits 16-30) | side (bit 31: 0 = su[0] - j, 1 = su[0] + j).  Lanes beyond the last batch hold an empty batch at a readable address, so
    // the loop can fetch two batches ahead without tests.
    const char *srec_s = (const char *)(srec + (size_t)g0 * 3);  // the stream's records: a uniform base + 32-bit byte offsets
    auto fetch = [&](unsigned b_start, int b_cnt, uint32_t &r0, uint32_t &r1, uint32_t &r2) {
        const uint32_t li = (b_start - g0) + (unsigned)min(lane, max(b_cnt - 1, 0));
        const uint32_t *e = (const uint32_t *)(srec_s + __umul24(li, 12u));  // (a picture has fewer than 2^24 positions: ferhip_create)
        r0 = e[0];
        r1 = e[1];
        r2 = e[2];
    };
    auto filter = [&](uint32_t info, uint32_t r0, uint32_t r1, uint32_t r2) -> bool {
        // |tx - sx| + |ty - sy| < 280 and both half sums within 100, on u16 pairs
        const int b_cnt = (int)(info & 127u);
        uint32_t dist = __builtin_amdgcn_sad_u16(r0, sxy, 0);
        uint32_t e12 = pk_abs16(pk_sub16(r1, sp.s12));
        bool ok = lane < b_cnt && dist < 280u && (pk_sub16(e12, 0x00640064u) & 0x80008000u) == 0x80008000u;
        unsigned long long mk = __ballot(ok);
        int rank = tren + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
        uint32_t D;
        if (!QUIRK) {
            // feature distance from the sorted payload (kar0 == a): |s0-a| + sum |si-qi| + sum |(s0-si) - (a-qi)|;
            // |s0 - a| is the step j
            const uint32_t aa = ((info >> 16) & 0x7fffu) * 0x10001u;
            D = __builtin_amdgcn_sad_u16(r1, sp.s12, (info >> 8) & 255u);
            D = __builtin_amdgcn_sad_u16(r2, sp.s34, D);
            D = __builtin_amdgcn_sad_u16(pk_sub16(aa, r1), sp.e12, D);
            D = __builtin_amdgcn_sad_u16(pk_sub16(aa, r2), sp.e34, D);
        } else {
            int px = min((int)(r0 >> 16), d.W - 1), py = min((int)(r0 & 0xffff), d.H - 1);
            D = (uint32_t)feat_dist_rec(F0 + ((size_t)py * d.W + px) * 6, sp);
        }
        const bool halt = sink(ok, rank, (int)pk_sub16(r0, sxy), (int)D, info);
        tren += __popcll(mk);
        return halt || tren > halt_cnt;  // the sink has what it wanted, or the caller only asked whether the count passes halt_cnt
    };
    // runs the nb <= 62 batches whose descriptors sit one per lane (lanes >= nb: empty batches); true = the walk is over
    auto run_table = [&](uint32_t dstart, uint32_t dinfo, int nb) -> bool {
        auto desc = [&](int b, unsigned &bs, uint32_t &bi) {
            const int bb = min(b, 63);
            bs = (unsigned)__builtin_amdgcn_readlane((int)dstart, bb);
            bi = (uint32_t)__builtin_amdgcn_readlane((int)dinfo, bb);
        };
        // four batches in flight (the walk draws 70 % of the HBM bandwidth a copy reaches: two in flight left the wavefront
        // waiting for every other batch), no register copies between them
        unsigned sA, sB, sC, sD;
        uint32_t iA, iB, iC, iD;
        uint32_t A0, A1, A2, B0, B1, B2, C0, C1, C2, D0, D1, D2;
        desc(0, sA, iA);
        fetch(sA, (int)(iA & 127u), A0, A1, A2);
        desc(1, sB, iB);
        fetch(sB, (int)(iB & 127u), B0, B1, B2);
        desc(2, sC, iC);
        fetch(sC, (int)(iC & 127u), C0, C1, C2);
        desc(3, sD, iD);
        fetch(sD, (int)(iD & 127u), D0, D1, D2);
#define WALK_STEP(SS, II, R0, R1, R2, NEXT)                                  \
        if (filter(II, R0, R1, R2)) { /* the sink has what it wanted */      \
            jend = (int)((II >> 8) & 255u);                                  \
            return true;                                                     \
        }                                                                    \
        if (II & 128u) {                                                     \
            if (tren > 128) {                                                \
                jend = (int)((II >> 8) & 255u);                              \
                return true;                                                 \
            }                                                                \
        }                                                                    \
        desc(NEXT, SS, II);                                                  \
        fetch(SS, (int)(II & 127u), R0, R1, R2);
        for (int b = 0; b < nb; b += 4) {
            WALK_STEP(sA, iA, A0, A1, A2, b + 4)
            if (b + 1 >= nb) break;
            WALK_STEP(sB, iB, B0, B1, B2, b + 5)
            if (b + 2 >= nb) break;
            WALK_STEP(sC, iC, C0, C1, C2, b + 6)
            if (b + 3 >= nb) break;
            WALK_STEP(sD, iD, D0, D1, D2, b + 7)
        }
#undef WALK_STEP
        return false;
    };
    for (int j0 = 0; j0 <= 180; j0 += 16) {
        // bucket bounds of 16 steps in ONE gather: lane = which * 16 + step, which = 0 / 1 first and end of the low
        // side's slice, 2 / 3 of the high side's
        uint32_t kb = 0u;
        {
            const int which = lane >> 4, jj = j0 + (lane & 15);
            const int a = (which & 2) ? su[0] + jj : su[0] - jj;
            const bool va = a >= 0 && a < 16384 && jj <= 180;
            if (!QUIRK) {
                if (va) kb = kol2[(size_t)a * kt + ((which & 1) ? t_hi + 1 : t_lo)];
            } else {
                if (va) kb = qstart(a + (which & 1));
            }
        }
        // slice k = step * 2 + side in lane k < 32
        const int step = (lane >> 1) & 15, side = lane & 1;
        const uint32_t st = (uint32_t)__shfl((int)kb, step + (side ? 32 : 0)), en = (uint32_t)__shfl((int)kb, step + (side ? 48 : 16));
        uint32_t cnt = (lane < 32 && en > st) ? en - st : 0u;
        if (cnt > max_slice && (j0 + step > 0 || skip0)) cnt = 0u;
        const int nb = (int)((cnt + 63u) >> 6);
        const int a_k = side ? su[0] + j0 + step : su[0] - (j0 + step);
        const uint32_t ja = ((uint32_t)(j0 + step) << 8) | ((uint32_t)(a_k & 0x7fff) << 16) | ((uint32_t)side << 31);
        int pre = nb;  // inclusive prefix over the lanes
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
            int v = __shfl_up(pre, o);
            if (lane >= o) pre += v;
        }
        const int TB = __builtin_amdgcn_readlane(pre, 31);
        if (TB == 0) continue;
        if (TB <= 62) {
            // the step's last batch: the high side's last one, or the low side's when the high side is empty
            const int nb_hi = __shfl(nb, lane | 1);
            const bool closes = side == 1 || nb_hi == 0;
            tbl[lane] = g0;  // empty batches behind the last one
            tbl[64 + lane] = 0u;
            WAVE_LDS_SYNC();
            for (int i = 0; __any(i < nb); i++) {
                if (i < nb) {
                    const int b = pre - nb + i;
                    const uint32_t c = min(cnt - 64u * (uint32_t)i, 64u);
                    tbl[b] = st + 64u * (uint32_t)i;
                    tbl[64 + b] = c | ((closes && i == nb - 1) ? 128u : 0u) | ja;
                }
            }
            WAVE_LDS_SYNC();
            const uint32_t dstart = tbl[lane], dinfo = tbl[64 + lane];
            WAVE_LDS_SYNC();
            if (run_table(dstart, dinfo, TB)) return tren;
        } else {
            // a crowded group: slice by slice, 62 batches of a slice at a time
            for (int k = 0; k < 32; k++) {
                const uint32_t ks = (uint32_t)__builtin_amdgcn_readlane((int)st, k), kc = (uint32_t)__builtin_amdgcn_readlane((int)cnt, k);
                const uint32_t kja = (uint32_t)__builtin_amdgcn_readlane((int)ja, k);
                const int knb = (int)((kc + 63u) >> 6);
                if (kc > (uint32_t)FER_BIG_SLICE) {
                    const int known = probe((int)((kja >> 16) & 0x7fffu));
                    if (known > 0 && tren + known > halt_cnt) {
                        tren += known;
                        jend = j0 + (k >> 1);
                        return tren;
                    }
                }
                for (int b0 = 0; b0 < knb; b0 += 62) {
                    const int m = min(62, knb - b0);
                    const uint32_t off = 64u * (uint32_t)(b0 + lane);
                    const bool on = lane < m;
                    if (run_table(on ? ks + off : g0, on ? (min(kc - off, 64u) | kja) : 0u, m)) return tren;  // stop tests fall between steps only
                }
                if ((k & 1) && tren > 128) {
                    jend = j0 + (k >> 1);
                    return tren;
                }
            }
        }
    }
    return tren;
}

template <class SINK, class PROBE = NoProbe>
__device__ __forceinline__ int walk_buckets(const FerDev &d, int s, const int (&su)[5], const SuPk &sp, int sx, int sy, int lane,
                                            uint32_t *tbl, int halt_cnt, int &jend, SINK sink, PROBE probe = PROBE())
{
    jend = 0;
    if (d.basic || FER_DBGF(d, 8)) return 0;
    if (d.zero_cnt[s] > 0) return walk_buckets_q<true>(d, s, su, sp, sx, sy, lane, tbl, 0x7fffffff, jend, sink);  // (the exact slow path needs every candidate)
    return walk_buckets_q<false>(d, s, su, sp, sx, sy, lane, tbl, halt_cnt, jend, sink, 0xffffffffu, false, probe);
}

// ------------------------------------------------------------------ k_me_walk
// Stage-2 candidate set of every 8x8 partition (the predictor weight is applied later, in k_me_resolve): the first
// FER_ST2_CAP candidates and their count.  A partition with more candidates than that -- large flat areas, where
// thousands of positions share one feature vector -- is walked again by k_me_resolve, which then knows the predictor
// and keeps an exact running top-33.  Kept apart from k_me_pre because it is a stream of 12-byte records that
// wants many resident wavefronts and few registers.
__global__ __launch_bounds__(64, 8) void k_me_walk(FerDev d)
{
    const int lane = threadIdx.x;
    const int s = blockIdx.y;
    if (d.hdr[s * 4 + 3] != 0) return;
    // plain picture order: neighbouring partitions enter the bucket index at the same column tiles and at nearby sums,
    // so its lines are shared (ordering the partitions by sum, or giving each XCD a band of the picture, lost 20 %)
    const int mb = blockIdx.x >> 2, part = blockIdx.x & 3;
    const int W = d.W;
    const uint8_t *Y = d.curY + (size_t)s * d.ysz;
    const int sx = ((mb % d.mbw) << 4) + (part & 1) * 8, sy = ((mb / d.mbw) << 4) + (part >> 1) * 8;
    const size_t pidx = ((size_t)s * d.nmb + mb) * 4 + part;
    // box sums of the source block, F/moestimation.cpp:440-451 (k_me_pre derives and stores the same)
    int px = lane & 7, py = lane >> 3;
    int v = Y[(size_t)(sy + py) * W + sx + px];
    int su[5];
    su[0] = wave_sum(v);
    su[1] = wave_sum(py > 3 ? 0 : v);
    su[2] = wave_sum(px > 3 ? 0 : v);
    su[3] = wave_sum((py & 3) > 1 ? 0 : v);
    su[4] = wave_sum((px & 3) > 1 ? 0 : v);
    const SuPk sp = su_pack(su);
    int2 *out = (int2 *)(d.st2 + pidx * FER_ST2_CAP * 2);
    __shared__ uint32_t tbl[128];
    int jend;
    // more than FER_ST2_CAP candidates make the partition "crowded" (below): the count itself is not needed then, and
    // the step the stop test would fire in is the current one (the count is already past 128)
    // A partition inside a flat area would read tens of thousands of records of the area's bucket before it has seen the
    // 385 candidates that make it crowded.  Instead the 25 x 25 positions around the block are looked up in the feature
    // map: those of the bucket's modal class (FerDev.bmodal) that pass the walk's tests ARE candidates of that slice.
    const uint16_t *F0p = d.feat0 + (size_t)s * 6 * d.ysz;
    auto flat_probe = [&](int a) -> int {
        if (d.zero_cnt[s] != 0) return 0;
        const uint32_t *bmq = d.bmodal + ((size_t)s * 16384 + a) * 4;
        const uint32_t m1 = bmq[0], m2 = bmq[1];
        if (bmq[2] > FER_OUTL) return 0;  // no modal class
        const uint32_t e12 = pk_abs16(pk_sub16(m1, sp.s12));
        if ((pk_sub16(e12, 0x00640064u) & 0x80008000u) != 0x80008000u) return 0;  // the class fails the half-sum test
        int n = 0;
        for (int k0 = 0; k0 < 625; k0 += 64) {
            const int k = k0 + lane;
            const int px = sx + k % 25 - 12, py = sy + k / 25 - 12;
            const bool in = k < 625 && px >= 0 && px < d.W && py >= 0 && py < d.H;
            const uint32_t *rec = (const uint32_t *)(F0p + ((size_t)iclamp(py, 0, d.H - 1) * d.W + iclamp(px, 0, d.W - 1)) * 6);
            const uint32_t ra = rec[0], rb = rec[1], rc = rec[2];
            n += __popcll(__ballot(in && (int)(ra & 0xffffu) == a && ((ra >> 16) | (rb << 16)) == m1 && ((rb >> 16) | (rc << 16)) == m2));
        }
        return n;
    };
    const int tren = walk_buckets(d, s, su, sp, sx, sy, lane, tbl, FER_ST2_CAP, jend, [&](bool ok, int rank, int rel, int D, uint32_t) {
        if (ok && rank < FER_ST2_CAP) out[rank] = make_int2(rel, D);
        return false;
    }, flat_probe);
    if (tren > FER_ST2_CAP && d.zero_cnt[s] == 0) {
        // A crowded partition (flat areas: thousands of positions share a feature vector).  k_me_resolve will not go
        // through the candidates again; it looks for the winners around the predictor (resolve_crowded), for which it
        // needs the last step J of the walk, the smallest positive distance among the candidates, and the candidates
        // of distance 0 -- their metric is 0 whatever the predictor, so the first 33 of them in arrival order lead the
        // list.  A second walk with another sink collects that; the list of the first 384 is not used then.
        // A lower bound of the feature distance over the whole candidate set comes from the ranges the other four sums
        // take in each bucket (FerDev.brange): |s - k| >= the distance of s from the range of k.  It also tells whether
        // a candidate of distance 0 can exist at all (only in bucket su[0], and only if every range contains its sum).
        // The smallest positive distance of the whole candidate set (what the ring scan of resolve_crowded stops on) is
        // put together from two sides.  Slices of more than FER_BIG_SLICE records (the flat area itself) are not read
        // again: their candidates are bounded from below by 5 |s_0 - a| -- |s_k - q_k| + |(s_0 - s_k) - (a - q_k)| >=
        // |s_0 - a| for each of the four other sums -- and by the ranges those sums take in the bucket.  All other
        // slices of the steps 0 .. J are small and are read: their exact smallest distance.  The same pass collects the
        // candidates of distance 0 (bucket su[0] only; read whatever its size when its ranges admit distance 0).
        int zc = 0, dmin = 0x7fffffff;
        bool skip0 = false;  // bucket su[0] is big and its ranges exclude distance 0: bounded, not read
        // the big slices (at most FER_BIGS of them are described; more = the general bound only)
        int nbig = 0, bigA[FER_BIGS], bigDj[FER_BIGS];
        bool fallback = false;
#pragma unroll
        for (int q = 0; q < FER_BIGS; q++) bigA[q] = bigDj[q] = 0;
        {
            const uint32_t *brs = d.brange + (size_t)s * 16384 * 8;
            const uint32_t *kol2p = d.kol2 + (size_t)s * 16384 * d.kt;
            const int bt_lo = max(sx - 279, 0) >> d.ktw_shift, bt_hi = min(sx + 279, d.W - 1) >> d.ktw_shift;
            for (int b0 = -jend; b0 <= jend; b0 += 64) {
                const int dj = b0 + lane;
                const int a = su[0] + dj;
                int lb = 0x7fffffff;
                bool big = false;
                if (dj <= jend && a >= 0 && a < 16384) {
                    // what the walk reads of the bucket: its positions in the column tiles the 280-diamond reaches
                    const uint32_t sz = kol2p[(size_t)a * d.kt + bt_hi + 1] - kol2p[(size_t)a * d.kt + bt_lo];
                    big = sz > (uint32_t)FER_BIG_SLICE;
                    if (sz > (dj == 0 ? 0u : (uint32_t)FER_BIG_SLICE)) {
                        const uint4 hi = *(const uint4 *)(brs + (size_t)a * 8), lo = *(const uint4 *)(brs + (size_t)a * 8 + 4);
                        const uint32_t h[4] = {hi.x, hi.y, hi.z, hi.w}, l[4] = {lo.x, lo.y, lo.z, lo.w};
                        lb = iabs(dj);
                        if (l[0] != 0) {  // ... and ranges (a bucket of more than FER_BRANGE_MIN positions)
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                const int kmin = 65535 - (int)l[q], kmax = (int)h[q];
                                const int s1 = su[q + 1], s2 = su[q + 1] + dj;  // |s_k - k_k| and |(s_0 - s_k) - (a - k_k)| = |k_k - (s_k + dj)|
                                lb += max(max(kmin - s1, s1 - kmax), 0) + max(max(kmin - s2, s2 - kmax), 0);
                            }
                        }
                        lb = max(lb, 5 * iabs(dj));
                        if (dj == 0) {  // bucket su[0] is read below (all of it) unless it is big and cannot hold distance 0
                            if (big && lb > 0) {
                                skip0 = true;
                            } else {
                                lb = 0x7fffffff;
                                big = false;
                            }
                        }
                    }
                }
                dmin = min(dmin, lb);
                unsigned long long bm = __ballot(big);
                while (bm) {
                    const int src = __ffsll((long long)bm) - 1;
                    bm &= bm - 1;
                    if (nbig < FER_BIGS) {
#pragma unroll
                        for (int q = 0; q < FER_BIGS; q++)
                            if (q == nbig) {
                                bigA[q] = lane_bcast(a, src);
                                bigDj[q] = lane_bcast(dj, src);
                            }
                        nbig++;
                    } else {
                        fallback = true;
                    }
                }
            }
            dmin = wave_min(dmin);
            skip0 = __any(skip0);
        }
        // candidates listed for resolve_crowded (slots 64 .. 383 of the partition's list): everything of the small slices,
        // and the outliers of the big ones
        int np2 = 0;
        const uint32_t g0_ = (uint32_t)s * (uint32_t)(d.W * d.H);
        auto p2_add = [&](bool take, int rel, uint32_t D, int j, int side) {
            const unsigned long long mt = __ballot(take);
            const int pos = np2 + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mt >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mt, 0u));
            if (take && pos < FER_P2_CAP) out[64 + pos] = make_int2(rel, (int)(D | ((uint32_t)j << 20) | ((uint32_t)side << 28)));
            np2 += __popcll(mt);
        };
        // The big slices: the modal class of the bucket (FerDev.bmodal) has ONE feature distance, which bounds it exactly;
        // its outliers are candidates of their own.
        int dminN = 0x7fffffff;
        int bigD[FER_BIGS];
#pragma unroll
        for (int q = 0; q < FER_BIGS; q++) bigD[q] = 0x7fffffff;
        const uint32_t sxy_ = ((uint32_t)sx << 16) | (uint32_t)sy;
#pragma unroll
        for (int q = 0; q < FER_BIGS; q++) {
            if (q < nbig && !fallback) {
                const uint32_t *bmq = d.bmodal + ((size_t)s * 16384 + bigA[q]) * 4;
                const uint32_t m1 = bmq[0], m2 = bmq[1], nout = bmq[2];
                const int jq = iabs(bigDj[q]), sideq = bigDj[q] > 0 ? 1 : 0;
                const uint32_t aa = (uint32_t)bigA[q] * 0x10001u;
                auto dist_of = [&](uint32_t r1, uint32_t r2) {
                    uint32_t D = __builtin_amdgcn_sad_u16(r1, sp.s12, (uint32_t)jq);
                    D = __builtin_amdgcn_sad_u16(r2, sp.s34, D);
                    D = __builtin_amdgcn_sad_u16(pk_sub16(aa, r1), sp.e12, D);
                    return __builtin_amdgcn_sad_u16(pk_sub16(aa, r2), sp.e34, D);
                };
                auto halves_ok = [&](uint32_t r1) {
                    const uint32_t e12 = pk_abs16(pk_sub16(r1, sp.s12));
                    return (pk_sub16(e12, 0x00640064u) & 0x80008000u) == 0x80008000u;
                };
                if (nout > FER_OUTL) {
                    fallback = true;
                } else {
                    const uint32_t DM = dist_of(m1, m2);
                    const bool members = halves_ok(m1);  // the half-sum test of the walk is the same for the whole class
                    bigD[q] = members && DM > 0 ? (int)DM : 0x7fffffff;
                    dminN = min(dminN, bigD[q]);
                    // the outliers, 64 to a batch
                    const uint32_t *ol = d.boutl + ((size_t)s * d.nlists + (d.kol2[((size_t)s * 16384 + bigA[q]) * d.kt] - g0_) / FER_BRANGE_MIN) * FER_OUTL;
                    for (uint32_t o0 = 0; o0 < nout; o0 += 64) {
                        const bool lv = o0 + (uint32_t)lane < nout;
                        const uint32_t idx = lv ? ol[o0 + lane] : g0_;
                        const uint32_t *rec = d.sort_rec + (size_t)idx * 3;
                        const uint32_t r0 = rec[0], r1 = rec[1], r2 = rec[2];
                        const uint32_t D = dist_of(r1, r2);
                        const bool take = lv && __builtin_amdgcn_sad_u16(r0, sxy_, 0u) < 280u && halves_ok(r1) && D > 0 && D != DM;
                        p2_add(take, (int)pk_sub16(r0, sxy_), D, jq, sideq);
                    }
                    // (an outlier at distance 0 is found by the read of bucket su[0] below: its ranges admit 0 then)
                }
            }
        }
        {
            // one more walk over the steps 0 .. J, without the big slices: candidates of distance 0 (step 0, first visit;
            // the second visit of bucket su[0] repeats them), every other candidate it reads -> the list, and the
            // smallest positive distance of those (the general bound)
            int j2, dpos = 0x7fffffff;
            walk_buckets_q<false>(d, s, su, sp, sx, sy, lane, tbl, 0x7fffffff, j2, [&](bool ok, int rank, int rel, int D, uint32_t info) {
                (void)rank;
                const int j = (int)((info >> 8) & 255u);
                if (j > jend) return true;
                if (j == 0 && (info >> 31)) return false;  // (the second visit of bucket su[0]: the same records)
                if (ok && D > 0) dpos = min(dpos, D);
                p2_add(ok && D > 0, rel, (uint32_t)D, j, (int)(info >> 31));
                if (j == 0) {
                    const bool z = ok && D == 0;
                    const unsigned long long mz = __ballot(z);
                    const int zr = zc + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mz >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mz, 0u));
                    if (z && zr < 33) out[zr] = make_int2(rel, 0);
                    zc += __popcll(mz);
                    if (zc >= 33) return true;  // 33 candidates of distance 0 are the whole list: nothing else is needed
                }
                return false;
            }, (uint32_t)FER_BIG_SLICE, skip0);
            dmin = max(min(dmin, wave_min(dpos)), 1);
            if (zc < 33 && zc > 0) {  // the repeats of the second visit
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");  // the list is read back
                // (agent-scope load: served by L2, where the stores above have landed)
                const unsigned long long ev = __hip_atomic_load((const unsigned long long *)&out[min(lane, 32)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int2 e = make_int2((int)(unsigned)ev, (int)(unsigned)(ev >> 32));
                if (lane < zc && zc + lane < 33) out[zc + lane] = e;
                zc = min(2 * zc, 33);
            }
        }
        if (np2 > FER_P2_CAP) fallback = true;
        if (lane == 0) {
            out[40] = make_int2(jend, dmin);
            out[41] = make_int2(min(zc, 33), (int)((uint32_t)min(np2, FER_P2_CAP) | ((uint32_t)nbig << 16) | (fallback ? 0x40000000u : 0u)));
#pragma unroll
            for (int q = 0; q < FER_BIGS; q++) out[42 + q] = make_int2(bigA[q], bigD[q]);
            out[46] = make_int2(dminN, 0);
        }
    }
    if (lane == 0) d.st2n[pidx] = tren;
}

// SAD of the K list entries (slot j in lane j) and the lane's key (cost << 6 | list index):
// cost = SAD + |mv - mvp| (F/moestimation.cpp:460-468).  No wave-level reduction in here.
template <int K>
__device__ __forceinline__ void sad_keys(const WList &L, int cnt, int lane, const IPlanes &ip, int W, int H, int sx, int sy,
                                         const SrcBlk &SB, int mvpx, int mvpy, int &best, int &bestxy)
{
    const bool on = lane < cnt && lane < K;
    const int cxv = on ? unp_x(L.xy) : 0, cyv = on ? unp_y(L.xy) : 0;
    const int sad = sad_lane(ip, W, H, sx, sy, cxv, cyv, SB);
    best = on ? ((sad + iabs(cxv - mvpx) + iabs(cyv - mvpy)) << 6) | lane : 0x7fffffff;
    bestxy = L.xy;
}
// the strict-< first-minimum update of the reference, in list order, from per-lane bests
__device__ __forceinline__ void take_best(int best, int bestxy, int &bmin, int &bx, int &by)
{
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
// The neighbour-dependent part of the search is a chain: a partition needs the final vectors
// of its left, up, up-right and up-left neighbours.  It runs as ONE persistent launch per
// picture: one workgroup per (stream, row of 8x8 partitions) walks its row left to right and
// follows the row above at the distance the prediction needs (3 partitions for partition 0,
// whose P_Skip predictor reads the up-right MACROBLOCK; 2 for partitions 1 and 2; 1 for
// partition 3), so streams and rows advance independently instead of meeting at a device-wide
// barrier (a kernel boundary) 639 times per picture.  Rows are handed out by an atomic ticket in
// row-major order: the row a workgroup waits on was always claimed earlier by a workgroup that
// is running or finished, so every wait terminates; a bounded spin count turns anything
// unexpected into an error flag instead of a hang.
//
// The searches that depend on the predictor were run beforehand for a GUESSED predictor (k_me_spec): the common step
// of the chain is "compute the true predictor, find that its integer part is the guessed one, price the stored
// candidate lists" -- a few hundred instructions.  After a wrong guess the chain searches itself (stage 1, then stages
// 2 and 3, in the same wavefront).  One wavefront per row: with thousands of rows in flight the launch is bound by the
// instructions it issues, not by the latency of a row.
// Everything after the vector of the partition is known (merge, mvd, final prediction, snapping, and the
// reconstruction of P_Skip macroblocks) is not on any other partition's dependency chain and lives in k_p_resid
// (fer_resid.hip).
#define ST1_UNROLL 7
#define RES_SPIN_LIMIT (1 << 23)

struct ResPre {  // predictor-independent operands of one partition (res_prefetch: role 1 = all of them, role 0 = the last two lines)
    int n2, n3, n2raw;
    int2 e2[FER_ST2_CAP / 64];  // a crowded partition: e2[0] = (J, Dmin), e2[1].x = distance-0 candidates listed
    int2 z;                     // ... and the lane's distance-0 candidate
    int c3x, c3y, c3s;
    int su[5];
    SrcBlk sb;  // source block (wave-uniform)
};

__device__ __forceinline__ void res_prefetch(const FerDev &d, int s, int gx, int gy, int lane, int role, ResPre &p)
{
    const int mb = (gy >> 1) * d.mbw + (gx >> 1), part = (gy & 1) * 2 + (gx & 1);
    const size_t pidx = ((size_t)s * d.nmb + mb) * 4 + part;
    p.n2 = p.n3 = p.n2raw = 0;
    p.z = make_int2(0, 0);
    p.c3x = p.c3y = p.c3s = 0;
#pragma unroll
    for (int u = 0; u < FER_ST2_CAP / 64; u++) p.e2[u] = make_int2(0, 0);
    if (role == 1 && !d.basic) {
        p.n2raw = d.st2n[pidx];
        p.n2 = min(p.n2raw, FER_ST2_CAP);
        p.n3 = FER_DBGF(d, 64) ? 0 : d.st3n[pidx];
        const int2 *c2 = (const int2 *)(d.st2 + pidx * FER_ST2_CAP * 2);
#pragma unroll
        for (int u = 0; u < FER_ST2_CAP / 64; u++) p.e2[u] = c2[u * 64 + lane];  // slots >= n2 hold stale data, masked later
        p.z = p.e2[0];  // (a crowded partition: slots 0 .. 32 = its candidates of distance 0, 40 .. 46 = its summary, 64 .. = its list)
        const int *c3 = d.st3 + pidx * 33 * 3;
        const int l3 = min(lane, 32);
        p.c3x = c3[l3 * 3];
        p.c3y = c3[l3 * 3 + 1];
        p.c3s = c3[l3 * 3 + 2];
    }
#pragma unroll
    for (int k = 0; k < 5; k++) p.su[k] = d.suma[pidx * 5 + k];
    p.sb = src_block_load(d.curY + (size_t)s * d.ysz, d.W, gx * 8, gy * 8);
}

// Vectors travel between rows as self-validating 64-bit words in d.chain64 [S][nmb][4]:
// bits 0-31 the packed vector, bits 32-62 the serial number of the picture, bit 63 "macroblock is
// P_Skip".  A reader polls the words it needs until they carry the current serial, so no separate
// flag, fence or store ordering is involved; loads and stores are agent scope (coherent across
// the XCDs' L2s).
#define CH_SKIP 0x80000000u
struct ResNbr {      // neighbour vectors of one partition (packed), v* = available
    bool vA, vB, vC, vD, vC16;
    int A, B, C, D, C16;
};

// Availability of the neighbours of the 8x8 partition (gx, gy) in PARTITION coordinates (F/mode_pred.cpp:60-110 for 8x8
// partitions of inter macroblocks): B = (gx, gy - 1), C = (gx + 1, gy - 1), D = (gx - 1, gy - 1), and for the 16x16
// (P_Skip) predictor of partition 0 C16 = (gx + 2, gy - 1).  C does not exist for partition 3 (not yet decoded) and
// always for partition 2 (the macroblock's own quadrant 1); everything is wave-uniform.
__device__ __forceinline__ void nbr_avail(int gx, int gy, int gw, bool &vB, bool &vC, bool &vD, bool &vC16)
{
    const int part = (gy & 1) * 2 + (gx & 1);
    vB = gy > 0;
    vD = gy > 0 && gx > 0;
    vC = part == 3 ? false : (part == 2 ? true : (part == 1 ? (gy > 0 && gx + 1 < gw) : gy > 0));
    vC16 = part == 0 && gy > 0 && gx + 2 < gw;
}
// index of partition (xa, ya) in an array laid out [macroblock][quadrant]
__device__ __forceinline__ int part_slot(int mbw, int xa, int ya) { return (((ya >> 1) * mbw + (xa >> 1)) << 2) + ((ya & 1) << 1) + (xa & 1); }

__device__ __forceinline__ void nbr_to(int v, bool ok, int &mx, int &my, int &ref)
{
    mx = ok ? (int)(short)(v & 0xffff) : FER_MV_NA;
    my = ok ? (v >> 16) : FER_MV_NA;
    ref = ok ? 0 : -1;
}
// PredictMV_Luma without the directional rules (8x8 partitions and the P_Skip 16x16 predictor)
__device__ __forceinline__ void predict_nbr(bool vA, int A, bool vB, int B, bool vC, int C, bool vD, int D, int &ox, int &oy)
{
    int mx[3], my[3], ref[3];
    nbr_to(A, vA, mx[0], my[0], ref[0]);
    nbr_to(B, vB, mx[1], my[1], ref[1]);
    if (vC)
        nbr_to(C, true, mx[2], my[2], ref[2]);
    else
        nbr_to(D, vD, mx[2], my[2], ref[2]);
    predict_core(mx, my, ref, ox, oy);
}

// wave-level first minimum of per-lane (cost << 6 | list index) keys: the key and its vector
__device__ __forceinline__ void wave_best(int best, int bestxy, int &wkey, int &wxy)
{
    wkey = wave_min(best);
    wxy = 0;
    if (wkey != 0x7fffffff) wxy = lane_bcast(bestxy, __ffsll((long long)__ballot(best == wkey)) - 1);
}

// ---- P_Skip (F/mode_pred.cpp:381-402 + F/moestimation.cpp:402-425) ----
// the vector of the P_Skip candidate: the 16x16 predictor, or zero at the picture's top / left edge and next to a zero
// neighbour
__device__ __forceinline__ void pskip_vector(const FerDev &d, int mb, int mbx, const ResNbr &N, int &smx, int &smy)
{
    smx = smy = 0;
    if (!(mb < d.mbw || mbx == 0)) {
        const bool zu = N.B == 0, zl = N.A == 0;  // up MB quadrant 2, left MB quadrant 1
        if (!(zu || zl)) predict_nbr(N.vA, N.A, N.vB, N.B, N.vC16, N.C16, N.vD, N.D, smx, smy);
    }
}
// the test itself: every luma sample of the macroblock within MAXDIFF of its prediction.  A pure function of the
// pictures and the vector (the reconstruction of a P_Skip macroblock is written by k_p_resid), so the speculative
// pre-pass and both wavefronts of a chain workgroup may evaluate it.
__device__ __forceinline__ bool pskip_test(const FerDev &d, int s, int mbx, int mby, int lane, int smx, int smy)
{
    const int W = d.W, H = d.H;
    const uint8_t *Y = d.curY + (size_t)s * d.ysz;
    const uint8_t *RY = d.refY + (size_t)s * d.ysz;
    const IPlanes ip = ip_stream(d, s);
    const int xp = mbx << 4, yp = mby << 4;
    // each lane owns 4 luma samples (lx..lx+3, ly)
    const int lx = (lane & 3) * 4, ly = lane >> 2;
    int srcv[4], pred[4];
    const uint32_t sv = *(const uint32_t *)(Y + (size_t)(yp + ly) * W + xp + lx);
#pragma unroll
    for (int k = 0; k < 4; k++) srcv[k] = (sv >> (8 * k)) & 0xff;
    mc_luma4(RY, ip, W, H, xp, yp, lx, ly, smx, smy, pred);
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
    return __all(exact);
}

// ---- stage 1: +-W/16 around the integer centre (genx, geny), all 16 fractional planes (K = 17) -> the list, its length.
// WindowSize 16 / 32: the metrics are in mtab (local_metrics, whoever ran it).  Depends on the predictor through its
// integer part only.
template <int WIN>
__device__ __forceinline__ int stage1_list(const FerDev &d, int s, int gx, int gy, int lane, const ResPre &P, int genx, int geny,
                                           int *sel_lds, const int *mtab, WList &L1)
{
    const int window = WIN ? WIN : d.window;
    const int W = d.W, H = d.H;
    const int sx = gx * 8, sy = gy * 8;
    L1.m = INF_M;
    L1.xy = 0;
    if (FER_DBGF(d, 16)) return 0;  // (probe: the chain without stage 1)
    const int r1 = window / 16, n1 = 2 * r1 + 1, tot1 = n1 * n1 * 16;
    auto raw1 = [&](int u) { return u * 64 + lane; };  // arrival index
    auto fin1 = [&](int cc) {
        int frac = cc & 15, pos = cc >> 4;
        return pack_xy((genx - r1 + pos / n1) * 4 + (frac & 3), (geny - r1 + pos % n1) * 4 + (frac >> 2));
    };
    if (WIN == 32 || WIN == 16) {
        constexpr int RR = (WIN ? WIN : 32) / 16;
        constexpr int NB1 = LocalGeo<RR>::NB;
        int m[NB1];
#pragma unroll
        for (int u = 0; u < NB1; u++) m[u] = mtab[u * 64 + lane];
        select_topk<NB1>(m, 17, lane, sel_lds, L1, raw1, fin1);
    } else {
        const SuPk sp = su_pack(P.su);
        const IPlanes ip = ip_stream(d, s);
        for (int base = 0; base < tot1; base += 64) {
            int cc = base + lane;
            int frac = cc & 15, pos = cc >> 4;
            int tx = genx - r1 + pos / n1, ty = geny - r1 + pos % n1;
            int rx = sx + tx, ry = sy + ty;
            bool ok = cc < tot1 && rx >= 0 && rx < W && ry >= 0 && ry < H;
            int m = 0;
            if (ok) {
                const FeatRec f = feat_rec_direct(ip, W, H, frac, ry, rx);
                m = (iabs(tx - genx) + iabs(ty - geny) + 4) * feat_dist_w(f.a, f.b, f.c, sp);
            }
            wl_insert(L1, 17, lane, ok, m, pack_xy(tx * 4 + (frac & 3), ty * 4 + (frac >> 2)));
        }
    }
    return __popcll(__ballot(lane < 17 && L1.m < 100000000));
}

// the chain's own search (after a wrong guess), stage 1: the local features, the list, its SADs -> the best (key, vector)
template <int WIN>
__device__ __forceinline__ void resolve_stage1(const FerDev &d, int s, int gx, int gy, int lane, const ResPre &P, int mvpx, int mvpy,
                                               uint32_t *htab, int *mtab, int &wkey, int &wxy)
{
    wkey = 0x7fffffff;
    wxy = 0;
    if (FER_DBGF(d, 16)) return;
    if (WIN == 32 || WIN == 16) {
        constexpr int RR = (WIN ? WIN : 32) / 16;
        local_metrics<RR>(ip_stream(d, s), d.W, d.H, gx * 8 + (mvpx >> 2) - RR, gy * 8 + (mvpy >> 2) - RR, su_pack(P.su), lane, htab, mtab);
    }
    WList L1;
    const int cnt1 = stage1_list<WIN>(d, s, gx, gy, lane, P, mvpx >> 2, mvpy >> 2, (int *)htab, mtab, L1);  // (the row sums are dead: the selection's scratch)
    if (FER_DBGF(d, 16)) return;
    int b1, b1xy;
    sad_keys<17>(L1, cnt1, lane, ip_stream(d, s), d.W, d.H, gx * 8, gy * 8, P.sb, mvpx, mvpy, b1, b1xy);
    wave_best(b1, b1xy, wkey, wxy);
}

// ---- stage 2 of a crowded partition: the winners are looked for AROUND THE PREDICTOR ----
// The candidate set of the bucket walk is "every position p of the 280-diamond around the block whose 8x8 sum is
// within J of the block's and whose two half sums are within 100" (J = the step at which the walk stopped), a position
// of sum exactly su[0] counting twice (bucket su[0] is visited from both sides at j = 0), and the list wanted is the
// 33 smallest by (metric, arrival), metric = (|p - (block + gen)|_1 + 4) * D(p), arrival = (j, side, tx, ty).  All of
// that can be evaluated per POSITION from the plane-0 feature records, without the sorted order.  So the L1 rings
// around the predictor are scanned outwards, members inserted into the running top-33; a ring at distance r can only
// hold metrics >= (r + 4) * Dmin, Dmin = the smallest positive distance of the whole set (from k_me_walk), which ends
// the scan as soon as that exceeds the 33rd best metric found.  Candidates of distance 0 have metric 0 for any
// predictor: k_me_walk listed the first 33 of them in arrival order, they lead the list.  In a flat area the scan ends
// after a handful of rings, where re-walking the buckets visits hundreds of thousands of records per partition.
struct CList {
    int m;         // metric of slot == lane
    unsigned ak;   // arrival key j << 21 | side << 20 | (tx + 280) << 10 | (ty + 280), relative to the block
    int xy;
};
__device__ __forceinline__ void cl_insert(CList &L, int lane, bool valid, int m, unsigned ak, int xy)
{
    // candidates of this batch that beat the 33rd entry, in any order: the list is ordered by (m, ak) itself
    const int tm = lane_bcast(L.m, 32);
    const unsigned tk = (unsigned)lane_bcast((int)L.ak, 32);
    unsigned long long mask = __ballot(valid && (m < tm || (m == tm && ak < tk)));
    while (mask) {
        const int src = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        const int cm = lane_bcast(m, src), cxy = lane_bcast(xy, src);
        const unsigned ck = (unsigned)lane_bcast((int)ak, src);
        const int pos = __popcll(__ballot(lane < 33 && (L.m < cm || (L.m == cm && L.ak < ck))));
        if (pos < 33) {
            const int um = FER_DPP(L.m, DPP_WAVE_SHR1), uxy = FER_DPP(L.xy, DPP_WAVE_SHR1);
            const unsigned uk = (unsigned)FER_DPP((int)L.ak, DPP_WAVE_SHR1);
            if (lane > pos && lane < 33) {
                L.m = um;
                L.ak = uk;
                L.xy = uxy;
            } else if (lane == pos) {
                L.m = cm;
                L.ak = ck;
                L.xy = cxy;
            }
        }
    }
}

__device__ __forceinline__ void resolve_crowded(const FerDev &d, int s, int sx, int sy, int lane, const ResPre &P, int genx, int geny,
                                                WList &L2)
{
    const int W = d.W, H = d.H;
    const uint16_t *F0 = d.feat0 + (size_t)s * 6 * d.ysz;
    const SuPk sp = su_pack(P.su);
    // summary of k_me_walk (slots 40 .. 46 of the list, held by those lanes of the first batch)
    const int J = lane_bcast(P.e2[0].x, 40), zc = lane_bcast(P.e2[0].x, 41);
    const uint32_t fl = (uint32_t)lane_bcast(P.e2[0].y, 41);
    const bool classes = (fl & 0x40000000u) == 0;  // big slices described by class, everything else listed
    const int np2 = classes ? (int)(fl & 0xffffu) : 0, nbig = classes ? (int)((fl >> 16) & 15u) : 0;
    const int dmin = classes ? lane_bcast(P.e2[0].x, 46) : lane_bcast(P.e2[0].y, 40);
    int bigA[FER_BIGS], bigD[FER_BIGS];
#pragma unroll
    for (int q = 0; q < FER_BIGS; q++) {
        bigA[q] = lane_bcast(P.e2[0].x, 42 + q);
        bigD[q] = lane_bcast(P.e2[0].y, 42 + q);
    }
    CList L;
    L.m = INF_M;
    L.ak = 0xffffffffu;
    L.xy = 0;
    if (lane < zc) {  // the candidates of distance 0, already in arrival order (all of j = 0, side 0 first)
        const int rel = P.z.x;
        const int tx = rel >> 16, ty = (int)(short)(rel & 0xffff);
        L.m = 0;
        L.ak = (unsigned)lane;  // only their order matters: nothing else has metric 0
        L.xy = pack_xy(tx * 4, ty * 4);
    }
    if (zc < 33) {
#pragma unroll
        for (int u = 1; u < FER_ST2_CAP / 64; u++) {  // the listed candidates: position relative to the block, D | j << 20 | side << 28
            if ((u - 1) * 64 < np2) {
                const bool on = (u - 1) * 64 + lane < np2;
                const int rel = P.e2[u].x;
                const uint32_t pk = (uint32_t)P.e2[u].y;
                const int tx = rel >> 16, ty = (int)(short)(rel & 0xffff);
                const int D = (int)(pk & 0xfffffu), j = (int)((pk >> 20) & 255u);
                const unsigned side = (pk >> 28) & 1u;
                const int m = (iabs(tx - genx) + iabs(ty - geny) + 4) * D;
                const unsigned akey = ((unsigned)j << 21) | ((unsigned)(tx + 280) << 10) | (unsigned)(ty + 280);
                const int xy = pack_xy(tx * 4, ty * 4);
                cl_insert(L, lane, on, m, akey | (side << 20), xy);
                if (__any(on && j == 0)) cl_insert(L, lane, on && j == 0, m, akey | (1u << 20), xy);  // the second visit of bucket su[0]
            }
        }
    }
    if (zc < 33 && dmin != 0x7fffffff) {
        const int cx = sx + genx, cy = sy + geny;  // the ring centre, picture coordinates
        const int rmax = 2 * 280 + iabs(genx) + iabs(geny);  // beyond, no position of the diamond is left
        int done_r = -1;  // rings 0 .. done_r are complete
        for (int k0 = 0;; k0 += 64) {
            // position k of the outward enumeration: ring r = the largest r with 2 r (r - 1) + 1 <= k (ring 0 = {k = 0})
            const int k = k0 + lane;
            int r = 0, i = 0;
            if (k > 0) {
                r = (int)((1.0f + __fsqrt_rn((float)(2 * k - 1))) * 0.5f);
                while (2 * r * (r - 1) + 1 > k) r--;
                while (2 * (r + 1) * r + 1 <= k) r++;
                i = k - (2 * r * (r - 1) + 1);  // 0 .. 4r - 1 along the ring
            }
            int dx = 0, dy = 0;
            if (r > 0) {
                const int q = i / r, t = i - q * r;
                dx = q == 0 ? r - t : (q == 1 ? -t : (q == 2 ? t - r : t));
                dy = q == 0 ? t : (q == 1 ? r - t : (q == 2 ? -t : t - r));
            }
            const int px = cx + dx, py = cy + dy;
            const bool inpic = px >= 0 && px < W && py >= 0 && py < H;
            const uint32_t *rec = (const uint32_t *)((const char *)F0 + (__umul24((uint32_t)iclamp(py, 0, H - 1), (uint32_t)W * 12u) + (uint32_t)iclamp(px, 0, W - 1) * 12u));
            const uint32_t a = rec[0], b = rec[1], c = rec[2];
            const int kk0 = (int)(a & 0xffffu), kk1 = (int)(a >> 16), kk2 = (int)(b & 0xffffu);
            const int j = iabs(kk0 - P.su[0]);
            const int tx = px - sx, ty = py - sy;
            const bool member = inpic && j <= J && iabs(tx) + iabs(ty) < 280 && iabs(kk1 - P.su[1]) < 100 && iabs(kk2 - P.su[2]) < 100;
            const int D = feat_dist_w(a, b, c, sp);
            bool take = member && D > 0;  // distance 0 is in the list already
            if (classes) {  // only the modal classes of the big slices: everything else was listed
                bool cls = false;
#pragma unroll
                for (int q = 0; q < FER_BIGS; q++) cls = cls || (q < nbig && kk0 == bigA[q] && D == bigD[q]);
                take = take && cls;
            }
            const int m = (r + 4) * D;
            const int side = kk0 > P.su[0] ? 1 : 0;
            const unsigned akey = ((unsigned)j << 21) | ((unsigned)(tx + 280) << 10) | (unsigned)(ty + 280);
            const int xy = pack_xy(tx * 4, ty * 4);
            cl_insert(L, lane, take, m, akey | ((unsigned)side << 20), xy);
            if (__any(take && j == 0)) cl_insert(L, lane, take && j == 0, m, akey | (1u << 20), xy);  // the second visit of bucket su[0]
            // rings complete after this batch: all k < k0 + 64
            {
                int rr = (int)((1.0f + __fsqrt_rn((float)(2 * (k0 + 64) - 1))) * 0.5f);
                while (2 * rr * (rr - 1) + 1 > k0 + 64) rr--;
                while (2 * (rr + 1) * rr + 1 <= k0 + 64) rr++;
                done_r = rr - 1;  // ring rr has started at or before position k0 + 64, rings below it are complete
            }
            const int t33 = lane_bcast(L.m, 32);
            if (done_r >= rmax) break;
            if (t33 < INF_M && (long long)(done_r + 1 + 4) * dmin > (long long)t33) break;
        }
    }
    L2.m = L.m;
    L2.xy = L.xy;
}

// ---- stage 2: K = 33 of the precomputed candidate set, weighted by the distance to the integer centre (genx, geny)
// -> the list, its length.  Depends on the predictor through its integer part only.
__device__ __forceinline__ int stage2_list(const FerDev &d, int s, int gx, int gy, int lane, const ResPre &P, int genx, int geny,
                                           int *sel_lds, WList &L2)
{
    const int sx = gx * 8, sy = gy * 8;
    L2.m = INF_M;
    L2.xy = 0;
    if (d.basic || FER_DBGF(d, 32)) return 0;
    // a crowded partition without a big slice whose candidates did not fit the list (a few hundred of them, all in small
    // slices): reading them again costs little
    const uint32_t cfl = (uint32_t)lane_bcast(P.e2[0].y, 41);
    const bool rewalk = (cfl & 0x40000000u) != 0 && ((cfl >> 16) & 15u) == 0;
    if (P.n2raw > FER_ST2_CAP && d.zero_cnt[s] == 0 && !rewalk) {
        resolve_crowded(d, s, sx, sy, lane, P, genx, geny, L2);
    } else if (P.n2raw > FER_ST2_CAP) {
        // crowded AND the reference's mis-filed bucket layout (black areas), or the case above: walk the buckets again,
        // now that the predictor is known, through an exact running top-33 (ordered insertion = the reference's own list update)
        const SuPk sp = su_pack(P.su);
        int jx;
        walk_buckets(d, s, P.su, sp, sx, sy, lane, (uint32_t *)sel_lds, 0x7fffffff, jx, [&](bool ok, int rank, int rel, int D, uint32_t) {
            (void)rank;
            int tx = rel >> 16, ty = (int)(short)(rel & 0xffff);
            wl_insert(L2, 33, lane, ok, (iabs(tx - genx) + iabs(ty - geny) + 4) * D, pack_xy(tx * 4, ty * 4));
            return false;
        });
    } else {
        int m2[FER_ST2_CAP / 64];
#pragma unroll
        for (int u = 0; u < FER_ST2_CAP / 64; u++) {
            int cc = u * 64 + lane;
            int tx = P.e2[u].x >> 16, ty = (int)(short)(P.e2[u].x & 0xffff);  // k_me_walk stores (tx << 16) | (ty & 0xffff)
            m2[u] = (int)__umul24((uint32_t)(iabs(tx - genx) + iabs(ty - geny) + 4), (uint32_t)P.e2[u].y) | (cc < P.n2 ? 0 : -1);
        }
        auto raw2 = [&](int u) { return P.e2[u].x; };
        auto fin2 = [&](int e) { return pack_xy((e >> 16) * 4, (int)(short)(e & 0xffff) * 4); };
        select_topk<FER_ST2_CAP / 64>(m2, 33, lane, sel_lds, L2, raw2, fin2);
    }
    return __popcll(__ballot(lane < 33 && L2.m < 100000000));
}

// the chain's own search, stage 2 and stage 3 (precomputed survivors): best (key, vector) of each
template <int WIN>
__device__ __forceinline__ void resolve_stage23(const FerDev &d, int s, int gx, int gy, int lane, const ResPre &P, int mvpx, int mvpy,
                                                int *sel_lds, int &k2, int &xy2, int &k3, int &xy3)
{
    k2 = k3 = 0x7fffffff;
    xy2 = xy3 = 0;
    if (d.basic || FER_DBGF(d, 32)) return;
    WList L2;
    const int cnt2 = stage2_list(d, s, gx, gy, lane, P, mvpx >> 2, mvpy >> 2, sel_lds, L2);
    if (FER_DBGF(d, 256)) L2.xy = pack_xy(lane & 7, lane >> 3);  // (probe: what the scattered SAD rows of stage 2 cost)
    int b2, b2xy;
    sad_keys<33>(L2, cnt2, lane, ip_stream(d, s), d.W, d.H, gx * 8, gy * 8, P.sb, mvpx, mvpy, b2, b2xy);
    wave_best(b2, b2xy, k2, xy2);
    int key = 0x7fffffff;
    if (lane < P.n3) key = ((P.c3s + iabs(P.c3x - mvpx) + iabs(P.c3y - mvpy)) << 6) | lane;
    wave_best(key, pack_xy(P.c3x, P.c3y), k3, xy3);
}

// neighbour vectors of partition (gx, gy) out of a per-stream array of packed vectors [nmb][4] (the guesses v0)
__device__ __forceinline__ ResNbr nbr_from_field(const FerDev &d, const int *vf, int gx, int gy, int lane)
{
    ResNbr N;
    nbr_avail(gx, gy, d.mbw * 2, N.vB, N.vC, N.vD, N.vC16);
    N.vA = gx > 0;
    // lane 0 = B, 1 = C, 2 = D, 3 = C of the 16x16 (P_Skip) predictor, 4 = A
    const int xa = gx + (lane == 0 ? 0 : (lane == 1 ? 1 : (lane == 2 ? -1 : (lane == 3 ? 2 : -1)))), ya = lane == 4 ? gy : gy - 1;
    const bool val = lane == 0 ? N.vB : (lane == 1 ? N.vC : (lane == 2 ? N.vD : (lane == 3 ? N.vC16 : (lane == 4 && N.vA))));
    const int w = val ? vf[part_slot(d.mbw, xa, ya)] : 0;
    N.B = lane_bcast(w, 0);
    N.C = lane_bcast(w, 1);
    N.D = lane_bcast(w, 2);
    N.C16 = lane_bcast(w, 3);
    N.A = lane_bcast(w, 4);
    return N;
}

// ------------------------------------------------------------------ k_me_spec
// The two searches of interEncoding that depend on the neighbours (stage 1 around the predictor, the re-ranking of the
// stage-2 set; F/moestimation.cpp:458-496) depend on them through ONE number pair: the integer part of the predicted
// vector.  It is guessed here for every partition from the neighbours' v0 (the SAD-best stage-3 survivor, k_me_pre),
// and both searches run for the guess in a fully parallel launch -- a workgroup per macroblock, a wavefront per 8x8
// partition -- that leaves their candidate lists with the SAD of every candidate.  k_me_resolve then only checks the
// guess against the true predictor and prices the lists (SAD + |mv - mvp|); a wrong guess sends it through its own search.
// Partition 0 also tries the P_Skip test for the guessed P_Skip vector; when it passes, the other three wavefronts skip
// their searches (the macroblock will most likely be skipped).
#ifndef SPEC_WAVES
#define SPEC_WAVES 7
#endif
template <int WIN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SPEC_WAVES, SPEC_WAVES))) void k_me_spec(FerDev d)
{
    __shared__ __attribute__((aligned(16))) uint32_t loc_all[4][LocalGeo<2>::HTAB + LocalGeo<2>::NB * 64];  // row sums (later the selection's scratch), metrics
    __shared__ int skipflag;
    static_assert(LocalGeo<2>::HTAB >= 256, "the selection's scratch fits the row sums");
    const int lane = threadIdx.x & 63;
    const int part = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int s = blockIdx.y;
    if (d.hdr[s * 4 + 3] != 0) return;
    const int mb = (int)xcd_swizzle(blockIdx.x, gridDim.x);
    const int mbx = mb % d.mbw, mby = mb / d.mbw;
    const int gx = mbx * 2 + (part & 1), gy = mby * 2 + (part >> 1);
    const size_t pidx = ((size_t)s * d.nmb + mb) * 4 + part;
    uint32_t *htab = loc_all[part];
    int *mtab = (int *)loc_all[part] + LocalGeo<2>::HTAB;
    int *sel_lds = (int *)loc_all[part];  // the row sums are dead when the selections start
    const ResNbr N = nbr_from_field(d, d.v0 + (size_t)s * d.nmb * 4, gx, gy, lane);
    int mvpx, mvpy;
    predict_nbr(N.vA, N.A, N.vB, N.B, N.vC, N.C, N.vD, N.D, mvpx, mvpy);
    const int genx = mvpx >> 2, geny = mvpy >> 2;
    const int genw = pack_xy(genx, geny);
    const int W = d.W, H = d.H;
    const IPlanes ip = ip_stream(d, s);
    const int sx = gx * 8, sy = gy * 8;
    // everything the searches will read is requested now, in front of the P_Skip test and its barrier: the launch is bound by
    // the latency of its wavefronts (a chain of memory round trips), not by how many instructions it issues
    constexpr int RRL = (WIN == 16) ? 1 : 2;
    LocalLoads<RRL> ld;
    if (WIN == 32 || WIN == 16) local_request<RRL>(ip, W, H, sx + genx - RRL, sy + geny - RRL, lane, ld);
    ResPre P;
    res_prefetch(d, s, gx, gy, lane, 1, P);
    int smw = 0;
    bool sk = false;
    if (part == 0) {
        int smx, smy;
        pskip_vector(d, mb, mbx, N, smx, smy);
        smw = pack_xy(smx, smy);
        sk = pskip_test(d, s, mbx, mby, lane, smx, smy);
        if (lane == 0) skipflag = sk;
    }
    __syncthreads();
    sk = skipflag != 0;
    if (sk) {  // no lists: if the chain finds the macroblock is not skipped after all, it searches itself
        if (lane == 0) d.spec_hdr[pidx] = make_int4(genw, part == 0 ? (1 << 17) : 0, smw, 1);
        return;
    }
    if (WIN == 32 || WIN == 16) local_metrics<RRL>(ip, W, H, sx + genx - RRL, sy + geny - RRL, su_pack(P.su), lane, htab, mtab, 0, 2, &ld);
    WList L1, L2;
    const int cnt1 = stage1_list<WIN>(d, s, gx, gy, lane, P, genx, geny, sel_lds, mtab, L1);
    const int cnt2 = stage2_list(d, s, gx, gy, lane, P, genx, geny, sel_lds, L2);
    {
        // ONE pass of SADs for both lists: the 17 stage-1 candidates in lanes 0 .. 16, the 33 stage-2 candidates in lanes
        // 17 .. 49 (one memory round trip instead of two: the launch is bound by the latency of its wavefronts)
        const int xy2 = __shfl(L2.xy, (lane - 17) & 63);
        const bool first = lane < 17;
        const int xy = first ? L1.xy : xy2;
        const bool on = first ? lane < cnt1 : (lane - 17 < cnt2 && lane < 50);
        const int cx = on ? unp_x(xy) : 0, cy = on ? unp_y(xy) : 0;
        const int sad = sad_lane(ip, W, H, sx, sy, cx, cy, P.sb);
        if (on) {
            int2 *o = first ? d.spec_l1 + (pidx * 17 + lane) : d.spec_l2 + (pidx * 33 + (lane - 17));
            *o = make_int2(xy, sad);
        }
    }
    if (lane == 0) d.spec_hdr[pidx] = make_int4(genw, cnt1 | (cnt2 << 8) | (1 << 16) | (part == 0 ? (1 << 17) : 0), smw, 0);
}

#ifndef RES_WIN
#define RES_WIN 16  // steps between two loads of the window over the row above (a power of two, RES_WIN + 3 <= 64)
#endif
#ifndef RES_WAVES
#define RES_WAVES 6  // wavefronts per SIMD the chain kernel is compiled for
#endif
// ONE WAVEFRONT PER ROW.  With the guessed lists in place a step of the chain is a few hundred instructions, and with
// thousands of rows in flight the launch is bound by how many instructions it issues, not by the latency of a row: a
// second wavefront per row (which the chain's own search used to be split over) would only repeat them.
template <int WIN>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RES_WAVES, RES_WAVES))) void k_me_resolve(FerDev d)
{
    __shared__ __attribute__((aligned(16))) uint32_t loc_lds[LocalGeo<2>::HTAB + LocalGeo<2>::NB * 64];  // the chain's own search: row sums / selection scratch, metrics
    const int lane = threadIdx.x;
    const int gw = d.mbw * 2, gh = d.mbh * 2;
    // Row tickets.  One ticket queue per XCD (when there are enough streams): stream s belongs to queue s % 8, and a queue
    // hands out its streams `resolve_group` at a time, row-major inside the group.  A workgroup starts on the queue of the
    // XCD it runs on (HW_REG_XCC_ID), so all rows of a stream normally run on ONE XCD: what a partition reads was fetched
    // into that XCD's L2 by the row above a few steps earlier.  When its queue is exhausted the workgroup moves on to the
    // next queue: every row is taken whatever the placement of the workgroups (a grid smaller than 8, a CU mask, a
    // partition mode), and the tail of a picture is shared.  Tickets of one queue are taken in order, so the row a
    // workgroup waits on was claimed before its own by a workgroup that is running or finished: every wait terminates.
    int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int nq = d.S >= 16 ? 8 : 1;                 // queues
    const int q_own = nq == 8 ? (xcc & 7) : 0;        // where this workgroup starts
    int q_done = 0;                                   // queues found exhausted
    const bool spec = d.speculate != 0;
    unsigned st_parts = 0, st_hits = 0, st_skq = 0, st_skh = 0;
    for (;;) {
    const int qu = (q_own + q_done) & (nq - 1);
    const int nsq = (d.S - qu + nq - 1) / nq;         // streams qu, qu + nq, ... of the queue
    int t = 0;
    if (lane == 0) t = atomicAdd(d.chain + qu, 1);
    t = __builtin_amdgcn_readfirstlane(t);
    const int G = min(d.resolve_group, nsq);  // (a group larger than the queue would only mint tickets that name no row)
    const int grp = t / (gh * G), rr = t - grp * (gh * G);
    const int k0 = grp * G, kn = min(G, nsq - k0);  // streams of this group, as indices into the queue's stream list
    if (k0 >= nsq) {
        if (++q_done == nq) break;
        continue;
    }
    const int gy = rr / kn, s = qu + nq * (k0 + (rr - gy * kn));
    if (gy >= gh) continue;  // short last group: its tickets beyond gh * kn name no row
    if (d.hdr[s * 4 + 3] != 0) continue;  // not a P picture: nobody waits on these rows
    unsigned long long *chw = d.chain64 + (size_t)s * d.nmb * 4;
    int *mbt = d.mb_type + (size_t)s * d.nmb;
    short *mvs = d.mv + (size_t)s * d.nmb * 8;
    const unsigned serial = (unsigned)d.serial & 0x7fffffffu;

#ifdef FER_PROBE
    const bool probe = FER_DBGF(d, 128) && s == 0 && gy == gh / 2;  // one row reports where its time goes
#define PR_MARK(k)                        \
    if (probe) {                          \
        long long now_ = wall_clock64();  \
        tacc[k] += now_ - tmark;          \
        tmark = now_;                     \
    }
#else
    const bool probe = false;
#define PR_MARK(k)
#endif
    long long tacc[6] = {0, 0, 0, 0, 0, 0}, tmark = 0;
    int prevw = 0;  // vector of the previous partition of this row (the left neighbour A)
    bool skip = false, timeout = false;
    unsigned cwl = 0, cwh = 0;  // the window over the row above: lane l = partition wbase + l
    int wbase = 0;
    const size_t pidx0 = ((size_t)s * d.nmb + (size_t)(gy >> 1) * d.mbw) * 4 + (size_t)(gy & 1) * 2;  // partition (0, gy)
    const int4 *row_hdr = d.spec_hdr + pidx0;
    const int2 *row_l1 = d.spec_l1 + pidx0 * 17, *row_l2 = d.spec_l2 + pidx0 * 33;
    const int *row_st3 = d.st3 + pidx0 * 33 * 3, *row_st3n = d.st3n + pidx0;
    for (int gx = 0; gx < gw; gx++) {
        if (probe) tmark = wall_clock64();
        const int part = (gy & 1) * 2 + (gx & 1);
        const int mbx = gx >> 1, mby = gy >> 1;
        const int mb = mby * d.mbw + mbx;
        // the lane id is made opaque per iteration: otherwise dozens of lane-derived constants of the loop body
        // are hoisted out of the loop and spilled
        int ln = lane;
        asm volatile("" : "+v"(ln));
        // what the guessed search left behind: the lane's entry of either list and of the stage-3 survivors (addressed
        // relative to the row's first partition: 32-bit offsets per step, the 64-bit bases once per row)
        int4 sh = make_int4(0, 0, 0, 0);
        int2 e1 = make_int2(0, 0), e2 = make_int2(0, 0);
        int c3x = 0, c3y = 0, c3s = 0, n3 = 0;
        if (spec) {
            const unsigned rp = (FER_DBGF(d, 512) ? 0u : (unsigned)((gx >> 1) * 4 + (gx & 1)));  // (probe: list reads served from cache)
            sh = row_hdr[rp];
            e1 = row_l1[rp * 17u + (unsigned)min(ln, 16)];
            e2 = row_l2[rp * 33u + (unsigned)min(ln, 32)];
            if (!d.basic) {
                const int *c3 = row_st3 + (rp * 33u + (unsigned)min(ln, 32)) * 3u;
                c3x = c3[0];
                c3y = c3[1];
                c3s = c3[2];
                n3 = row_st3n[rp];
            }
        }
        // A window over the row above: every RES_WIN steps ONE 64-lane load fetches the words of the partitions gx - 1 ...
        // gx + 62 of that row; the four a step needs come out of it with two ds_bpermute.  A word that was not valid when
        // the window was loaded (the row above was not that far ahead) is polled for.  (Agent-scope loads cross the
        // fabric -- the XCDs' L2s are not coherent --: one per sixteen steps instead of one per step.)
        if ((gx & (RES_WIN - 1)) == 0) {
            wbase = gx - 1;
            const int xa = wbase + ln;
            unsigned long long w = 0;
            if (gy > 0 && xa >= 0 && xa < gw) w = __hip_atomic_load(chw + part_slot(d.mbw, xa, gy - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            cwl = (unsigned)w;
            cwh = (unsigned)(w >> 32);
        }
        // neighbours in the row above: lane 0 = B, 1 = C, 2 = D, 3 = C of the 16x16 (P_Skip) predictor -- the partitions
        // gx, gx + 1, gx - 1, gx + 2 of row gy - 1; which of them exist is wave-uniform (nbr_avail)
        ResNbr N;
        nbr_avail(gx, gy, gw, N.vB, N.vC, N.vD, N.vC16);
        // (bit tricks instead of per-lane selects of wave-uniform booleans, which compile to exec-mask branches)
        const unsigned vmask = (N.vB ? 1u : 0u) | (N.vC ? 2u : 0u) | (N.vD ? 4u : 0u) | (N.vC16 ? 8u : 0u);
        const int noff = (int)((0x3021u >> ((ln & 3) * 4)) & 15u) - 1;  // 0, +1, -1, +2 for lanes 0..3
        const bool val = ln < 4 && ((vmask >> (ln & 3)) & 1u) != 0;
        unsigned wl = (unsigned)__shfl((int)cwl, (gx + noff - wbase) & 63), wh = (unsigned)__shfl((int)cwh, (gx + noff - wbase) & 63);
        bool ok = !val || (wh & 0x7fffffffu) == serial;
        for (int it = 0; !__all(ok); it++) {
            if (val && !ok) {
                unsigned long long w = __hip_atomic_load(chw + part_slot(d.mbw, gx + noff, gy - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                wl = (unsigned)w;
                wh = (unsigned)(w >> 32);
                ok = (wh & 0x7fffffffu) == serial;
            }
            if (__all(ok)) break;
            if (it > RES_SPIN_LIMIT || timeout) {  // never expected: flag it and stop waiting so that the grid drains
                timeout = true;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        PR_MARK(0)
        N.vA = gx > 0;
        N.A = prevw;
        N.B = lane_bcast((int)wl, 0);
        N.C = lane_bcast((int)wl, 1);
        N.D = lane_bcast((int)wl, 2);
        N.C16 = lane_bcast((int)wl, 3);
        if (part == 2) {  // P_Skip was decided by the row above (B is quadrant 0 of this macroblock)
            skip = (lane_bcast((int)wh, 0) & CH_SKIP) != 0;
            if (skip) prevw = N.B;
        }
        if (part == 0 || !skip) {
            bool sk = false;
            int r = 0;
            if (part == 0) {  // ---- P_Skip candidate
                int smx, smy;
                pskip_vector(d, mb, mbx, N, smx, smy);
                r = pack_xy(smx, smy);
                st_skq++;
                if ((((sh.y >> 17) & 1) && sh.z == r) || FER_DBGF(d, 512)) {
                    sk = sh.w != 0 && !FER_DBGF(d, 512);
                    st_skh++;
                } else {
                    sk = pskip_test(d, s, mbx, mby, ln, smx, smy);
                }
            }
            if (!sk) {
                int mvpx, mvpy;
                predict_nbr(N.vA, N.A, N.vB, N.B, N.vC, N.C, N.vD, N.D, mvpx, mvpy);
                st_parts++;
                if ((((sh.y >> 16) & 1) && sh.x == pack_xy(mvpx >> 2, mvpy >> 2)) || FER_DBGF(d, 512)) {
                    // ---- the guess was right: cost = SAD + |mv - mvp| over the three lists; the reference's ordered
                    // first minimum (stage 1, 2, 3 in turn, strict <, F/moestimation.cpp:460-520) is the minimum of
                    // cost << 8 | stage << 6 | list index
                    st_hits++;
                    const int cnt1 = sh.y & 255, cnt2 = (sh.y >> 8) & 255;
                    int key = 0x7fffffff, kxy = 0;
                    if (ln < cnt1) {
                        key = ((e1.y + iabs(unp_x(e1.x) - mvpx) + iabs(unp_y(e1.x) - mvpy)) << 8) | ln;
                        kxy = e1.x;
                    }
                    if (ln < cnt2) {
                        const int k2 = ((e2.y + iabs(unp_x(e2.x) - mvpx) + iabs(unp_y(e2.x) - mvpy)) << 8) | 64 | ln;
                        if (k2 < key) {
                            key = k2;
                            kxy = e2.x;
                        }
                    }
                    if (ln < n3) {
                        const int k3 = ((c3s + iabs(c3x - mvpx) + iabs(c3y - mvpy)) << 8) | 128 | ln;
                        if (k3 < key) {
                            key = k3;
                            kxy = pack_xy(c3x, c3y);
                        }
                    }
                    int wkey;
                    wave_best(key, kxy, wkey, r);
                } else {
                    // ---- the chain's own search: stage 1, then stages 2 and 3
                    ResPre cur;
                    res_prefetch(d, s, gx, gy, ln, 1, cur);
                    int k1, xy1, k2, xy2, k3, xy3;
                    resolve_stage1<WIN>(d, s, gx, gy, ln, cur, mvpx, mvpy, loc_lds, (int *)loc_lds + LocalGeo<2>::HTAB, k1, xy1);
                    PR_MARK(1)
                    resolve_stage23<WIN>(d, s, gx, gy, ln, cur, mvpx, mvpy, (int *)loc_lds, k2, xy2, k3, xy3);
                    PR_MARK(2)
                    int bmin = 2000000000;  // ordered first minimum over stage 1, 2, 3 (strict <)
                    r = 0;
                    if (k1 != 0x7fffffff && (k1 >> 6) < bmin) {
                        bmin = k1 >> 6;
                        r = xy1;
                    }
                    if (k2 != 0x7fffffff && (k2 >> 6) < bmin) {
                        bmin = k2 >> 6;
                        r = xy2;
                    }
                    if (k3 != 0x7fffffff && (k3 >> 6) < bmin) {
                        bmin = k3 >> 6;
                        r = xy3;
                    }
                }
            }
            const unsigned long long w = (unsigned)r | ((unsigned long long)(serial | (sk ? CH_SKIP : 0u)) << 32);
            if (part == 0 && ln == 0) {
                mbt[mb] = sk ? FER_P_SKIP : FER_P_8x8ref0;  // also clears a P_Skip left by the previous picture
                // BasicInterEncoding makes the P_Skip test twice and counts it twice (F/moestimation.cpp:324,421)
                if (sk) atomicAdd(&d.stats[s * 5 + 0], d.basic ? 2 : 1);
            }
            const int nw = sk ? 4 : 1, q0 = sk ? 0 : part;
            if (ln < nw) {
                __hip_atomic_store(chw + (size_t)mb * 4 + q0 + ln, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *(int *)(mvs + ((size_t)mb * 4 + q0 + ln) * 2) = r;
            }
            PR_MARK(3)
            prevw = r;
            if (part == 0) skip = sk;
        }
    }
    if (probe && lane == 0) {
        for (int k = 0; k < 6; k++) d.timing[k] = tacc[k];
        d.timing[7] = gw;
    }
#undef PR_MARK
    if (timeout && lane == 0) atomicOr(&d.status[s], FER_ERR_CHAIN_TIMEOUT);
    }
    if (lane == 0 && st_parts + st_skq > 0) {
        atomicAdd(&d.spec_stat[0], (unsigned long long)st_parts);
        atomicAdd(&d.spec_stat[1], (unsigned long long)st_hits);
        atomicAdd(&d.spec_stat[2], (unsigned long long)st_skq);
        atomicAdd(&d.spec_stat[3], (unsigned long long)st_skh);
    }
}

// ------------------------------------------------------------------ k_basic_stat
// BasicInterEncoding == true: interEncoding first runs basicInterEncoding (F/moestimation.cpp:298-390), whose
// vectors it then discards (:394-397); what survives of that pass is brojTipova.  Its exhaustive loop scores
// partition i with sadLuma8x8(predL, i), which reads the TOP-LEFT 8x8 of predL for every i (:197-212), and moves
// only sub-block 0 of the partition (mvL0x[CurrMbAddr][i][0]); so for i = 0 the SAD varies with the candidate
// through the first 4x4 block alone, and for i = 1..3 it is constant and the first candidate (-W/2, -W/2) stays.
// The macroblock therefore counts as 16x16 when no later candidate of partition 0 beats the first one, as
// 8x8 otherwise (the 16x8 / 8x16 tests cannot hold unless all four agree).  One wavefront per macroblock,
// run between k_me_resolve (P_Skip known) and k_p_resid (which overwrites the source).
__global__ __launch_bounds__(64) void k_basic_stat(FerDev d)
{
    const int lane = threadIdx.x;
    const int s = blockIdx.y, mb = blockIdx.x;
    if (d.hdr[s * 4 + 3] != 0) return;
    if (d.mb_type[(size_t)s * d.nmb + mb] == FER_P_SKIP) return;
    const int W = d.W, H = d.H;
    const uint8_t *Y = d.curY + (size_t)s * d.ysz;
    const uint8_t *RY = d.refY + (size_t)s * d.ysz;
    const IPlanes ip = ip_stream(d, s);
    const int xp = (mb % d.mbw) << 4, yp = (mb / d.mbw) << 4;
    int src[4][4];
#pragma unroll
    for (int y = 0; y < 4; y++) {
        uint32_t v = *(const uint32_t *)(Y + (size_t)(yp + y) * W + xp);
#pragma unroll
        for (int k = 0; k < 4; k++) src[y][k] = (v >> (8 * k)) & 0xff;
    }
    const int R = d.window / 2, n = 2 * R + 1;
    // the first minimum in arrival order: per lane (its candidates arrive in increasing order: strict <), then over the lanes
    // by SAD, then by arrival index among the lanes that hold that SAD (no packing of the two: any WindowSize)
    int bsad = 0x7fffffff, bidx = 0x7fffffff;
    for (int c = lane; c < n * n; c += 64) {
        const int mvx = c / n - R, mvy = c % n - R;  // quarter-pel units, tmvx outer, tmvy inner
        int sad = 0;
#pragma unroll
        for (int y = 0; y < 4; y++) {
            int p[4];
            mc_luma4(RY, ip, W, H, xp, yp, 0, y, mvx, mvy, p);
#pragma unroll
            for (int k = 0; k < 4; k++) sad += iabs(src[y][k] - p[k]);
        }
        if (sad < bsad) {
            bsad = sad;
            bidx = c;
        }
    }
    const int wsad = wave_min(bsad);
    const int widx = wave_min(bsad == wsad ? bidx : 0x7fffffff);
    if (lane == 0) atomicAdd(&d.stats[s * 5 + (widx == 0 ? 1 : 4)], 1);
}

void fer_launch_basic_stat(const FerDev &d, hipStream_t st)
{
    if (d.basic) hipLaunchKernelGGL(k_basic_stat, dim3(d.nmb, d.S), dim3(64), 0, st, d);
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

void fer_launch_me_walk(const FerDev &d, hipStream_t st)
{
    if (d.basic) return;  // stage 2 is not run (F/moestimation.cpp:470)
    hipLaunchKernelGGL(k_me_walk, dim3(d.nmb * 4, d.S), dim3(64), 0, st, d);
}

void fer_launch_me_spec(const FerDev &d, hipStream_t st)
{
    if (!d.speculate) return;
    dim3 g(d.nmb, d.S);
    if (d.window == 32)
        hipLaunchKernelGGL(k_me_spec<32>, g, dim3(256), 0, st, d);
    else if (d.window == 16)
        hipLaunchKernelGGL(k_me_spec<16>, g, dim3(256), 0, st, d);
    else
        hipLaunchKernelGGL(k_me_spec<0>, g, dim3(256), 0, st, d);
}

void fer_launch_me_resolve(const FerDev &d, hipStream_t st)
{
    const int gh = 2 * d.mbh;
    hipMemsetAsync(d.chain, 0, sizeof(int) * 8, st);
    const int cap = d.resolve_wgs;
    dim3 g(min(gh * d.S, max(cap, 1)));
    if (d.window == 32)
        hipLaunchKernelGGL(k_me_resolve<32>, g, dim3(64), 0, st, d);
    else if (d.window == 16)
        hipLaunchKernelGGL(k_me_resolve<16>, g, dim3(64), 0, st, d);
    else
        hipLaunchKernelGGL(k_me_resolve<0>, g, dim3(64), 0, st, d);
}
