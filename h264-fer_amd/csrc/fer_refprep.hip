// fer_refprep.hip -- reference-picture preparation kernels (row a16 of SURVEY.md 8a):
// the work FillInterpolatedRefFrame does once per picture (F/moestimation.cpp:74-173),
// re-designed for HBM streaming instead of a 2-D integral image:
//   k_interp    16 quarter-pel planes of the reconstructed luma        (:76-103 via F/mocomp.cpp:50-78)
//   k_features  the five 8x8 box features of every position/plane     (:105-138), uint16, no integral image
//   sort        positions of plane 0 ordered by (8x8 sum, tx, ty)      (:140-172) = stable counting sort
//   k_frame_sad |cur - ref| over the luma plane for the IDR decision   (F/ref_frames.cpp:210-219)
#include <cstring>
#include <string.h>
#include "fer_internal.h"

// ------------------------------------------------------------------ k_interp
// block = 64x4 threads, each thread one pixel; LDS tile (64+5) x (4+5) of the clamped reference.
#define IT_W 64
#define IT_H 4
__global__ __launch_bounds__(256) void k_interp(FerDev d)
{
    __shared__ uint8_t tile[IT_H + 5][IT_W + 8];
    int s = blockIdx.z;
    if (d.hdr[s * 4 + 3] != 0) return;  // only P pictures search
    const uint8_t *R = d.refY + (size_t)s * d.ysz;
    uint8_t *P = d.interp + (size_t)s * 16 * d.ysz;
    int x0 = blockIdx.x * IT_W, y0 = blockIdx.y * IT_H;
    int tid = threadIdx.y * IT_W + threadIdx.x;
    for (int i = tid; i < (IT_H + 5) * (IT_W + 5); i += 256) {
        int ty = i / (IT_W + 5), tx = i % (IT_W + 5);
        int sx = iclamp(x0 + tx - 2, 0, d.W - 1), sy = iclamp(y0 + ty - 2, 0, d.H - 1);
        tile[ty][tx] = R[sy * d.W + sx];
    }
    __syncthreads();
    int x = x0 + threadIdx.x, y = y0 + threadIdx.y;
    if (x >= d.W || y >= d.H) return;
    int lx = threadIdx.x + 2, ly = threadIdx.y + 2;
#define T(dx, dy) ((int)tile[ly + (dy)][lx + (dx)])
    int G = T(0, 0);
    int b = tap6(T(-2, 0), T(-1, 0), G, T(1, 0), T(2, 0), T(3, 0));
    int h = tap6(T(0, -2), T(0, -1), G, T(0, 1), T(0, 2), T(0, 3));
    int m = tap6(T(1, -2), T(1, -1), T(1, 0), T(1, 1), T(1, 2), T(1, 3));
    int sS = tap6(T(-2, 1), T(-1, 1), T(0, 1), T(1, 1), T(2, 1), T(3, 1));
    int cc = tap6(T(-2, -2), T(-2, -1), T(-2, 0), T(-2, 1), T(-2, 2), T(-2, 3));
    int dd = tap6(T(-1, -2), T(-1, -1), T(-1, 0), T(-1, 1), T(-1, 2), T(-1, 3));
    int ee = tap6(T(2, -2), T(2, -1), T(2, 0), T(2, 1), T(2, 2), T(2, 3));
    int ff = tap6(T(3, -2), T(3, -1), T(3, 0), T(3, 1), T(3, 2), T(3, 3));
    int j = tap6(cc, dd, h, m, ee, ff);
    int v[16];
    v[0] = G;
    v[1] = FER_MID(G, b);
    v[2] = b;
    v[3] = FER_MID(b, T(1, 0));
    v[4] = FER_MID(G, h);
    v[5] = FER_MID(b, h);
    v[6] = FER_MID(b, j);
    v[7] = FER_MID(b, m);
    v[8] = h;
    v[9] = FER_MID(h, j);
    v[10] = j;
    v[11] = FER_MID(j, m);
    v[12] = FER_MID(h, T(0, 1));
    v[13] = FER_MID(h, sS);
    v[14] = FER_MID(j, sS);
    v[15] = FER_MID(sS, m);
#undef T
    size_t o = (size_t)y * d.W + x;
#pragma unroll
    for (int f = 0; f < 16; f++) P[(size_t)f * d.ysz + o] = (uint8_t)v[f];
}

// ------------------------------------------------------------------ k_features
// The five 8x8 box features of every (position, plane), F/moestimation.cpp:105-138, without an
// integral image.  A wavefront owns a strip of 4 positions x FS_ROWS rows; lane = (position, plane):
// 16 lanes of one position hold its 16 planes, so each output row is ONE contiguous 768-byte store
// of finished 12-byte records [k0..k4,pad].  Per input row a lane reads the 8 samples x..x+7 of its
// plane (three aligned dwords), reduces them with v_sad_u8 against zero to the three horizontal
// partial sums, and keeps the last 8 rows in registers; vertical sums come straight from that ring.
// Samples beyond the picture replicate the last row / column (the reference pads by 8).
#define FS_ROWS 64
// the 8 samples x..x+7 of a row as three aligned dwords (fetched without control flow; a row that
// runs over the right picture edge reads into the next row, inside the allocation)
struct FeatRow {
    uint32_t w0, w1, w2;
};
__device__ __forceinline__ FeatRow feat_row_load(const uint8_t *__restrict__ row, int x)
{
    const uint8_t *p = row + x;
    const uint32_t *a = (const uint32_t *)(p - ((uintptr_t)p & 3));
    FeatRow r;
    r.w0 = a[0];
    r.w1 = a[1];
    r.w2 = a[2];
    return r;
}
// horizontal partial sums: all 8 samples, the left 4, and samples {0,1,4,5}; nv = samples left of the right edge
__device__ __forceinline__ void feat_hsum(const FeatRow &r, int sh, int nv, int &h8, int &h4, int &hc)
{
    uint32_t lo = __builtin_amdgcn_alignbyte(r.w1, r.w0, (uint32_t)sh);
    uint32_t hi = __builtin_amdgcn_alignbyte(r.w2, r.w1, (uint32_t)sh);
    if (__any(nv < 8)) {  // the reference pads by replicating the last column
        if (nv < 8) {
            unsigned long long v = ((unsigned long long)hi << 32) | lo;
            unsigned long long last = (v >> (8 * (nv - 1))) & 0xffull;
            unsigned long long keep = (1ull << (8 * nv)) - 1ull;
            v = (v & keep) | ((last * 0x0101010101010101ull) & ~keep);
            lo = (uint32_t)v;
            hi = (uint32_t)(v >> 32);
        }
    }
    h4 = (int)__builtin_amdgcn_sad_u8(lo, 0u, 0u);
    h8 = (int)__builtin_amdgcn_sad_u8(hi, 0u, (uint32_t)h4);
    hc = (int)__builtin_amdgcn_sad_u8(hi & 0xffffu, 0u, __builtin_amdgcn_sad_u8(lo & 0xffffu, 0u, 0u));
}

__global__ __launch_bounds__(256) void k_features(FerDev d)
{
    const int lane = threadIdx.x & 63;
    const int xgroups = d.W >> 2, nstrips = (d.H + FS_ROWS - 1) / FS_ROWS;
    long long wid = (long long)xcd_swizzle(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
    if (wid >= (long long)xgroups * nstrips * d.S) return;
    const int xg = (int)(wid % xgroups);
    const int strip = (int)((wid / xgroups) % nstrips);
    const int s = (int)(wid / ((long long)xgroups * nstrips));
    if (d.hdr[s * 4 + 3] != 0) return;
    const int f = lane & 15, x = xg * 4 + (lane >> 4);
    const int W = d.W, H = d.H;
    const uint8_t *P = d.interp + ((size_t)s * 16 + f) * d.ysz;
    const int y0 = strip * FS_ROWS, y1 = min(y0 + FS_ROWS, H);  // output rows [y0, y1)
    uint32_t *out = (uint32_t *)(d.feat + (size_t)s * 96 * d.ysz);
    uint32_t *out0 = (uint32_t *)(d.feat0 + (size_t)s * 6 * d.ysz);
    const int sh = (int)((uintptr_t)(P + x) & 3);  // W is a multiple of 16: the same byte offset in every row
    const int nv = W - x;
    int h8[8], h4[8], hc[8];
    for (int base = 0; base < FS_ROWS + 8; base += 8) {
        if (y0 + base - 7 >= y1) break;
        // the 8 input rows of this round are requested together; a row at a time would leave the wavefront
        // waiting on one memory round trip per output row
        FeatRow rr[8];
#pragma unroll
        for (int j = 0; j < 8; j++) rr[j] = feat_row_load(P + (size_t)min(y0 + base + j, H - 1) * W, x);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int y = y0 + base + j;  // input row (clamped), completes the window of output row y - 7
            feat_hsum(rr[j], sh, nv, h8[j], h4[j], hc[j]);
            const int yo = y - 7;
            if (yo >= y0 && yo < y1) {
                // slot of window row r (picture row yo + r) is (j + 1 + r) & 7
                int k0 = 0, k2 = 0, k4 = 0;
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    k0 += h8[r];
                    k2 += h4[r];
                    k4 += hc[r];
                }
                int k1 = h8[(j + 1) & 7] + h8[(j + 2) & 7] + h8[(j + 3) & 7] + h8[(j + 4) & 7];
                int k3 = h8[(j + 1) & 7] + h8[(j + 2) & 7] + h8[(j + 5) & 7] + h8[(j + 6) & 7];
                uint32_t a = (uint32_t)k0 | ((uint32_t)k1 << 16), b = (uint32_t)k2 | ((uint32_t)k3 << 16),
                         c = (uint32_t)k4;
                size_t pos = (size_t)yo * W + x;
                uint32_t *o = out + (pos * 16 + f) * 3;  // 420 MB per picture, read back much later: streaming stores
                __builtin_nontemporal_store(a, o);
                __builtin_nontemporal_store(b, o + 1);
                __builtin_nontemporal_store(c, o + 2);
                if (f == 0) {
                    uint32_t *o0 = out0 + pos * 3;
                    o0[0] = a;
                    o0[1] = b;
                    o0[2] = c;
                }
            }
        }
    }
}

// ------------------------------------------------------------------ sort by 8x8 sum
// keys in arrival order b = tx*H + ty (the reference scans columns, F/moestimation.cpp:142-151)
__global__ void k_sort_keys(FerDev d, uint32_t *keys, uint32_t *vals)
{
    const int s = blockIdx.y;
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    int n = d.W * d.H;
    if (b >= n) return;
    if (d.hdr[s * 4 + 3] != 0) return;
    int tx = b / d.H, ty = b % d.H;
    uint16_t k = d.feat0[((size_t)s * d.ysz + (size_t)ty * d.W + tx) * 6];
    keys[(size_t)s * n + b] = k;
    vals[(size_t)s * n + b] = ((uint32_t)tx << 16) | (uint32_t)ty;
    if (k == 0) atomicAdd(&d.zero_cnt[s], 1);  // the reference mis-files sum 0: see k_sort_finish
}

// Payload of the sorted order + the two-level bucket index.  Record i opens every (sum, column tile) bin
// after its predecessor's up to its own: kol2[bin] = i for those bins (lower bound of the bin in the sorted
// order).  Gaps are short except at the ends of the sum range; long ones are filled by the whole wavefront.
//
// Bucket 0.  The reference's counting sort (F/moestimation.cpp:153-172) leaves bucket 0 out of its prefix sum:
// with n0 positions of sum 0, every other bucket starts n0 places early (the sorted array is the other
// positions from 0, its last n0 places keep what the previous picture left there), the k-th sum-0 position
// is written to place n0 + k, where it replaces, or is replaced by, the regular occupant -- whichever the
// scatter loop reaches later in arrival order --, bucket 0 reads as [0, 2 n0) and bucket 1 starts at 2 n0.
// That layout is reproduced here for a stream with n0 > 0 (black areas in full-range content); the walk then
// scans whole buckets by these rules (walk_buckets).  The index kol2 keeps describing the plain sorted order.
__device__ __forceinline__ void sort_record(const FerDev &d, int s, uint32_t v, uint32_t *o)
{
    int tx = v >> 16, ty = v & 0xffff;
    const uint32_t *r = (const uint32_t *)(d.feat0 + ((size_t)s * d.ysz + (size_t)ty * d.W + tx) * 6);
    uint32_t a = r[0], b = r[1], c = r[2];  // k0|k1<<16, k2|k3<<16, k4
    o[0] = v;
    o[1] = (a >> 16) | (b << 16);
    o[2] = (b >> 16) | (c << 16);
}

__global__ __launch_bounds__(256) void k_sort_finish(FerDev d, const uint32_t *skeys, const uint32_t *svals)
{
    const int s = blockIdx.y;
    if (d.hdr[s * 4 + 3] != 0) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = d.W * d.H;
    const int nb = 16384 * d.kt;
    const int n0 = d.zero_cnt[s];
    uint32_t *kol2 = d.kol2 + (size_t)s * nb;
    const size_t g0 = (size_t)s * n;
    int gap_lo = 0, gap_hi = 0;  // bins [gap_lo, gap_hi) get value gval
    uint32_t gval = 0;
    if (i < n) {
        const uint32_t v = svals[g0 + i];
        const int tx = v >> 16;
        d.sort_pos[g0 + i] = v;
        if (n0 == 0) {
            sort_record(d, s, v, d.sort_rec + (g0 + i) * 3);
        } else if (i >= n0) {  // a regular position: n0 places early, unless the sum-0 position aimed there comes later
            const int p = i - n0;
            uint32_t w = v;
            if (p >= n0 && p < 2 * n0) {
                const uint32_t z = svals[g0 + p - n0];
                const int bz = (int)(z >> 16) * d.H + (int)(z & 0xffff), bn = tx * d.H + (int)(v & 0xffff);
                if (bz > bn) w = z;
            }
            sort_record(d, s, w, d.sort_rec + (g0 + p) * 3);
        } else {  // a sum-0 position: place n0 + i, written here only where no regular position lands
            const int p = n0 + i;
            if (p >= n - n0 && p < n) sort_record(d, s, v, d.sort_rec + (g0 + p) * 3);
        }
        int bin = (int)(skeys[g0 + i] & 0x7fff) * d.kt + (tx >> d.ktw_shift);
        int prev = -1;
        if (i > 0) {
            uint32_t pv = svals[g0 + i - 1];
            prev = (int)(skeys[g0 + i - 1] & 0x7fff) * d.kt + (int)((pv >> 16) >> d.ktw_shift);
        }
        gap_lo = prev + 1;
        gap_hi = bin + 1;
        gval = (uint32_t)(g0 + i);
    }
    // the last record also closes the index: every bin after its own, and the end marker, start at the segment end
    const bool tail = i == n - 1;
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 1) {
            if (!__any(tail)) break;
            gap_lo = tail ? gap_hi : 0;
            gap_hi = tail ? nb + 1 : 0;
            gval = (uint32_t)(g0 + n);
        }
        unsigned long long big = __ballot(gap_hi - gap_lo > 8);
        if (gap_hi - gap_lo <= 8)
            for (int b = gap_lo; b < gap_hi; b++) kol2[b] = gval;
        while (big) {
            int src = __ffsll((long long)big) - 1;
            big &= big - 1;
            int lo = __shfl(gap_lo, src), hi = __shfl(gap_hi, src);
            uint32_t val = (uint32_t)__shfl((int)gval, src);
            for (int b = lo + (threadIdx.x & 63); b < hi; b += 64) kol2[b] = val;
        }
    }
}

// ---- stable LSD radix sort of one stream's (sum, position) pairs: two passes of 8 + 7 bits ----
// The order the reference's counting sort produces (F/moestimation.cpp:140-172) is "by sum, ties in arrival
// order", i.e. a stable sort of the arrival sequence.  Each pass is three launches -- per-tile digit
// histograms, one exclusive scan per stream over (digit, tile), stable scatter -- and no block ever waits on
// another one, so the sort keeps its speed when another context's kernels share the GPU (a single-pass
// look-back sort stalls there).  Streams are independent segments (blockIdx.y).
#define RS_THREADS 256
#define RS_ITEMS 16
#define RS_TILE (RS_THREADS * RS_ITEMS)

__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(FerDev d, const uint32_t *keys, uint32_t *hist, int ntiles, int shift)
{
    __shared__ unsigned h[256];
    const int s = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    if (d.hdr[s * 4 + 3] != 0) return;
    const int n = d.W * d.H;
    keys += (size_t)s * n;
    h[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        int idx = tile * RS_TILE + i * RS_THREADS + tid;
        if (idx < n) atomicAdd(&h[(keys[idx] >> shift) & 0xff], 1u);
    }
    __syncthreads();
    hist[((size_t)s * 256 + tid) * ntiles + tile] = h[tid];  // digit-major: the scan order is the output order
}

__global__ __launch_bounds__(256) void k_rs_scan(FerDev d, uint32_t *hist, int ntiles)
{
    __shared__ unsigned part[256];
    const int s = blockIdx.x, tid = threadIdx.x;
    if (d.hdr[s * 4 + 3] != 0) return;
    uint32_t *h = hist + (size_t)s * 256 * ntiles;
    const int n = 256 * ntiles;
    const int per = (n + 255) / 256;
    const int b0 = min(tid * per, n), b1 = min(b0 + per, n);
    unsigned sum = 0;
    for (int i = b0; i < b1; i++) sum += h[i];
    part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        unsigned v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    unsigned run = tid ? part[tid - 1] : 0;
    for (int i = b0; i < b1; i++) {
        unsigned v = h[i];
        h[i] = run;
        run += v;
    }
}

__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(FerDev d, const uint32_t *keys_in, const uint32_t *vals_in,
                                                          uint32_t *keys_out, uint32_t *vals_out, const uint32_t *hist,
                                                          int ntiles, int shift)
{
    __shared__ unsigned run[RS_THREADS / 64][256];  // per wavefront: items of each digit seen so far
    __shared__ unsigned base[256];
    const int s = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    if (d.hdr[s * 4 + 3] != 0) return;
    const int lane = tid & 63, wv = tid >> 6;
    const int n = d.W * d.H;
    keys_in += (size_t)s * n;
    vals_in += (size_t)s * n;
    keys_out += (size_t)s * n;
    vals_out += (size_t)s * n;
#pragma unroll
    for (int w = 0; w < RS_THREADS / 64; w++) run[w][tid] = 0;
    base[tid] = hist[((size_t)s * 256 + tid) * ntiles + tile];
    __syncthreads();
    // a wavefront owns a contiguous quarter of the tile and walks it in arrival order, 64 items per round
    const int w0 = tile * RS_TILE + wv * (RS_TILE / (RS_THREADS / 64));
    uint32_t key[RS_ITEMS], val[RS_ITEMS];
    unsigned rk[RS_ITEMS];  // rank among the wavefront's earlier items of the same digit
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        int idx = w0 + r * 64 + lane;
        bool ok = idx < n;
        key[r] = ok ? keys_in[idx] : 0xffffffffu;
        val[r] = ok ? vals_in[idx] : 0u;
    }
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        const bool ok = w0 + r * 64 + lane < n;
        const unsigned dg = (key[r] >> shift) & 0xff;
        unsigned long long peers = __ballot(ok);  // lanes with the same digit
#pragma unroll
        for (int b = 0; b < 8; b++) {
            unsigned long long bal = __ballot((dg >> b) & 1);
            peers &= ((dg >> b) & 1) ? bal : ~bal;
        }
        const unsigned before = run[wv][dg];
        rk[r] = before + (unsigned)__popcll(peers & lt);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (ok && (peers & lt) == 0) run[wv][dg] = before + (unsigned)__popcll(peers);  // first lane of the group
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // The tile is first put in output order inside LDS (digit runs one after the other), then written out: a
    // thread's neighbours write neighbouring addresses of the same digit run instead of 256 scattered streams.
    __shared__ unsigned dstart[256];
    __shared__ uint32_t sk[RS_TILE], sv[RS_TILE];
    unsigned tot = 0;
    {  // digit tid: wavefront totals -> offsets inside the digit's run of this tile
#pragma unroll
        for (int w = 0; w < RS_THREADS / 64; w++) {
            unsigned c = run[w][tid];
            run[w][tid] = tot;
            tot += c;
        }
    }
    {  // exclusive prefix of the tile's digit totals (256 values, one per thread)
        unsigned inc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            unsigned v = __shfl_up(inc, o);
            if (lane >= o) inc += v;
        }
        __shared__ unsigned wsum[RS_THREADS / 64];
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        unsigned before = 0;
        for (int w = 0; w < wv; w++) before += wsum[w];
        dstart[tid] = before + inc - tot;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        if (w0 + r * 64 + lane < n) {
            const unsigned dg = (key[r] >> shift) & 0xff;
            const unsigned lp = dstart[dg] + run[wv][dg] + rk[r];
            sk[lp] = key[r];
            sv[lp] = val[r];
        }
    }
    __syncthreads();
    const int cnt = min(RS_TILE, n - tile * RS_TILE);
    for (int i = tid; i < cnt; i += RS_THREADS) {
        const uint32_t k = sk[i];
        const unsigned dg = (k >> shift) & 0xff;
        const unsigned pos = base[dg] + ((unsigned)i - dstart[dg]);
        keys_out[pos] = k;
        vals_out[pos] = sv[i];
    }
}

size_t fer_sort_tmp_bytes(int n, int S)
{
    const int ntiles = (n + RS_TILE - 1) / RS_TILE;
    return (size_t)S * 256 * ntiles * sizeof(uint32_t);
}

// host side: prepare the reference structures of all P-picture streams (three profiled steps)
void fer_launch_interp(const FerDev &d, hipStream_t st)
{
    dim3 gi((d.W + IT_W - 1) / IT_W, (d.H + IT_H - 1) / IT_H, d.S);
    hipLaunchKernelGGL(k_interp, gi, dim3(IT_W, IT_H), 0, st, d);
}

void fer_launch_features(const FerDev &d, hipStream_t st)
{
    long long fw = (long long)(d.W >> 2) * ((d.H + FS_ROWS - 1) / FS_ROWS) * d.S;
    hipLaunchKernelGGL(k_features, dim3((unsigned)((fw + 3) / 4)), dim3(256), 0, st, d);
}

// the sort in three profiled steps: keys, the two radix passes, payload + bucket index
void fer_launch_sort_keys(const FerDev &d, FerSortTmp &t, hipStream_t st)
{
    int n = d.W * d.H;
    hipMemsetAsync(d.zero_cnt, 0, sizeof(int) * d.S, st);
    hipLaunchKernelGGL(k_sort_keys, dim3((n + 255) / 256, d.S), dim3(256), 0, st, d, t.keys_in, t.vals_in);
}

void fer_launch_sort_radix(const FerDev &d, FerSortTmp &t, hipStream_t st)
{
    int n = d.W * d.H;
    const int ntiles = (n + RS_TILE - 1) / RS_TILE;
    uint32_t *hist = (uint32_t *)t.tmp;
    uint32_t *ki = t.keys_in, *vi = t.vals_in, *ko = t.keys_out, *vo = t.vals_out;
    for (int pass = 0; pass < 2; pass++) {  // sum bits 0-7, then 8-14
        const int shift = pass * 8;
        hipLaunchKernelGGL(k_rs_hist, dim3(ntiles, d.S), dim3(RS_THREADS), 0, st, d, ki, hist, ntiles, shift);
        hipLaunchKernelGGL(k_rs_scan, dim3(d.S), dim3(256), 0, st, d, hist, ntiles);
        hipLaunchKernelGGL(k_rs_scatter, dim3(ntiles, d.S), dim3(RS_THREADS), 0, st, d, ki, vi, ko, vo, hist, ntiles, shift);
        uint32_t *x = ki;
        ki = ko;
        ko = x;
        x = vi;
        vi = vo;
        vo = x;
    }
}

void fer_launch_sort_finish(const FerDev &d, FerSortTmp &t, hipStream_t st)
{
    int n = d.W * d.H;
    // after two passes the sorted pairs are back in keys_in / vals_in
    hipLaunchKernelGGL(k_sort_finish, dim3((n + 255) / 256, d.S), dim3(256), 0, st, d, t.keys_in, t.vals_in);
}

void fer_launch_sort(const FerDev &d, FerSortTmp &t, hipStream_t st)
{
    fer_launch_sort_keys(d, t, st);
    fer_launch_sort_radix(d, t, st);
    fer_launch_sort_finish(d, t, st);
}

void fer_launch_refprep(const FerDev &d, FerSortTmp &t, const int *types, hipStream_t st)
{
    (void)types;
    fer_launch_interp(d, st);
    fer_launch_features(d, st);
    fer_launch_sort(d, t, st);
}

// ------------------------------------------------------------------ k_frame_sad
__global__ __launch_bounds__(256) void k_frame_sad(FerDev d)
{
    int s = blockIdx.y;
    const uint8_t *a = d.curY + (size_t)s * d.ysz, *b = d.refY + (size_t)s * d.ysz;
    size_t n4 = d.ysz / 4;
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t va = ((const uint32_t *)a)[i], vb = ((const uint32_t *)b)[i];
        acc = __builtin_amdgcn_sad_u8(va, vb, acc);
    }
    int v = wave_sum((int)acc);
    if ((threadIdx.x & 63) == 0) atomicAdd(&d.sad[s], (unsigned long long)(unsigned)v);
}

void fer_launch_frame_sad(const FerDev &d, hipStream_t st)
{
    hipMemsetAsync(d.sad, 0, sizeof(unsigned long long) * d.S, st);
    hipLaunchKernelGGL(k_frame_sad, dim3(256, d.S), dim3(256), 0, st, d);
}
