// fer_refprep.hip -- reference-picture preparation kernels (row a16 of SURVEY.md 8a):
// the work FillInterpolatedRefFrame does once per picture (F/moestimation.cpp:74-173),
// re-designed for HBM streaming instead of a 2-D integral image:
//   k_interp    16 quarter-pel planes of the reconstructed luma        (:76-103 via F/mocomp.cpp:50-78)
//   k_features  the five 8x8 box features of every position/plane     (:105-138), uint16, no integral image
//   sort        positions of plane 0 ordered by (8x8 sum, tx, ty)      (:140-172) = stable counting sort
//   k_frame_sad |cur - ref| over the luma plane for the IDR decision   (F/ref_frames.cpp:210-219)
#include <cstring>
#include <string.h>
#include <algorithm>
#include "fer_internal.h"

// ------------------------------------------------------------------ k_interp
// A workgroup = a 128 x 8 tile of the picture (a wavefront stores two full 128-byte lines per plane; 64 x 16 tiles
// measured 3.33 ms per 256-stream picture against 3.09), a thread = four samples side by side of one row.  The tile and its
// 2 / 3-sample apron sit in LDS shifted by two columns, so that the 12 bytes a thread needs of each of its six rows
// (columns -2 .. +9 around its first sample) are three aligned dwords.  The nine vertical 6-tap values of columns
// -2 .. +6 are shared by the four samples (the centre sample j filters them AFTER clipping, F/mocomp.cpp:71); every
// plane gets one dword per thread.  Tiles are dealt so that one XCD owns a band of the picture: the apron rows a tile
// shares with the tiles above and below are then fetched into one L2.
#ifndef IT_W
#define IT_W 128
#define IT_H 8
#endif
typedef short s16x2_t __attribute__((ext_vector_type(2)));
#define IT_PITCH (IT_W + 8)  // bytes of a tile row: IT_W + 5 apron columns, rounded up to dwords
__global__ __launch_bounds__(256) void k_interp(FerDev d)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[IT_H + 5][IT_PITCH];
    const int s = blockIdx.y;
    if (d.hdr[s * 4 + 3] != 0) return;  // only P pictures search
    const int W = d.W, H = d.H;
    const uint8_t *R = d.refY + (size_t)s * d.ysz;
    uint8_t *P = d.interp + (size_t)s * 16 * d.iplane + d.ioff;
    const int tw = (W + IT_W - 1) / IT_W;
    const int t = (int)xcd_swizzle(blockIdx.x, gridDim.x);
    const int x0 = (t % tw) * IT_W, y0 = (t / tw) * IT_H;
    const int tid = threadIdx.x;
    {  // the tile: every load of the thread is requested before the first one is waited for
        constexpr int NL = ((IT_H + 5) * (IT_PITCH / 4) + 255) / 256;
        uint32_t v[NL];
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int i = min(tid + k * 256, (IT_H + 5) * (IT_PITCH / 4) - 1);
            const int r = i / (IT_PITCH / 4), c4 = (i % (IT_PITCH / 4)) * 4;
            const int y = iclamp(y0 + r - 2, 0, H - 1), xs = x0 + c4 - 2;  // tile byte c4 = picture column xs
            if (xs >= 0 && xs + 3 < W) {
                v[k] = load_u8x4(R + (size_t)y * W + xs);
            } else {
                v[k] = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) v[k] |= (uint32_t)R[(size_t)y * W + iclamp(xs + b, 0, W - 1)] << (8 * b);
            }
        }
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int i = tid + k * 256;
            if (i < (IT_H + 5) * (IT_PITCH / 4)) *(uint32_t *)&tile[i / (IT_PITCH / 4)][(i % (IT_PITCH / 4)) * 4] = v[k];
        }
    }
    __syncthreads();
    const int g = tid % (IT_W / 4), ty = tid / (IT_W / 4);
    const int x = x0 + g * 4, y = y0 + ty;
    if (x >= W || y >= H) return;
    // Rows y-2 .. y+3, columns x-2 .. x+9 of the tile as three dwords per row.  The six-tap filters run on PAIRS of
    // neighbouring samples (two 16-bit lanes per register, v_pk_*: every intermediate fits 16 bits), the twelve averaged
    // planes on four samples packed in a dword ((a | b) - (((a ^ b) >> 1) & 0x7f7f7f7f) = (a + b + 1) >> 1 per byte).
    uint32_t w[6][3];
#pragma unroll
    for (int r = 0; r < 6; r++) {
        const uint32_t *q = (const uint32_t *)&tile[ty + r][g * 4];
        w[r][0] = q[0];
        w[r][1] = q[1];
        w[r][2] = q[2];
    }
    // pr(r, c) = samples c, c + 1 of row r (c = 0 is column x - 2) as 16-bit lanes
    auto pr = [&](int r, int c) -> s16x2_t {
        const uint32_t lo = w[r][c >> 2], hi = w[r][c >> 2 < 2 ? (c >> 2) + 1 : 2];
        const uint32_t i0 = (uint32_t)(c & 3), i1 = i0 + 1;  // (sample 3 of a dword pairs with byte 0 of the next one)
        return __builtin_bit_cast(s16x2_t, __builtin_amdgcn_perm(hi, lo, 0x0c000c00u | (i1 << 16) | i0));
    };
    auto tap6p = [](s16x2_t E, s16x2_t F, s16x2_t G, s16x2_t H, s16x2_t I, s16x2_t J) -> s16x2_t {
        const s16x2_t c20 = {20, 20}, c5 = {-5, -5}, c16 = {16, 16}, c0 = {0, 0}, c255 = {255, 255};
        s16x2_t t = (G + H) * c20 + c16;
        t = (F + I) * c5 + t;
        t = (t + (E + J)) >> 5;
        return __builtin_elementwise_min(__builtin_elementwise_max(t, c0), c255);
    };
    auto pack4 = [](s16x2_t a, s16x2_t b2) -> uint32_t {  // four clipped samples -> bytes
        return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, b2), __builtin_bit_cast(uint32_t, a), 0x06040200u);
    };
    auto mid4 = [](uint32_t a, uint32_t b2) -> uint32_t { return (a | b2) - (((a ^ b2) >> 1) & 0x7f7f7f7fu); };
    // vertical half samples of columns 0 .. 8 (pairs at the even columns), then the pairs at the odd ones
    s16x2_t VP[8];
#pragma unroll
    for (int c = 0; c < 10; c += 2) {
        const s16x2_t v = tap6p(pr(0, c), pr(1, c), pr(2, c), pr(3, c), pr(4, c), pr(5, c));
        if (c < 8) VP[c] = v;
        if (c >= 2)  // (V[c-1], V[c]) from (V[c-2], V[c-1]) and (V[c], V[c+1])
            VP[c - 1] = __builtin_bit_cast(s16x2_t, __builtin_amdgcn_alignbit(__builtin_bit_cast(uint32_t, v), __builtin_bit_cast(uint32_t, VP[c - 2]), 16));
    }
    const uint32_t G4 = __builtin_amdgcn_alignbyte(w[2][1], w[2][0], 2);   // the samples themselves
    const uint32_t Gn4 = __builtin_amdgcn_alignbyte(w[2][1], w[2][0], 3);  // ... one column on
    const uint32_t Gd4 = __builtin_amdgcn_alignbyte(w[3][1], w[3][0], 2);  // ... one row on
    const uint32_t b4 = pack4(tap6p(pr(2, 0), pr(2, 1), pr(2, 2), pr(2, 3), pr(2, 4), pr(2, 5)),
                              tap6p(pr(2, 2), pr(2, 3), pr(2, 4), pr(2, 5), pr(2, 6), pr(2, 7)));
    const uint32_t s4 = pack4(tap6p(pr(3, 0), pr(3, 1), pr(3, 2), pr(3, 3), pr(3, 4), pr(3, 5)),
                              tap6p(pr(3, 2), pr(3, 3), pr(3, 4), pr(3, 5), pr(3, 6), pr(3, 7)));
    const uint32_t h4 = pack4(VP[2], VP[4]), m4 = pack4(VP[3], VP[5]);
    // the centre sample filters the CLIPPED vertical values (F/mocomp.cpp:71)
    const uint32_t j4 = pack4(tap6p(VP[0], VP[1], VP[2], VP[3], VP[4], VP[5]), tap6p(VP[2], VP[3], VP[4], VP[5], VP[6], VP[7]));
    const uint32_t o[16] = {G4,           mid4(G4, b4), b4,           mid4(b4, Gn4), mid4(G4, h4), mid4(b4, h4), mid4(b4, j4), mid4(b4, m4),
                            h4,           mid4(h4, j4), j4,           mid4(j4, m4),  mid4(h4, Gd4), mid4(h4, s4), mid4(j4, s4), mid4(s4, m4)};
    const size_t off = (size_t)y * d.ipitch + x;
#pragma unroll
    for (int f = 0; f < 16; f++) *(uint32_t *)(P + (size_t)f * d.iplane + off) = o[f];  // (streaming stores: 6.1 ms against 3.1)
}

// right and bottom margins of the 16 planes: pixel (x, y) beyond the picture = the plane's (min(x, W-1), min(y, H-1))
__global__ __launch_bounds__(256) void k_interp_pad(FerDev d)
{
    const int s = blockIdx.z, f = blockIdx.y;
    if (d.hdr[s * 4 + 3] != 0) return;
    uint8_t *P = d.interp + ((size_t)s * 16 + f) * d.iplane + d.ioff;
    const int W = d.W, H = d.H;
    const int nr = FER_IP_R * H, nb = (W + FER_IP_R) * FER_IP_B;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nr + nb; i += gridDim.x * blockDim.x) {
        int x, y;
        if (i < nr) {
            x = W + i % FER_IP_R;
            y = i / FER_IP_R;
        } else {
            x = (i - nr) % (W + FER_IP_R);
            y = H + (i - nr) / (W + FER_IP_R);
        }
        P[(size_t)y * d.ipitch + x] = P[(size_t)min(y, H - 1) * d.ipitch + min(x, W - 1)];
    }
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
    const uint8_t *P = d.interp + ((size_t)s * 16 + f) * d.iplane + d.ioff;
    const int y0 = strip * FS_ROWS, y1 = min(y0 + FS_ROWS, H);  // output rows [y0, y1)
    uint32_t *out = (uint32_t *)(d.feat + (size_t)s * 96 * d.ysz);
    const int sh = (int)((uintptr_t)(P + x) & 3);  // W is a multiple of 16: the same byte offset in every row
    const int nv = W - x;
    int h8[8], h4[8], hc[8];
    for (int base = 0; base < FS_ROWS + 8; base += 8) {
        if (y0 + base - 7 >= y1) break;
        // the 8 input rows of this round are requested together; a row at a time would leave the wavefront
        // waiting on one memory round trip per output row
        FeatRow rr[8];
#pragma unroll
        for (int j = 0; j < 8; j++) rr[j] = feat_row_load(P + (size_t)min(y0 + base + j, H - 1) * d.ipitch, x);
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
            }
        }
    }
}

// ------------------------------------------------------------------ k_feat0
// Plane 0 of the feature table straight from the reference luma (interpolated plane 0 IS the reference picture):
//   feat0 [H][W]   12-byte records (k0|k1, k2|k3, k4) row-major, what the wide integer search streams;
//   keyT  [W][H]   the sort key k0 alone (uint16) in ARRIVAL order b = tx*H + ty of the reference's counting sort
//                  (F/moestimation.cpp:142-151 scans columns): what the histogram of the first radix pass reads.
// The records of the sort itself are not written here: the first scatter pass rebuilds them from the samples of its
// own tile (k_rs_scatter<0>), which is cheaper than a 16-byte record per position written and read back once.
// One workgroup per 64x64 tile of positions: the (64+7)^2 samples (replicated beyond the picture, like the
// reference's 8-pixel padding :107-115) go to LDS once; a thread owns a column of 16 positions, takes the horizontal
// partial sums of each input row with v_sad_u8 and keeps the last 8 rows in registers (the scheme of k_features);
// the keys are transposed through LDS so that both outputs are written in long contiguous runs.
#define F0_T 64
__global__ __launch_bounds__(256) void k_feat0(FerDev d, uint16_t *keyT)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[F0_T + 7][F0_T + 8];
    __shared__ uint16_t stage[F0_T][F0_T + 2];  // [x][y]: + 2 breaks the bank stride
    const int s = blockIdx.z;
    if (d.hdr[s * 4 + 3] != 0) return;
    const int W = d.W, H = d.H;
    const uint8_t *R = d.refY + (size_t)s * d.ysz;
    const int x0 = blockIdx.x * F0_T, y0 = blockIdx.y * F0_T;
    const int tid = threadIdx.x;
    for (int i = tid; i < (F0_T + 7) * ((F0_T + 8) / 4); i += 256) {
        const int r = i / ((F0_T + 8) / 4), c4 = (i % ((F0_T + 8) / 4)) * 4;
        const int y = min(y0 + r, H - 1);
        uint32_t v;
        if (x0 + c4 + 3 < W) {
            v = *(const uint32_t *)(R + (size_t)y * W + x0 + c4);
        } else {
            v = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) v |= (uint32_t)R[(size_t)y * W + min(x0 + c4 + k, W - 1)] << (8 * k);
        }
        *(uint32_t *)&tile[r][c4] = v;
    }
    __syncthreads();
    const int x = tid & 63, g = tid >> 6;  // column of the tile, group of 16 rows
    const int sh = x & 3;
    int h8[8], h4[8], hc[8];
    int nzero = 0;
#pragma unroll
    for (int j = 0; j < 16 + 7; j++) {
        const int r = g * 16 + j;  // input row of the tile; completes the window of output row r - 7
        const uint32_t *w = (const uint32_t *)&tile[r][x & ~3];
        const uint32_t lo = __builtin_amdgcn_alignbyte(w[1], w[0], (uint32_t)sh);
        const uint32_t hi = __builtin_amdgcn_alignbyte(w[2], w[1], (uint32_t)sh);
        const int a4 = (int)__builtin_amdgcn_sad_u8(lo, 0u, 0u);
        h4[j & 7] = a4;
        h8[j & 7] = (int)__builtin_amdgcn_sad_u8(hi, 0u, (uint32_t)a4);
        hc[j & 7] = (int)__builtin_amdgcn_sad_u8(hi & 0xffffu, 0u, __builtin_amdgcn_sad_u8(lo & 0xffffu, 0u, 0u));
        if (j >= 7) {
            const int yo = r - 7;  // window rows yo .. yo+7 sit in slots (j + 1 + k) & 7
            int k0 = 0, k2 = 0, k4 = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                k0 += h8[k];
                k2 += h4[k];
                k4 += hc[k];
            }
            const int k1 = h8[(j + 1) & 7] + h8[(j + 2) & 7] + h8[(j + 3) & 7] + h8[(j + 4) & 7];
            const int k3 = h8[(j + 1) & 7] + h8[(j + 2) & 7] + h8[(j + 5) & 7] + h8[(j + 6) & 7];
            const uint32_t a = (uint32_t)k0 | ((uint32_t)k1 << 16), b = (uint32_t)k2 | ((uint32_t)k3 << 16), c = (uint32_t)k4;
            stage[x][yo] = (uint16_t)k0;
            const int gx = x0 + x, gy = y0 + yo;
            if (gx < W && gy < H) {
                uint32_t *o = (uint32_t *)(d.feat0 + ((size_t)s * d.ysz + (size_t)gy * W + gx) * 6);
                o[0] = a;
                o[1] = b;
                o[2] = c;
                nzero += k0 == 0;
            }
        }
    }
    if (__any(nzero)) {  // the reference mis-files sum 0: see k_sort_quirk
        const int tot = wave_sum(nzero);
        if ((tid & 63) == 0 && tot) atomicAdd(&d.zero_cnt[s], tot);
    }
    __syncthreads();
    // column-major keys: a pair of rows per thread, 32 threads = one column of the tile (H is even)
    for (int i = tid; i < F0_T * F0_T / 2; i += 256) {
        const int cx = i >> 5, cy = (i & 31) * 2;
        const int gx = x0 + cx, gy = y0 + cy;
        if (gx < W && gy < H)
            *(uint32_t *)(keyT + (size_t)s * d.ysz + (size_t)gx * H + gy) = (uint32_t)stage[cx][cy] | ((uint32_t)stage[cx][cy + 1] << 16);
    }
}

// ------------------------------------------------------------------ sort by 8x8 sum
// The order the reference's counting sort produces (F/moestimation.cpp:140-172) is "by sum, ties in arrival order",
// i.e. a stable sort of the arrival sequence by a 14-bit key (sums <= 64 * 255).  Two stable LSD radix passes of
// 7 bits; each pass is three launches -- per-tile digit histograms, one exclusive scan per stream over (digit, tile),
// stable scatter -- and no block ever waits on another one, so the sort keeps its speed when another context's
// kernels share the GPU.  The 16-byte records travel with their keys (no gather at the end): pass 1 reads them in
// arrival order, pass 2 writes the 12-byte records the bucket walk streams, the sorted positions and the sorted keys.
// Streams are independent segments (blockIdx.y).
// a tile = 4096 items = 64 KB of records in LDS, so two workgroups per CU: 512 threads each keep 16 wavefronts per CU in
// flight (256 threads x 16 items left 8: the scatter was waiting on memory 78 % of its cycles; measured 13.9 -> 11.1 ms
// per 256-stream picture, 1024 x 8 the same, 512 x 4 = 14.4)
#ifndef RS_THREADS
#define RS_THREADS 512
#endif
#ifndef RS_ITEMS
#define RS_ITEMS 8
#endif
#define RS_TILE (RS_THREADS * RS_ITEMS)
#define RS_BITS 7
#define RS_ND (1 << RS_BITS)

__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(FerDev d, const uint16_t *keyT, const uint8_t *dig2, uint32_t *hist, int ntiles,
                                                       int pass)
{
    __shared__ unsigned h[RS_ND];
    const int s = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    if (d.hdr[s * 4 + 3] != 0) return;
    const int n = d.W * d.H;  // (a multiple of 256: a thread's RS_ITEMS = 8 consecutive items are all inside or all outside)
    const size_t g0 = (size_t)s * n;
    if (tid < RS_ND) h[tid] = 0;
    static_assert(RS_ITEMS == 8, "one 16-byte / 8-byte load per thread");
    const int idx = tile * RS_TILE + tid * RS_ITEMS;
    const bool ok = idx < n;
    unsigned dgs[RS_ITEMS];
    if (pass == 0) {
        const uint4 v = ok ? *(const uint4 *)(keyT + g0 + idx) : make_uint4(0, 0, 0, 0);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < RS_ITEMS; i++) dgs[i] = (w[i >> 1] >> (16 * (i & 1))) & (RS_ND - 1);
    } else {
        const uint2 v = ok ? *(const uint2 *)(dig2 + g0 + idx) : make_uint2(0, 0);
        const uint32_t w[2] = {v.x, v.y};
#pragma unroll
        for (int i = 0; i < RS_ITEMS; i++) dgs[i] = (w[i >> 2] >> (8 * (i & 3))) & 0xffu;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        const unsigned dg = dgs[i];
        // flat content puts a whole wavefront on one digit: its lanes are counted with one atomic
        const unsigned d0 = (unsigned)__builtin_amdgcn_readfirstlane((int)dg);
        const unsigned long long same = __ballot(ok && dg == d0);
        if (same && (tid & 63) == __ffsll((long long)same) - 1) atomicAdd(&h[d0], (unsigned)__popcll(same));
        if (ok && dg != d0) atomicAdd(&h[dg], 1u);
    }
    __syncthreads();
    if (tid < RS_ND) hist[((size_t)s * RS_ND + tid) * ntiles + tile] = h[tid];  // digit-major: the scan order is the output order
}

// exclusive scan of one stream's (digit, tile) counts: sixteen wavefronts, each over a contiguous sixteenth of the table,
// four counts per lane and step (the table length is a multiple of RS_ND, so of four)
#define RSS_WAVES 16
__global__ __launch_bounds__(64 * RSS_WAVES) void k_rs_scan(FerDev d, uint32_t *hist, int ntiles)
{
    __shared__ unsigned wtot[RSS_WAVES];
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (d.hdr[s * 4 + 3] != 0) return;
    uint4 *h4 = (uint4 *)(hist + (size_t)s * RS_ND * ntiles);
    const int n4 = RS_ND * ntiles / 4;
    const int per = (n4 + RSS_WAVES - 1) / RSS_WAVES;
    const int u0 = min(wv * per, n4), u1 = min(u0 + per, n4);
    unsigned sum = 0;
    for (int u = u0 + lane; u < u1; u += 64) {
        const uint4 v = h4[u];
        sum += v.x + v.y + v.z + v.w;
    }
    sum = (unsigned)wave_sum((int)sum);
    if (lane == 0) wtot[wv] = sum;
    __syncthreads();
    unsigned carry = 0;
    for (int w = 0; w < wv; w++) carry += wtot[w];
    for (int ub = u0; ub < u1; ub += 64) {
        const int u = ub + lane;
        const uint4 v = u < u1 ? h4[u] : make_uint4(0, 0, 0, 0);
        const unsigned t = v.x + v.y + v.z + v.w;
        unsigned inc = t;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned x = __shfl_up(inc, o);
            if (lane >= o) inc += x;
        }
        const unsigned ex = carry + inc - t;
        if (u < u1) h4[u] = make_uint4(ex, ex + v.x, ex + v.x + v.y, ex + v.x + v.y + v.z);
        carry += __shfl(inc, 63);
    }
}

// Stable scatter of one tile.  A wavefront owns a contiguous quarter of the tile and walks it in arrival order,
// 64 items per round: the rank of an item among the earlier items of its digit comes from ballot matching plus a
// per-wavefront running count per digit.  The tile is then put in output order inside LDS (digit runs one after the
// other) and written out, so that neighbouring threads write neighbouring addresses of the same digit run.
// pass 0: in = the reference luma (records rebuilt per tile, below), out = rec1 (16 B) + dig2 (high digit);
// pass 1: in = rec1, out = final arrays.
//
// Pass 0 builds its 4096 records (k0|k1, k2|k3, k4, tx<<16|ty -- the values k_feat0 writes row-major) itself.  The tile
// is a run of arrival indices b = tx*H + ty, i.e. nc column segments (tx, rows [ys, ye)); rows (segment, r) for
// r in [ys, ye + 7) are numbered q = 0 .. Q-1 one segment after the other (u = q + ya = c*(H+7) + r).
//   1. the samples those rows need go to LDS -- one box of columns [xa, xb + 7] x all rows when the tile spans three
//      or more columns, one 12-byte-wide box per segment otherwise (tall pictures) --, replicated beyond the picture;
//   2. a thread takes a run of q's: horizontal sums (h8 | h4 << 16, hc in the high dword: one u64 per row), summed
//      into an exclusive prefix PL over the whole q sequence (an exact integer, 55 bits at most: a difference of two
//      entries of one segment is the three window sums side by side, 16 bits apart);
//   3. an item reads PL at its window's rows 0, 2, 4, 6, 8.
// Samples, PL and the output staging share the same 64 KB of LDS, one after the other.
#define RS_SEG_PITCH 12
__device__ __forceinline__ void rs_divmod(unsigned v, unsigned dv, float rdv, unsigned &qo, unsigned &rem)
{  // v < 2^28, dv <= 2^15: the float quotient is off by one at most
    int q = (int)((float)v * rdv);
    int r = (int)v - q * (int)dv;
    if (r < 0) {
        q--;
        r += (int)dv;
    }
    if (r >= (int)dv) {
        q++;
        r -= (int)dv;
    }
    qo = (unsigned)q;
    rem = (unsigned)r;
}

template <int PASS>
__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(FerDev d, const uint8_t *dig2_in, const uint4 *rec_in, uint4 *rec1_out,
                                                          uint8_t *dig2_out, uint32_t *rec_tmp, uint16_t *skey, const uint32_t *hist,
                                                          int ntiles)
{
    __shared__ unsigned run[RS_THREADS / 64][RS_ND];  // per wavefront: items of each digit seen so far
    __shared__ unsigned base[RS_ND], dstart[RS_ND];
    __shared__ unsigned wsum[RS_THREADS / 64];
    __shared__ unsigned long long wtot[RS_THREADS / 64];
    __shared__ __attribute__((aligned(16))) uint4 srec[RS_TILE];
    const int s = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    if (d.hdr[s * 4 + 3] != 0) return;
    const int lane = tid & 63, wv = tid >> 6;
    const int n = d.W * d.H;
    const size_t g0 = (size_t)s * n;
    if (tid < RS_ND) {
#pragma unroll
        for (int w = 0; w < RS_THREADS / 64; w++) run[w][tid] = 0;
        base[tid] = hist[((size_t)s * RS_ND + tid) * ntiles + tile];
    }
    const int w0 = tile * RS_TILE + wv * (RS_TILE / (RS_THREADS / 64));
    unsigned dg[RS_ITEMS], rk[RS_ITEMS];
    uint4 rec[PASS == 0 ? RS_ITEMS : 1];
    if (PASS == 0) {
        const int W = d.W, H = d.H;
        const uint8_t *R = d.refY + (size_t)s * d.ysz;
        uint8_t *smp = (uint8_t *)srec;
        unsigned long long *PL = (unsigned long long *)srec;
        const int t0 = tile * RS_TILE, cnt = min(RS_TILE, n - t0);
        const int xa = t0 / H, ya = t0 - xa * H, xb = (t0 + cnt - 1) / H;
        const int nc = xb - xa + 1, Q = cnt + 7 * nc;
        const int HR = H + 7;
        const float rHR = 1.0f / (float)HR, rH = 1.0f / (float)H;
        const bool shared_box = nc >= 3;
        // sample (segment c, picture row r, first column of the segment) sits at smp[segb(c) + r * pitch]; an odd number of
        // dwords per row and an odd number of rows per thread (below) keep the lanes of a wavefront on different LDS banks
        const int nd0 = ((xb + 7) >> 2) - (xa >> 2) + 1;
        const int nd = shared_box ? (((nd0 | 1) * 4 * HR <= (int)sizeof(srec)) ? (nd0 | 1) : nd0) : RS_SEG_PITCH / 4;
        const int pitch = nd * 4;
        const int rows0 = shared_box ? HR : (nc == 1 ? cnt + 7 : HR - ya);  // rows of the first box
        // 1. samples: a thread fetches whole rows of a narrow box (two rows = up to 14 dwords in flight), single dwords of
        //    a wide one (pictures of a few rows)
        {
            const int nboxes = shared_box ? 1 : nc;
            for (int bx = 0; bx < nboxes; bx++) {
                const int nr = bx == 0 ? rows0 : (t0 + cnt - xb * H) + 7;  // the last segment starts at row 0
                const int xs = (xa + bx) & ~3, r0 = (shared_box || bx) ? 0 : ya;
                uint8_t *dst = smp + (bx ? rows0 * RS_SEG_PITCH : 0);
                if (nd <= 7) {
                    const bool inside = xs + nd * 4 <= W;
                    for (int ra = tid; ra < nr; ra += 2 * RS_THREADS) {
                        uint32_t v[2][7];
#pragma unroll
                        for (int k = 0; k < 2; k++) {
                            const int r = ra + k * RS_THREADS;
                            const uint8_t *row = R + (size_t)min(r0 + min(r, nr - 1), H - 1) * W;
                            const uint32_t last = inside ? 0u : (uint32_t)row[W - 1] * 0x01010101u;
#pragma unroll
                            for (int j = 0; j < 7; j++)
                                if (j < nd) v[k][j] = (inside || xs + j * 4 < W) ? *(const uint32_t *)(row + xs + j * 4) : last;
                        }
#pragma unroll
                        for (int k = 0; k < 2; k++) {
                            const int r = ra + k * RS_THREADS;
                            if (r < nr) {
#pragma unroll
                                for (int j = 0; j < 7; j++)
                                    if (j < nd) *(uint32_t *)(dst + r * pitch + j * 4) = v[k][j];
                            }
                        }
                    }
                } else {
                    const float rnd = 1.0f / (float)nd;
                    for (int i0 = 0; i0 < nr * nd; i0 += 8 * RS_THREADS) {
                        uint32_t v[8];
                        int at[8];
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            const int i = i0 + k * RS_THREADS + tid;
                            at[k] = -1;
                            if (i < nr * nd) {
                                unsigned r, c4;
                                rs_divmod((unsigned)i, (unsigned)nd, rnd, r, c4);
                                const int gx = xs + (int)c4 * 4, gy = min(r0 + (int)r, H - 1);
                                const uint8_t *row = R + (size_t)gy * W;
                                v[k] = gx < W ? *(const uint32_t *)(row + gx) : (uint32_t)row[W - 1] * 0x01010101u;
                                at[k] = (int)r * pitch + (int)c4 * 4;
                            }
                        }
#pragma unroll
                        for (int k = 0; k < 8; k++)
                            if (at[k] >= 0) *(uint32_t *)(dst + at[k]) = v[k];
                    }
                }
            }
        }
        __syncthreads();
        // 2. horizontal sums of a run of rows, prefix
        const int L = ((Q + RS_THREADS - 1) / RS_THREADS) | 1;  // <= 13 (H = 16: 256 segments)
        const int q0 = min(tid * L, Q), q1 = min(q0 + L, Q);
        unsigned long long hv[13];
        unsigned long long tot = 0;
        {
            unsigned c, r;
            rs_divmod((unsigned)(q0 + ya), (unsigned)HR, rHR, c, r);
#pragma unroll
            for (int k = 0; k < 13; k++) {
                hv[k] = 0;
                if (q0 + k < q1) {
                    int a;
                    if (shared_box)
                        a = (int)r * pitch + (xa & 3) + (int)c;
                    else
                        a = (c ? rows0 * RS_SEG_PITCH + (int)r * RS_SEG_PITCH : ((int)r - ya) * RS_SEG_PITCH) + ((xa + (int)c) & 3);
                    const uint32_t *w = (const uint32_t *)(smp + (a & ~3));
                    const uint32_t sh = (uint32_t)(a & 3);
                    const uint32_t lo = __builtin_amdgcn_alignbyte(w[1], w[0], sh);
                    const uint32_t hi = __builtin_amdgcn_alignbyte(w[2], w[1], sh);
                    const uint32_t a4 = __builtin_amdgcn_sad_u8(lo, 0u, 0u);
                    const uint32_t a8 = __builtin_amdgcn_sad_u8(hi, 0u, a4);
                    const uint32_t ac = __builtin_amdgcn_sad_u8(hi & 0xffffu, 0u, __builtin_amdgcn_sad_u8(lo & 0xffffu, 0u, 0u));
                    hv[k] = ((unsigned long long)ac << 32) | (a8 | (a4 << 16));
                    tot += hv[k];
                    if (++r == (unsigned)HR) {
                        r = 0;
                        c++;
                    }
                }
            }
        }
        unsigned long long inc = tot;  // inclusive scan of the thread totals over the workgroup
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned lo = (unsigned)__shfl_up((int)(unsigned)inc, o), hi = (unsigned)__shfl_up((int)(unsigned)(inc >> 32), o);
            if (lane >= o) inc += ((unsigned long long)hi << 32) | lo;
        }
        if (lane == 63) wtot[wv] = inc;
        __syncthreads();  // (also: every thread has read its samples)
        unsigned long long pre = inc - tot;
        for (int w = 0; w < wv; w++) pre += wtot[w];
#pragma unroll
        for (int k = 0; k < 13; k++)
            if (q0 + k < q1) {
                PL[q0 + k] = pre;
                pre += hv[k];
            }
        if (q0 < Q && q1 == Q) PL[Q] = pre;
        __syncthreads();
        // 3. the items of this thread
        unsigned x = 0, y = 0;
#pragma unroll
        for (int r = 0; r < RS_ITEMS; r++) {
            const int idx = w0 + r * 64 + lane;
            dg[r] = 0;
            rec[r] = make_uint4(0, 0, 0, 0);
            if (r == 0 || H < 64) {
                rs_divmod((unsigned)min(idx, n - 1), (unsigned)H, rH, x, y);
            } else {  // 64 positions on: at most one column further
                y += 64;
                if (y >= (unsigned)H) {
                    y -= (unsigned)H;
                    x++;
                }
            }
            if (idx < n) {
                const int c = (int)x - xa;
                const int q = c * HR + (int)y - ya;  // row y of segment c
                const unsigned long long p0 = PL[q], p8 = PL[q + 8];
                const uint32_t *pl = (const uint32_t *)PL;
                const uint32_t l0 = (uint32_t)p0, l2 = pl[(q + 2) * 2], l4 = pl[(q + 4) * 2], l6 = pl[(q + 6) * 2];
                const unsigned long long dd = p8 - p0;
                const uint32_t k0 = (uint32_t)dd & 0xffffu, k2 = (uint32_t)(dd >> 16) & 0xffffu, k4 = (uint32_t)(dd >> 32) & 0xffffu;
                const uint32_t k1 = (l4 - l0) & 0xffffu, k3 = ((l2 - l0) + (l6 - l4)) & 0xffffu;
                rec[r] = make_uint4(k0 | (k1 << 16), k2 | (k3 << 16), k4, (x << 16) | y);
                dg[r] = k0 & (RS_ND - 1);
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < RS_ITEMS; r++) {
            const int idx = w0 + r * 64 + lane;
            dg[r] = idx < n ? (unsigned)dig2_in[g0 + idx] : 0u;
        }
    }
    __syncthreads();
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        const bool ok = w0 + r * 64 + lane < n;
        // lanes with the same digit: those whose ballot of every digit bit agrees with this lane's bit
        // (differences OR-ed per 32-bit half: a v_xor per bit and half, a three-input OR per two bits)
        unsigned long long peers;
        {
            uint32_t dlo = 0, dhi = 0;
#pragma unroll
            for (int b = 0; b < RS_BITS; b++) {
                const unsigned long long bal = __ballot((dg[r] >> b) & 1);
                const uint32_t mine = 0u - ((dg[r] >> b) & 1u);
                dlo |= (uint32_t)bal ^ mine;
                dhi |= (uint32_t)(bal >> 32) ^ mine;
            }
            peers = __ballot(ok) & ~(((unsigned long long)dhi << 32) | dlo);
        }
        const unsigned before = run[wv][dg[r]];
        rk[r] = before + (unsigned)__popcll(peers & lt);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (ok && (peers & lt) == 0) run[wv][dg[r]] = before + (unsigned)__popcll(peers);  // first lane of the group
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    unsigned tot = 0;
    if (tid < RS_ND) {  // digit tid: wavefront totals -> offsets inside the digit's run of this tile
#pragma unroll
        for (int w = 0; w < RS_THREADS / 64; w++) {
            unsigned c = run[w][tid];
            run[w][tid] = tot;
            tot += c;
        }
    }
    {  // exclusive prefix of the tile's digit totals (RS_ND values, threads 0 .. RS_ND-1 = wavefronts 0 and 1)
        unsigned inc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            unsigned v = __shfl_up(inc, o);
            if (lane >= o) inc += v;
        }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        unsigned before = 0;
        for (int w = 0; w < wv; w++) before += wsum[w];
        if (tid < RS_ND) dstart[tid] = before + inc - tot;
    }
    __syncthreads();  // (pass 0: the prefix table in srec's place is dead)
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        const int idx = w0 + r * 64 + lane;
        if (idx < n) srec[dstart[dg[r]] + run[wv][dg[r]] + rk[r]] = PASS == 0 ? rec[r] : rec_in[g0 + idx];
    }
    __syncthreads();
    const int cnt = min(RS_TILE, n - tile * RS_TILE);
    if (PASS == 0) {
        for (int i = tid; i < cnt; i += RS_THREADS) {
            const uint4 v = srec[i];
            const unsigned key = v.x & 0xffffu;
            const unsigned dgi = key & (RS_ND - 1);
            const size_t pos = g0 + min(base[dgi] + ((unsigned)i - dstart[dgi]), (unsigned)n - 1u);  // (in range by construction)
            rec1_out[pos] = v;
            dig2_out[pos] = (uint8_t)(key >> RS_BITS);
        }
    } else {
        // a stream with positions of sum 0 gets its final layout from k_sort_quirk: leave the persistent array alone
        uint32_t *recs = d.zero_cnt[s] > 0 ? rec_tmp : d.sort_rec;
        for (int i = tid; i < cnt; i += RS_THREADS) {
            const uint4 v = srec[i];
            const unsigned key = v.x & 0xffffu;
            const unsigned dgi = key >> RS_BITS;
            const size_t pos = g0 + base[dgi] + ((unsigned)i - dstart[dgi]);
            uint32_t *o = recs + pos * 3;  // {(tx << 16) | ty, k1 | k2 << 16, k3 | k4 << 16}: what the bucket walk reads
            o[0] = v.w;
            o[1] = (v.x >> 16) | (v.y << 16);
            o[2] = (v.y >> 16) | (v.z << 16);
            d.sort_pos[pos] = v.w;
            skey[pos] = (uint16_t)key;
        }
    }
}

// Two-level bucket index over the plain sorted order.  Record i opens every (sum, column tile) bin after its
// predecessor's up to its own: kol2[bin] = i for those bins (lower bound of the bin in the sorted order).  Gaps are
// short except at the ends of the sum range; long ones are filled by the whole wavefront.  A thread takes SI_ITEMS
// records (256 apart) and requests all of them before it looks at any: with one record per thread the kernel was one
// memory round trip per wavefront and nothing else (2.5 ms per 256-stream picture for 7 GB).
#define SI_ITEMS 4
__global__ __launch_bounds__(256) void k_sort_index(FerDev d, const uint16_t *skey)
{
    const int s = blockIdx.y;
    if (d.hdr[s * 4 + 3] != 0) return;
    const int n = d.W * d.H;
    const int nb = 16384 * d.kt;
    uint32_t *kol2 = d.kol2 + (size_t)s * nb;
    const size_t g0 = (size_t)s * n;
    const int lane = threadIdx.x & 63;
    int key[SI_ITEMS], pkey[SI_ITEMS], fkey[SI_ITEMS];
    uint32_t pos[SI_ITEMS], ppos[SI_ITEMS];
#pragma unroll
    for (int k = 0; k < SI_ITEMS; k++) {
        const int i = (blockIdx.x * SI_ITEMS + k) * 256 + (int)threadIdx.x;
        key[k] = pkey[k] = fkey[k] = -1;
        pos[k] = ppos[k] = 0;
        if (i < n) {
            key[k] = (int)skey[g0 + i];
            pos[k] = d.sort_pos[g0 + i];
            if (lane == 0 && i > 0) {  // (the other lanes take the predecessor from their neighbour)
                pkey[k] = (int)skey[g0 + i - 1];
                ppos[k] = d.sort_pos[g0 + i - 1];
            }
            if (i + FER_BRANGE_MIN < n) fkey[k] = (int)skey[g0 + i + FER_BRANGE_MIN];
        }
    }
#pragma unroll
    for (int k = 0; k < SI_ITEMS; k++) {
        const int i = (blockIdx.x * SI_ITEMS + k) * 256 + (int)threadIdx.x;
        int gap_lo = 0, gap_hi = 0;  // bins [gap_lo, gap_hi) get value gval
        uint32_t gval = 0;
        const int nkey = __shfl_up(key[k], 1);
        const uint32_t npos = (uint32_t)__shfl_up((int)pos[k], 1);
        if (i < n) {
            const int pk = lane ? nkey : pkey[k];
            const uint32_t pp = lane ? npos : ppos[k];
            const int bin = key[k] * d.kt + (int)((pos[k] >> 16) >> d.ktw_shift);
            const int prev = i > 0 ? pk * d.kt + (int)((pp >> 16) >> d.ktw_shift) : -1;
            gap_lo = prev + 1;
            gap_hi = bin + 1;
            gval = (uint32_t)(g0 + i);
            // the first record of a bucket clears what the big-bucket kernels accumulate for it (ranges, modal class), and a
            // record whose key is still the same 1024 places on says "this stream has big buckets" for this picture
            if (i == 0 || pk != key[k]) {
                uint4 *br = (uint4 *)(d.brange + ((size_t)s * 16384 + key[k]) * 8);
                br[0] = make_uint4(0, 0, 0, 0);
                br[1] = make_uint4(0, 0, 0, 0);
                *(uint4 *)(d.bmodal + ((size_t)s * 16384 + key[k]) * 4) = make_uint4(0, 0, 0, 0);
            }
            if (fkey[k] == key[k]) d.nbig[s] = d.serial;
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
                for (int b = lo + lane; b < hi; b += 64) kol2[b] = val;
            }
        }
    }
}

// The big buckets (more than FER_BRANGE_MIN positions: flat areas), one thread per sorted record, for the streams k_sort_index
// found any in:
//  * brange: the ranges of the other four sums over the bucket's positions (k_me_walk bounds the feature distance of a crowded
//    partition's candidates with them where no better description applies).  A wavefront usually sits inside one bucket and
//    folds its 64 records into one set of atomics, which it skips when the range already covers them.
//  * bmodal / boutl: the bucket's modal class = the feature dwords of its middle record, the number of records that differ
//    from it, and their sorted-array indices (list number = the bucket's first place / FER_BRANGE_MIN: unique, since every
//    big bucket has more places than that).  A wavefront counts its outliers with one atomic per bucket it touches and stops
//    adding once the bucket is known to have too many.
// Both kernels of this kind below usually have nothing to do for a stream (no big bucket, no sum-0 position): a stream
// gets SF_BLOCKS workgroups that loop over its records, not one workgroup per 256 records that exits at once (the empty
// launches alone took 1.7 ms per 256-stream picture).
#define SF_BLOCKS 64
__device__ __forceinline__ void bucket_classes_records(const FerDev &d, const uint16_t *skey, int s, int i)
{
    const int n = d.W * d.H;
    const size_t g0 = (size_t)s * n;
    const uint32_t *kol2 = d.kol2 + (size_t)s * 16384 * d.kt;
    const bool in = i < n;
    const int key = in ? (int)skey[g0 + i] : -1;
    uint32_t bs = 0, be = 0;
    if (in) {
        bs = kol2[(size_t)key * d.kt];
        be = kol2[(size_t)(key + 1) * d.kt];
    }
    const bool on = in && be - bs > FER_BRANGE_MIN;
    if (!__any(on)) return;
    const int lane = threadIdx.x & 63;
    uint32_t v[4] = {0, 0, 0, 0};
    bool outl = false;
    if (on) {
        const uint32_t *m = d.sort_rec + (size_t)(bs + (be - bs) / 2) * 3, *r = d.sort_rec + (g0 + i) * 3;
        const uint32_t r1 = r[1], r2 = r[2], m1 = m[1], m2 = m[2];
        v[0] = r1 & 0xffffu;
        v[1] = r1 >> 16;
        v[2] = r2 & 0xffffu;
        v[3] = r2 >> 16;
        outl = r1 != m1 || r2 != m2;
        if (g0 + i == bs) {
            uint32_t *bm = d.bmodal + ((size_t)s * 16384 + key) * 4;
            bm[0] = m1;
            bm[1] = m2;
        }
    }
    unsigned long long todo = __ballot(on);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        const int kcur = __builtin_amdgcn_readlane(key, src);
        const bool mine = on && key == kcur;
        todo &= ~__ballot(mine);
        uint32_t *br = d.brange + ((size_t)s * 16384 + (size_t)kcur) * 8;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int hi = wave_max(mine ? (int)v[k] : 0), lo = wave_max(mine ? 65535 - (int)v[k] : 0);
            if (lane == 0) {
                if ((uint32_t)hi > __hip_atomic_load(&br[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&br[k], (uint32_t)hi);
                if ((uint32_t)lo > __hip_atomic_load(&br[4 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&br[4 + k], (uint32_t)lo);
            }
        }
        const unsigned long long mo = __ballot(mine && outl);
        if (mo) {
            uint32_t *bm = d.bmodal + ((size_t)s * 16384 + kcur) * 4;
            const uint32_t list = ((uint32_t)__builtin_amdgcn_readlane((int)bs, src) - (uint32_t)g0) / FER_BRANGE_MIN;
            uint32_t base = FER_OUTL + 1;
            if (lane == 0 && __hip_atomic_load(bm + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= FER_OUTL) base = atomicAdd(bm + 2, (uint32_t)__popcll(mo));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (mine && outl) {
                const uint32_t k = base + (uint32_t)__popcll(mo & ((1ull << lane) - 1ull));
                if (k < FER_OUTL) d.boutl[((size_t)s * d.nlists + list) * FER_OUTL + k] = (uint32_t)(g0 + i);
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_bucket_classes(FerDev d, const uint16_t *skey)
{
    const int s = blockIdx.y;
    if (d.nbig[s] != d.serial) return;  // (k_sort_index of this picture)
    if (d.hdr[s * 4 + 3] != 0 || d.zero_cnt[s] != 0) return;  // (a stream with sum-0 positions: its crowded partitions take the exact slow path)
    const int nblk = (d.W * d.H + 255) / 256;
    for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) bucket_classes_records(d, skey, s, blk * 256 + (int)threadIdx.x);
}

// Bucket 0.  The reference's counting sort (F/moestimation.cpp:153-172) leaves bucket 0 out of its prefix sum: with
// n0 positions of sum 0, every other bucket starts n0 places early (the sorted array is the other positions from 0,
// its last n0 places keep what the previous picture left there), the k-th sum-0 position is written to place n0 + k,
// where it replaces, or is replaced by, the regular occupant -- whichever the scatter loop reaches later in arrival
// order --, bucket 0 reads as [0, 2 n0) and bucket 1 starts at 2 n0.  That layout is built here, place by place,
// from the plain sorted records (rec_tmp) into the persistent array for a stream with n0 > 0 (black areas in
// full-range content); the walk then scans whole buckets by these rules (walk_buckets).  The index kol2 keeps
// describing the plain sorted order.
__global__ __launch_bounds__(256) void k_sort_quirk(FerDev d, const uint32_t *rec_tmp)
{
    const int s = blockIdx.y;
    if (d.hdr[s * 4 + 3] != 0) return;
    const int n0 = d.zero_cnt[s];
    if (n0 == 0) return;
    const int n = d.W * d.H;
    const size_t g0 = (size_t)s * n;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {  // place of the final array
        const int ir = p + n0;   // the regular position that lands here (sorted index), if any
        const int iz = p - n0;   // the sum-0 position aimed here, if any
        const bool hr = ir < n, hz = iz >= 0 && iz < n0;
        int src = -1;
        if (hr && hz) {
            const uint32_t vr = d.sort_pos[g0 + ir], vz = d.sort_pos[g0 + iz];
            const int br = (int)(vr >> 16) * d.H + (int)(vr & 0xffff), bz = (int)(vz >> 16) * d.H + (int)(vz & 0xffff);
            src = bz > br ? iz : ir;
        } else if (hr) {
            src = ir;
        } else if (hz) {
            src = iz;
        }
        if (src < 0) continue;  // keeps the previous picture's entry
        const uint32_t *in = rec_tmp + (g0 + src) * 3;
        uint32_t *o = d.sort_rec + (g0 + p) * 3;
        o[0] = in[0];
        o[1] = in[1];
        o[2] = in[2];
    }
}

size_t fer_sort_tmp_bytes(int n, int S)
{
    const int ntiles = (n + RS_TILE - 1) / RS_TILE;
    return (size_t)S * RS_ND * ntiles * sizeof(uint32_t);
}

// host side: prepare the reference structures of all P-picture streams (profiled steps)
void fer_launch_interp(const FerDev &d, hipStream_t st)
{
    dim3 gi(((d.W + IT_W - 1) / IT_W) * ((d.H + IT_H - 1) / IT_H), d.S);
    hipLaunchKernelGGL(k_interp, gi, dim3(256), 0, st, d);
    hipLaunchKernelGGL(k_interp_pad, dim3(8, 16, d.S), dim3(256), 0, st, d);
}

void fer_launch_features(const FerDev &d, hipStream_t st)
{
    long long fw = (long long)(d.W >> 2) * ((d.H + FS_ROWS - 1) / FS_ROWS) * d.S;
    hipLaunchKernelGGL(k_features, dim3((unsigned)((fw + 3) / 4)), dim3(256), 0, st, d);
}

// plane-0 features + the sort input (profiled as "sort_keys")
void fer_launch_sort_keys(const FerDev &d, FerSortTmp &t, hipStream_t st)
{
    hipMemsetAsync(d.zero_cnt, 0, sizeof(int) * d.S, st);
    dim3 g((d.W + F0_T - 1) / F0_T, (d.H + F0_T - 1) / F0_T, d.S);
    hipLaunchKernelGGL(k_feat0, g, dim3(256), 0, st, d, t.keyT);
}

void fer_launch_sort_radix(const FerDev &d, FerSortTmp &t, hipStream_t st)
{
    const int n = d.W * d.H;
    const int ntiles = (n + RS_TILE - 1) / RS_TILE;
    uint32_t *hist = (uint32_t *)t.tmp;
    for (int pass = 0; pass < 2; pass++) {  // sum bits 0-6, then 7-13
        hipLaunchKernelGGL(k_rs_hist, dim3(ntiles, d.S), dim3(RS_THREADS), 0, st, d, t.keyT, t.dig2, hist, ntiles, pass);
        hipLaunchKernelGGL(k_rs_scan, dim3(d.S), dim3(64 * RSS_WAVES), 0, st, d, hist, ntiles);
        if (pass == 0)
            hipLaunchKernelGGL(k_rs_scatter<0>, dim3(ntiles, d.S), dim3(RS_THREADS), 0, st, d, t.dig2, t.rec1, t.rec1, t.dig2, t.rec_tmp, t.skey,
                               hist, ntiles);
        else
            hipLaunchKernelGGL(k_rs_scatter<1>, dim3(ntiles, d.S), dim3(RS_THREADS), 0, st, d, t.dig2, t.rec1, t.rec1, t.dig2, t.rec_tmp, t.skey,
                               hist, ntiles);
    }
}

// bucket index (+ the reference's mis-filed layout for streams with sum-0 positions)
void fer_launch_sort_finish(const FerDev &d, FerSortTmp &t, hipStream_t st)
{
    const int n = d.W * d.H;
    hipLaunchKernelGGL(k_sort_index, dim3((n + 256 * SI_ITEMS - 1) / (256 * SI_ITEMS), d.S), dim3(256), 0, st, d, t.skey);
    const int sfb = std::min((n + 255) / 256, SF_BLOCKS);
    hipLaunchKernelGGL(k_bucket_classes, dim3(sfb, d.S), dim3(256), 0, st, d, t.skey);
    hipLaunchKernelGGL(k_sort_quirk, dim3(sfb, d.S), dim3(256), 0, st, d, t.rec_tmp);
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
    fer_launch_sort(d, t, st);
}

// ------------------------------------------------------------------ k_frame_sad
// selectNALUnitType's whole-picture SAD (F/ref_frames.cpp:185-234): a pure stream of 2 bytes per luma sample, read as
// 16-byte words, four of them in flight per lane
__global__ __launch_bounds__(256) void k_frame_sad(FerDev d)
{
    const int s = blockIdx.y;
    const uint4 *a = (const uint4 *)(d.curY + (size_t)s * d.ysz), *b = (const uint4 *)(d.refY + (size_t)s * d.ysz);
    const size_t n16 = d.ysz / 16, stride = (size_t)gridDim.x * blockDim.x;  // the luma plane is a multiple of 256 bytes
    unsigned acc0 = 0, acc1 = 0;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + stride < n16; i += 2 * stride) {
        const uint4 va = a[i], vb = b[i], wa = a[i + stride], wb = b[i + stride];
        acc0 = __builtin_amdgcn_sad_u8(va.x, vb.x, acc0);
        acc1 = __builtin_amdgcn_sad_u8(va.y, vb.y, acc1);
        acc0 = __builtin_amdgcn_sad_u8(va.z, vb.z, acc0);
        acc1 = __builtin_amdgcn_sad_u8(va.w, vb.w, acc1);
        acc0 = __builtin_amdgcn_sad_u8(wa.x, wb.x, acc0);
        acc1 = __builtin_amdgcn_sad_u8(wa.y, wb.y, acc1);
        acc0 = __builtin_amdgcn_sad_u8(wa.z, wb.z, acc0);
        acc1 = __builtin_amdgcn_sad_u8(wa.w, wb.w, acc1);
    }
    if (i < n16) {
        const uint4 va = a[i], vb = b[i];
        acc0 = __builtin_amdgcn_sad_u8(va.x, vb.x, acc0);
        acc1 = __builtin_amdgcn_sad_u8(va.y, vb.y, acc1);
        acc0 = __builtin_amdgcn_sad_u8(va.z, vb.z, acc0);
        acc1 = __builtin_amdgcn_sad_u8(va.w, vb.w, acc1);
    }
    int v = wave_sum((int)(acc0 + acc1));
    if ((threadIdx.x & 63) == 0) atomicAdd(&d.sad[s], (unsigned long long)(unsigned)v);
}

void fer_launch_frame_sad(const FerDev &d, hipStream_t st)
{
    hipMemsetAsync(d.sad, 0, sizeof(unsigned long long) * d.S, st);
    const int gx = (int)std::min<size_t>(64, std::max<size_t>(1, d.ysz / 16 / (256 * 8)));  // ~8 words of 16 bytes per lane
    hipLaunchKernelGGL(k_frame_sad, dim3(gx, d.S), dim3(256), 0, st, d);
}
