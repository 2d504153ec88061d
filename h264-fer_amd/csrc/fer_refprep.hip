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
#include <rocprim/rocprim.hpp>

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
// One block = one 32x8 tile of positions, all 16 planes.  Per plane: separable box sums in LDS
// (horizontal partial sums of every needed row, then 8 / 4 rows added per output).  Positions
// beyond the picture replicate the last row/column (the reference pads its integral image by
// 8).  Results are staged in LDS as one 192-byte record per position -- [frac][k0..k4,pad] --
// and written with 16-byte stores, so a searcher gets the five features of one (position, frac)
// with a single 12-byte load and the 16 fracs of neighbouring positions from contiguous lines.
#define FT_W 32
#define FT_H 8
#define FT_REC 96  // uint16 per position: 16 fracs x 6
__global__ __launch_bounds__(256) void k_features(FerDev d)
{
    __shared__ uint8_t tile[FT_H + 7][FT_W + 8];
    __shared__ uint16_t h8[FT_H + 7][FT_W], h4[FT_H + 7][FT_W], hc[FT_H + 7][FT_W];
    __shared__ __attribute__((aligned(16))) uint16_t rec[FT_W * FT_H][FT_REC];
    const int s = blockIdx.z;
    if (d.hdr[s * 4 + 3] != 0) return;
    const int x0 = blockIdx.x * FT_W, y0 = blockIdx.y * FT_H;
    const int tid = threadIdx.x;
    const int tx = tid & (FT_W - 1), ty = tid / FT_W;
    for (int f = 0; f < 16; f++) {
        const uint8_t *P = d.interp + ((size_t)s * 16 + f) * d.ysz;
        __syncthreads();
        for (int i = tid; i < (FT_H + 7) * (FT_W + 7); i += 256) {
            int r = i / (FT_W + 7), c = i % (FT_W + 7);
            int sx = min(x0 + c, d.W - 1), sy = min(y0 + r, d.H - 1);
            tile[r][c] = P[(size_t)sy * d.W + sx];
        }
        __syncthreads();
        for (int i = tid; i < (FT_H + 7) * FT_W; i += 256) {
            int r = i / FT_W, c = i % FT_W;
            const uint8_t *q = &tile[r][c];
            int a = q[0] + q[1], b = q[2] + q[3], cc = q[4] + q[5], e = q[6] + q[7];
            h8[r][c] = (uint16_t)(a + b + cc + e);
            h4[r][c] = (uint16_t)(a + b);
            hc[r][c] = (uint16_t)(a + cc);
        }
        __syncthreads();
        int k0 = 0, k1 = 0, k2 = 0, k3 = 0, k4 = 0;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            int v8 = h8[ty + r][tx];
            k0 += v8;
            if (r < 4) k1 += v8;
            k2 += h4[ty + r][tx];
            if ((r & 3) < 2) k3 += v8;
            k4 += hc[ty + r][tx];
        }
        uint16_t *o = &rec[tid][f * 6];
        o[0] = (uint16_t)k0;
        o[1] = (uint16_t)k1;
        o[2] = (uint16_t)k2;
        o[3] = (uint16_t)k3;
        o[4] = (uint16_t)k4;
        o[5] = 0;
    }
    __syncthreads();
    // plane-0 copy, 12 bytes per position, for the wide integer search and the sort payload
    const int x = x0 + tx, y = y0 + ty;
    if (x < d.W && y < d.H) {
        uint16_t *o0 = d.feat0 + ((size_t)s * d.ysz + (size_t)y * d.W + x) * 6;
#pragma unroll
        for (int k = 0; k < 6; k++) o0[k] = rec[tid][k];
    }
    const int vw = min(FT_W, d.W - x0);           // valid positions per tile row
    const int row_vec = vw * (FT_REC * 2 / 16);   // 16-byte vectors per tile row
    for (int v = tid; v < FT_H * row_vec; v += 256) {
        int r = v / row_vec, o = v % row_vec;
        if (y0 + r >= d.H) break;
        const uint4 *src = (const uint4 *)&rec[r * FT_W][0] + o;
        uint4 *dst = (uint4 *)(d.feat + ((size_t)s * d.ysz + (size_t)(y0 + r) * d.W + x0) * FT_REC) + o;
        *dst = *src;
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
    if (d.hdr[s * 4 + 3] != 0) {  // not a P picture: keep the segment populated so the device-wide sort stays aligned
        keys[(size_t)s * n + b] = (uint32_t)s << 15;
        vals[(size_t)s * n + b] = 0;
        return;
    }
    int tx = b / d.H, ty = b % d.H;
    uint16_t k = d.feat0[((size_t)s * d.ysz + (size_t)ty * d.W + tx) * 6];
    keys[(size_t)s * n + b] = ((uint32_t)s << 15) | k;  // one device-wide sort: stream id above the 15-bit sum
    vals[(size_t)s * n + b] = ((uint32_t)tx << 16) | (uint32_t)ty;
    if (k == 0) atomicOr(&d.status[s], FER_ERR_ZERO_SUM);  // the reference mis-files sum 0 (F/moestimation.cpp:153)
}

__global__ void k_sort_finish(FerDev d, const uint32_t *skeys, const uint32_t *svals)
{
    const int s = blockIdx.y;
    if (d.hdr[s * 4 + 3] != 0) return;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int n = d.W * d.H;
    skeys += (size_t)s * n;
    svals += (size_t)s * n;
    if (i < n) {
        uint32_t v = svals[i];
        int tx = v >> 16, ty = v & 0xffff;
        const uint32_t *r = (const uint32_t *)(d.feat0 + ((size_t)s * d.ysz + (size_t)ty * d.W + tx) * 6);
        uint32_t a = r[0], b = r[1], c = r[2];  // k0|k1<<16, k2|k3<<16, k4
        d.sort_pos[(size_t)s * n + i] = v;
        d.sort_k12[(size_t)s * n + i] = (a >> 16) | (b << 16);
        d.sort_k34[(size_t)s * n + i] = (b >> 16) | (c << 16);
    }
    if (i <= 16384) {  // koliko[a] = number of positions with sum < a
        int lo = 0, hi = n;
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if ((skeys[mid] & 0x7fff) < (uint32_t)i)
                lo = mid + 1;
            else
                hi = mid;
        }
        d.koliko[(size_t)s * 16385 + i] = lo;
    }
}

static int sort_end_bit(int S)
{
    int b = 15;
    while ((1 << (b - 15)) < S) b++;
    return b;
}

size_t fer_sort_tmp_bytes(int n, int S)
{
    size_t bytes = 0;
    rocprim::radix_sort_pairs((void *)nullptr, bytes, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr,
                              (uint32_t *)nullptr, (size_t)n * S, 0, sort_end_bit(S), (hipStream_t)0);
    return bytes;
}

// host side: prepare the reference structures of all P-picture streams
void fer_launch_refprep(const FerDev &d, FerSortTmp &t, const int *types, hipStream_t st)
{
    (void)types;
    dim3 gi((d.W + IT_W - 1) / IT_W, (d.H + IT_H - 1) / IT_H, d.S);
    hipLaunchKernelGGL(k_interp, gi, dim3(IT_W, IT_H), 0, st, d);
    dim3 gf((d.W + FT_W - 1) / FT_W, (d.H + FT_H - 1) / FT_H, d.S);
    hipLaunchKernelGGL(k_features, gf, dim3(256), 0, st, d);
    int n = d.W * d.H;
    hipLaunchKernelGGL(k_sort_keys, dim3((n + 255) / 256, d.S), dim3(256), 0, st, d, t.keys_in, t.vals_in);
    size_t bytes = t.tmp_bytes;
    rocprim::radix_sort_pairs(t.tmp, bytes, t.keys_in, t.keys_out, t.vals_in, t.vals_out, (size_t)n * d.S, 0,
                              sort_end_bit(d.S), st);
    int m = n > 16385 ? n : 16385;
    hipLaunchKernelGGL(k_sort_finish, dim3((m + 255) / 256, d.S), dim3(256), 0, st, d, t.keys_out, t.vals_out);
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
