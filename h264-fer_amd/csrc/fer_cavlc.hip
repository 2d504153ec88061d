// fer_cavlc.hip -- macroblock-layer syntax and CAVLC residual coding (rows a13, a14 and the
// bit counting of a12 in SURVEY.md 8a): the write half of F/rbsp_encoding.cpp:175-315,
// F/residual.cpp:300-666 and the bit writer F/rbsp_IO.cpp:123-190.
//
// Entropy coding is a prefix sum: pass 1 sizes every macroblock (same code path as the writer,
// the reference does the same with residual_block_cavlc_size), a per-stream exclusive scan turns
// sizes into bit offsets behind the slice header, pass 2 writes each macroblock at its offset.
// Words shared by two macroblocks are merged with atomicOr on a pre-zeroed buffer.
#include "fer_cavlc_dev.h"
#include "fer_internal.h"

// One thread per macroblock (+ one virtual thread per stream for the slice tail).  (Round 3 tried one LANE PER SYNTAX ITEM,
// two macroblocks per wavefront with a prefix sum over the lanes: 8.1 ms per 256-stream picture against 4.1 -- the block coder's
// control flow follows the coefficient pattern, so 64 lanes diverge whether they hold 64 macroblocks or 56 blocks of two, and the
// serial macroblock header then runs on two lanes.)
template <bool WRITE>
__global__ __launch_bounds__(64) void k_cavlc(FerDev d)
{
    __shared__ int16_t stage[16 * 64];  // coefficients of the block being coded, one column per thread
    int16_t *st = stage + threadIdx.x;
    const int s = blockIdx.y;
    const int mb = blockIdx.x * blockDim.x + threadIdx.x;
    if (mb > d.nmb) return;
    const int slice_type = (int)d.hdr[s * 4 + 3];
    const int *mbt = d.mb_type + (size_t)s * d.nmb;
    uint32_t *sizes = d.mb_bits + (size_t)s * (d.nmb + 1);
    BitW w;
    bw_init<WRITE>(w, d.bits + (size_t)s * d.bits_cap_words, d.bits_cap_words, WRITE ? (size_t)sizes[mb] : 0);

    if (mb == d.nmb) {  // trailing mb_skip_run + rbsp_trailing_bits, F/rbsp_encoding.cpp:309-315
        if (slice_type == 0) {
            int run = 0;
            for (int k = d.nmb - 1; k >= 0 && mbt[k] == FER_P_SKIP; k--) run++;
            if (run > 0) bw_ue<WRITE>(w, (unsigned)run);
        }
        bw_put<WRITE>(w, 1, 1);
        if (WRITE) {
            size_t endbits = (size_t)sizes[mb] + w.bits;
            bw_flush<WRITE>(w);
            size_t nbytes = (endbits + 7) >> 3;
            d.out_bytes[s] = (uint32_t)nbytes;
            if (nbytes > d.bits_cap_words * 4) atomicOr(&d.status[s], FER_ERR_BITS_OVERFLOW);
        } else {
            sizes[mb] = w.bits;
        }
        return;
    }
    const int t = mbt[mb];
    if (slice_type == 0 && t == FER_P_SKIP) {
        if (!WRITE) sizes[mb] = 0;
        return;
    }
    const int16_t *lv = d.levels + ((size_t)s * d.nmb + mb) * FER_LEVELS;
    const int cbpL = d.cbp[((size_t)s * d.nmb + mb) * 2], cbpC = d.cbp[((size_t)s * d.nmb + mb) * 2 + 1];
    bool i16 = false, i4 = false;
    if (slice_type == 0) {
        int run = 0;
        for (int k = mb - 1; k >= 0 && mbt[k] == FER_P_SKIP; k--) run++;
        bw_ue<WRITE>(w, (unsigned)run);
        bw_ue<WRITE>(w, (unsigned)t);
        const short *mvd = d.mvd + ((size_t)s * d.nmb + mb) * 8;
        if (t == FER_P_8x8ref0) {
            for (int i = 0; i < 4; i++) bw_ue<WRITE>(w, 0);  // sub_mb_type P_L0_8x8
            for (int i = 0; i < 4; i++) {
                bw_se<WRITE>(w, mvd[i * 2]);
                bw_se<WRITE>(w, mvd[i * 2 + 1]);
            }
        } else {
            int np = t == FER_P_L0_16x16 ? 1 : 2;
            for (int i = 0; i < np; i++) {
                bw_se<WRITE>(w, mvd[i * 2]);
                bw_se<WRITE>(w, mvd[i * 2 + 1]);
            }
        }
        bw_ue<WRITE>(w, c_cbp_inter_code[(cbpC << 4) | cbpL]);
    } else {
        i4 = t == 0;
        i16 = !i4;
        bw_ue<WRITE>(w, (unsigned)t);
        if (i4) {
            const uint8_t *fl = d.i4flag + ((size_t)s * d.nmb + mb) * 16;
            for (int b = 0; b < 16; b++) {
                int f = fl[b];
                bw_put<WRITE>(w, 1, (unsigned)(f >> 3));
                if (!(f >> 3)) bw_put<WRITE>(w, 3, (unsigned)(f & 7));
            }
        }
        bw_ue<WRITE>(w, d.chroma_mode[(size_t)s * d.nmb + mb]);
        if (i4) bw_ue<WRITE>(w, c_cbp_intra_code[(cbpC << 4) | cbpL]);
    }
    if (cbpL > 0 || cbpC > 0 || i16) {
        bw_se<WRITE>(w, 0);  // mb_qp_delta
        // residual block order of F/residual.cpp:300-372
        if (i16) cavlc_block_staged<WRITE>(w, lv + FER_LV_DC16, 16, cavlc_nC(d, s, mb, true, 0, 0), st, 64);
        for (int i8 = 0; i8 < 4; i8++)
            if (cbpL & (1 << i8))
                for (int i4x = 0; i4x < 4; i4x++) {
                    int blk = i8 * 4 + i4x;
                    cavlc_block_staged<WRITE>(w, lv + blk * 16, i16 ? 15 : 16, cavlc_nC(d, s, mb, true, blk, 0), st, 64);
                }
        if (cbpC & 3)
            for (int k = 0; k < 2; k++) cavlc_block_staged<WRITE>(w, lv + FER_LV_CDC + k * 4, 4, -1, st, 64);
        if (cbpC & 2)
            for (int k = 0; k < 2; k++)
                for (int b = 0; b < 4; b++)
                    cavlc_block_staged<WRITE>(w, lv + FER_LV_CAC + (k * 4 + b) * 15, 15, cavlc_nC(d, s, mb, false, b, k), st, 64);
    }
    if (WRITE)
        bw_flush<WRITE>(w);
    else
        sizes[mb] = w.bits;
}

// per-stream exclusive scan of the nmb+1 sizes, offset by the slice header, which this kernel
// also writes (one block of 256 threads per stream: a small block finds a free slot sooner when another context's
// kernels fill the GPU)
__global__ __launch_bounds__(256) void k_bits_scan(FerDev d)
{
    __shared__ unsigned part[256];
    const int s = blockIdx.x, tid = threadIdx.x;
    uint32_t *sizes = d.mb_bits + (size_t)s * (d.nmb + 1);
    const int n = d.nmb + 1;
    const int per = (n + 255) / 256;
    int b0 = tid * per, b1 = min(b0 + per, n);
    unsigned sum = 0;
    for (int i = b0; i < b1; i++) sum += sizes[i];
    part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        unsigned v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    const unsigned hbits = d.hdr[s * 4 + 2];
    unsigned run = hbits + (tid ? part[tid - 1] : 0);
    for (int i = b0; i < b1; i++) {
        unsigned v = sizes[i];
        sizes[i] = run;
        run += v;
    }
    if (tid == 255) d.out_bytes[s] = (run + 7) >> 3;  // the picture's RBSP length (the emit pass writes the same number)
}

// The emit pass merges macroblocks into shared words with atomicOr, so the RBSP must start out zero -- but only as far as
// this picture reaches (clearing the whole capacity, 1 KB per macroblock, was 2 GB per 256-stream picture against 0.2 GB
// of bitstream).  Also places the slice header bits (<= 64) at the start of the RBSP.
__global__ __launch_bounds__(256) void k_bits_zero(FerDev d)
{
    const int s = blockIdx.y;
    uint4 *buf = (uint4 *)(d.bits + (size_t)s * d.bits_cap_words);
    const size_t cap16 = d.bits_cap_words / 4;
    const size_t n16 = min(((size_t)d.out_bytes[s] + 15) / 16 + 2, cap16);  // + slack for the flush of the last words
    const unsigned hbits = d.hdr[s * 4 + 2];
    unsigned long long h = ((unsigned long long)d.hdr[s * 4] << 32) | d.hdr[s * 4 + 1];
    h <<= (64 - hbits);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (i == 0) {
            v.x = __builtin_bswap32((uint32_t)(h >> 32));
            v.y = __builtin_bswap32((uint32_t)h);
        }
        buf[i] = v;
    }
}

void fer_launch_cavlc(const FerDev &d, hipStream_t st)
{
    dim3 g((d.nmb + 1 + 63) / 64, d.S);
    hipLaunchKernelGGL(k_cavlc<false>, g, dim3(64), 0, st, d);
    hipLaunchKernelGGL(k_bits_scan, dim3(d.S), dim3(256), 0, st, d);
    hipLaunchKernelGGL(k_bits_zero, dim3(64, d.S), dim3(256), 0, st, d);
    hipLaunchKernelGGL(k_cavlc<true>, g, dim3(64), 0, st, d);
}
