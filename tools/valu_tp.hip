// Dev microbenchmark: issue throughput of the integer VALU instructions the motion kernels are made of, on gfx950.
// Every CU runs `waves` wavefronts per SIMD, each executing a long stream of independent instructions of one kind
// (8 accumulator chains); reports shader cycles (s_memtime) per wave-instruction per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_tp.hip -o tools/valu_tp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define REP 64
#define ITERS 1024

#define KERNEL(NAME, BODY)                                                                    \
    __global__ __launch_bounds__(256) void NAME(unsigned *sink, long long *cyc)               \
    {                                                                                         \
        unsigned a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 ^ 13, a6 = a0 + 17, \
                 a7 = a0 * 19;                                                                \
        unsigned b = blockIdx.x * 2654435761u + threadIdx.x, c = b ^ 0x5bd1e995u;             \
        unsigned long long msk = __ballot(b & 1);                                             \
        long long t0 = __builtin_amdgcn_s_memtime();                                          \
        for (int it = 0; it < ITERS; it++) {                                                  \
            _Pragma("unroll") for (int r = 0; r < REP / 8; r++) { BODY }                      \
        }                                                                                     \
        __builtin_amdgcn_s_waitcnt(0);                                                        \
        long long t1 = __builtin_amdgcn_s_memtime();                                          \
        sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;  \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;      \
    }

#define X8(OP) OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)

#define OP_SAD16(a) asm volatile("v_sad_u16 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));
#define OP_SAD8(a) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));
#define OP_ADD(a) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a) : "v"(b));
#define OP_PERM(a) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define OP_PKSUB(a) asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(a) : "v"(b));
#define OP_PKMAX(a) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a) : "v"(b));
#define OP_MULLO(a) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b));
#define OP_MUL24(a) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(b));
#define OP_MAD24(a) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));
#define OP_CNDMASK(a) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "s"(msk));
#define OP_SUB(a) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a) : "v"(b));
#define OP_SHL(a) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a));
#define OP_OR(a) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a) : "v"(b));
#define OP_MAX(a) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a) : "v"(b));
#define OP_MOV(a) asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "v"(b));
#define OP_ADD3(a) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define OP_ANDOR(a) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define OP_CMPONLY(a) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");
#define OP_CMPS(a) { unsigned long long s_; asm volatile("v_cmp_lt_u32 %0, %1, %2" : "=s"(s_) : "v"(a), "v"(b)); asm volatile("" : : "s"(s_)); }
#define OP_DOT4(a) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));
#define OP_PKADD(a) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a) : "v"(b));
#define OP_ADDU16(a) asm volatile("v_add_u16 %0, %0, %1" : "+v"(a) : "v"(b));
#define OP_SAD32(a) asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));
#define OP_SUBREV(a) asm volatile("v_subrev_u32 %0, %0, %1" : "+v"(a) : "v"(b));
#define OP_ADDSDWA(a) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(a) : "v"(b));
#define OP_LDSR(a) { unsigned t_; asm volatile("ds_read_b32 %0, %1" : "=v"(t_) : "v"((threadIdx.x & 255) * 4)); asm volatile("s_waitcnt lgkmcnt(4)"); a += t_; }
#define OP_CMP(a) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a) : "v"(b), "v"(c) : "vcc");
#define OP_DPP(a) asm volatile("v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a));
#define OP_DPPROW(a) asm volatile("v_add_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(a));
#define OP_ALIGN(a) asm volatile("v_alignbyte_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define OP_BFE(a) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(a));
#define OP_AND(a) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(b));
#define OP_MIN3(a) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define OP_LSHLADD(a) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a) : "v"(b));
#define OP_MBCNT(a) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a) : "v"(b));
#define OP_READLANE(a) { unsigned s_; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s_) : "v"(a)); asm volatile("v_add_u32 %0, %1, %0" : "+v"(a) : "s"(s_)); }
#define OP_QSAD(a) { unsigned long long q_; asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %1" : "=v"(q_) : "v"((unsigned long long)a << 32 | b), "v"(c)); a += (unsigned)q_; }

KERNEL(k_sad16, X8(OP_SAD16))
KERNEL(k_sad8, X8(OP_SAD8))
KERNEL(k_add, X8(OP_ADD))
KERNEL(k_perm, X8(OP_PERM))
KERNEL(k_pksub, X8(OP_PKSUB))
KERNEL(k_pkmax, X8(OP_PKMAX))
KERNEL(k_mullo, X8(OP_MULLO))
KERNEL(k_mul24, X8(OP_MUL24))
KERNEL(k_mad24, X8(OP_MAD24))
KERNEL(k_cndmask, X8(OP_CNDMASK))
KERNEL(k_sub, X8(OP_SUB))
KERNEL(k_shl, X8(OP_SHL))
KERNEL(k_or, X8(OP_OR))
KERNEL(k_max, X8(OP_MAX))
KERNEL(k_add3, X8(OP_ADD3))
KERNEL(k_andor, X8(OP_ANDOR))
KERNEL(k_cmponly, X8(OP_CMPONLY))
KERNEL(k_cmps, X8(OP_CMPS))
KERNEL(k_dot4, X8(OP_DOT4))
KERNEL(k_pkadd, X8(OP_PKADD))
KERNEL(k_addu16, X8(OP_ADDU16))
KERNEL(k_sad32, X8(OP_SAD32))
KERNEL(k_addsdwa, X8(OP_ADDSDWA))
KERNEL(k_cmp_cnd, X8(OP_CMP))
KERNEL(k_dpp_quad, X8(OP_DPP))
KERNEL(k_dpp_row, X8(OP_DPPROW))
KERNEL(k_alignbyte, X8(OP_ALIGN))
KERNEL(k_bfe, X8(OP_BFE))
KERNEL(k_and, X8(OP_AND))
KERNEL(k_min3, X8(OP_MIN3))
KERNEL(k_lshladd, X8(OP_LSHLADD))
KERNEL(k_mbcnt, X8(OP_MBCNT))
KERNEL(k_readlane_add, X8(OP_READLANE))

typedef void (*kern_t)(unsigned *, long long *);

int main()
{
    unsigned *sink;
    long long *cyc;
    const int maxblocks = 256 * 8;
    hipMalloc(&sink, (size_t)maxblocks * 256 * 4);
    hipMalloc(&cyc, (size_t)maxblocks * 4 * 8);
    struct {
        const char *name;
        kern_t k;
        int per;  // instructions per OP
    } ks[] = {{"v_sad_u16", k_sad16, 1}, {"v_sad_u8", k_sad8, 1}, {"v_add_u32", k_add, 1}, {"v_perm_b32", k_perm, 1},
              {"v_pk_sub_u16", k_pksub, 1}, {"v_pk_max_i16", k_pkmax, 1}, {"v_mul_lo_u32", k_mullo, 1},
              {"v_mul_u32_u24", k_mul24, 1}, {"v_mad_u32_u24", k_mad24, 1}, {"v_cndmask(sgpr)", k_cndmask, 1}, {"v_sub_u32", k_sub, 1}, {"v_lshlrev_b32", k_shl, 1}, {"v_or_b32", k_or, 1},
              {"v_max_u32", k_max, 1}, {"v_add3_u32", k_add3, 1}, {"v_and_or_b32", k_andor, 1}, {"v_cmp->vcc", k_cmponly, 1},
              {"v_cmp->sgpr", k_cmps, 1}, {"v_dot4_u32_u8", k_dot4, 1}, {"v_pk_add_u16", k_pkadd, 1}, {"v_add_u16", k_addu16, 1},
              {"v_sad_u32", k_sad32, 1}, {"v_add_u32 sdwa", k_addsdwa, 1},
              {"v_cmp+v_cndmask", k_cmp_cnd, 2}, {"v_add dpp quad", k_dpp_quad, 1}, {"v_add dpp row_mirror", k_dpp_row, 1},
              {"v_alignbyte", k_alignbyte, 1}, {"v_bfe_u32", k_bfe, 1}, {"v_and_b32", k_and, 1}, {"v_min3_u32", k_min3, 1},
              {"v_lshl_add_u32", k_lshladd, 1}, {"v_mbcnt_lo", k_mbcnt, 1}, {"v_readlane+v_add(s)", k_readlane_add, 2}};
    printf("%-24s", "instruction");
    for (int w : {1, 2, 4, 8}) printf("  %dw/SIMD cyc/inst", w);
    printf("\n");
    for (auto &e : ks) {
        printf("%-24s", e.name);
        for (int w : {1, 2, 4, 8}) {
            // 256 threads = 4 wavefronts = one per SIMD; w workgroups per CU
            int blocks = 256 * w;
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, sink, cyc);
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, sink, cyc);
            hipDeviceSynchronize();
            std::vector<long long> h(blocks * 4);
            hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            double med = (double)h[h.size() / 2];
            // each wave issued ITERS * REP ops; a SIMD hosted w waves for about `med` cycles
            double per_inst = med / ((double)ITERS * REP * e.per * w);
            printf("  %17.2f", per_inst);
        }
        printf("\n");
    }
    return 0;
}
