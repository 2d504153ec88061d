// Dev microbenchmark: single-wavefront latencies on gfx950 (dependent loads, LDS, ballot chains).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

__global__ void k_chase(const unsigned *buf, int steps, long long *out, unsigned *sink)
{
    unsigned p = threadIdx.x == 0 ? 0 : 0;
    long long t0 = wall_clock64();
    for (int i = 0; i < steps; i++) p = buf[p];
    long long t1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[0] = t1 - t0;
        sink[0] = p;
    }
}
__global__ void k_chase_coh(const unsigned *buf, int steps, long long *out, unsigned *sink)
{
    unsigned p = 0;
    long long t0 = wall_clock64();
    for (int i = 0; i < steps; i++) p = __hip_atomic_load(buf + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long long t1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[0] = t1 - t0;
        sink[0] = p;
    }
}
// scattered: every lane chases its own chain (64 different lines per step)
__global__ void k_chase64(const unsigned *buf, int steps, long long *out, unsigned *sink, unsigned stride)
{
    unsigned p = threadIdx.x * stride;
    long long t0 = wall_clock64();
    for (int i = 0; i < steps; i++) p = buf[p];
    long long t1 = wall_clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = p;
}
__global__ void k_valu(int steps, long long *out, unsigned *sink)
{
    unsigned a = threadIdx.x;
    long long t0 = wall_clock64();
    for (int i = 0; i < steps; i++) a = a * 1664525u + 1013904223u;
    long long t1 = wall_clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = a;
}
__global__ void k_ballot(int steps, long long *out, unsigned *sink)
{
    unsigned a = threadIdx.x * 2654435761u;
    unsigned lo = 0, hi = 0xffffffffu;
    long long t0 = wall_clock64();
    for (int i = 0; i < steps; i++) {  // one binary-search step per iteration
        unsigned mid = lo + ((hi - lo) >> 1);
        int cnt = __popcll(__ballot(a <= mid));
        if (cnt >= 32) hi = mid; else lo = mid + 1;
        if (lo >= hi) { lo = 0; hi = 0xffffffffu; a = a * 1664525u + 1013904223u; }
    }
    long long t1 = wall_clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = lo + a;
}
__global__ void k_lds(int steps, long long *out, unsigned *sink)
{
    __shared__ unsigned l[256];
    l[threadIdx.x] = (threadIdx.x * 7 + 1) & 63;
    __syncthreads();
    unsigned p = threadIdx.x;
    long long t0 = wall_clock64();
    for (int i = 0; i < steps; i++) p = l[p];
    long long t1 = wall_clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = p;
}
__global__ void k_barrier(int steps, long long *out, unsigned *sink)
{
    __shared__ unsigned l[256];
    unsigned p = threadIdx.x;
    long long t0 = wall_clock64();
    for (int i = 0; i < steps; i++) {
        l[(p + i) & 63] = p;
        __syncthreads();
        p = l[(p + 1) & 63];
        __syncthreads();
    }
    long long t1 = wall_clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = p;
}

int main()
{
    const size_t n_big = (size_t)1 << 30;  // 4 GB of uint32
    const size_t n_small = 1 << 14;        // 64 KB
    unsigned *d_big, *d_small, *sink;
    long long *out;
    hipMalloc(&d_big, n_big * 4);
    hipMalloc(&d_small, n_small * 4);
    hipMalloc(&sink, 4096);
    hipMalloc(&out, 64);
    // random cyclic permutation with large strides: element i -> (i * A + B) mod n  (n power of two, A odd => bijection... use LCG cycle)
    {
        std::vector<unsigned> h(n_small);
        std::vector<unsigned> perm(n_small);
        std::iota(perm.begin(), perm.end(), 0u);
        std::mt19937 rng(1);
        std::shuffle(perm.begin() + 1, perm.end(), rng);
        for (size_t i = 0; i < n_small; i++) h[perm[i]] = perm[(i + 1) % n_small];
        hipMemcpy(d_small, h.data(), n_small * 4, hipMemcpyHostToDevice);
    }
    {
        // sparse chain over the 4 GB buffer: 1M nodes at random 4 KB-separated places
        const size_t nodes = 1 << 20;
        std::vector<unsigned> pos(nodes);
        std::mt19937 rng(2);
        for (size_t i = 0; i < nodes; i++) pos[i] = (unsigned)(i * (n_big / nodes));
        std::shuffle(pos.begin() + 1, pos.end(), rng);
        std::vector<unsigned> h(n_big / 1024);
        hipMemset(d_big, 0, n_big * 4);
        for (size_t i = 0; i < nodes; i++) {
            unsigned v = pos[(i + 1) % nodes];
            hipMemcpy(d_big + pos[i], &v, 4, hipMemcpyHostToDevice);
            if (i > 200000) break;  // 200k nodes are plenty
        }
    }
    long long h;
    auto rep = [&](const char *name, int steps) {
        hipDeviceSynchronize();
        hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
        printf("%-28s %8.1f ns/step\n", name, (double)h * 10.0 / steps);
    };
    for (int r = 0; r < 2; r++) {
        k_chase<<<1, 64>>>(d_small, 20000, out, sink); rep("global dependent (64KB set)", 20000);
        k_chase<<<1, 64>>>(d_big, 100000, out, sink); rep("global dependent (4GB set)", 100000);
        k_chase_coh<<<1, 64>>>(d_small, 20000, out, sink); rep("agent-scope load (64KB set)", 20000);
        k_chase_coh<<<1, 64>>>(d_big, 100000, out, sink); rep("agent-scope load (4GB set)", 100000);
        k_valu<<<1, 64>>>(100000, out, sink); rep("VALU dependent mad", 100000);
        k_ballot<<<1, 64>>>(100000, out, sink); rep("ballot binary-search step", 100000);
        k_lds<<<1, 64>>>(100000, out, sink); rep("LDS dependent read", 100000);
        k_barrier<<<1, 64>>>(20000, out, sink); rep("LDS write+bar+read+bar", 20000);
    }
    return 0;
}
