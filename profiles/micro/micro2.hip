// scratch microbenchmarks (not product): cost of the building blocks of one FPS round on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../epnet_amd/csrc/dpp.h"
using namespace epnet;

typedef int veci __attribute__((ext_vector_type(32)));

#define T0 unsigned long long c0 = __builtin_amdgcn_s_memtime()
#define T1(slot) do { unsigned long long c1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[slot] = c1 - c0; } while (0)

// 1. wave-uniform indexed register read feeding a dependent add
__global__ void k_gpridx(unsigned long long *out, int iters, int seed) {
    veci t;
    for (int j = 0; j < 32; ++j) t[j] = threadIdx.x * j + seed;
    int a = seed, j = seed & 31;
    T0;
    for (int i = 0; i < iters; ++i) {
        a += t[j];
        j = __builtin_amdgcn_readfirstlane(a) & 31;
    }
    T1(0);
    if (threadIdx.x == 0) out[7] = a;
}
// 1b. indexed write + read
__global__ void k_gpridx_rw(unsigned long long *out, int iters, int seed) {
    veci t;
    for (int j = 0; j < 32; ++j) t[j] = threadIdx.x * j + seed;
    int a = seed, j = seed & 31;
    T0;
    for (int i = 0; i < iters; ++i) {
        const int v = t[j] + a;
        t[j] = v;
        a = v ^ i;
        j = (j * 5 + 1) & 31;  // scalar chain, no VALU->SALU
    }
    T1(0);
    int s = 0;
    for (int k = 0; k < 32; ++k) s += t[k];
    if (threadIdx.x == 0) out[7] = a + s;
}
// 2. dependent DPP max chain (4 steps) per iteration
__global__ void k_dpp(unsigned long long *out, int iters, int seed) {
    int v = threadIdx.x * 7 + seed;
    T0;
    for (int i = 0; i < iters; ++i) {
        v = max(v, dpp_i32<0xB1>(v));
        v = max(v, dpp_i32<0x4E>(v));
        v = max(v, dpp_i32<0x141>(v));
        v = max(v, dpp_i32<0x140>(v));
        v ^= i;
    }
    T1(0);
    if (threadIdx.x == 0) out[7] = v;
}
// 3. wave_max_all
__global__ void k_wavemax(unsigned long long *out, int iters, int seed) {
    int v = threadIdx.x * 7 + seed;
    T0;
    for (int i = 0; i < iters; ++i) v = wave_max_all(v) ^ (i + (int)threadIdx.x);
    T1(0);
    if (threadIdx.x == 0) out[7] = v;
}
// 4. VALU -> ballot -> ff1 -> readlane -> VALU round trip
__global__ void k_ballot(unsigned long long *out, int iters, int seed) {
    int v = threadIdx.x * 7 + seed;
    T0;
    for (int i = 0; i < iters; ++i) {
        const unsigned long long b = __ballot((v & 63) == (i & 63) || threadIdx.x == 63);
        const int l = (int)__builtin_ctzll(b);
        v += __builtin_amdgcn_readlane(v, l);
    }
    T1(0);
    if (threadIdx.x == 0) out[7] = v;
}
// 5. LDS write -> barrier -> LDS read, W waves
__global__ void k_barrier(unsigned long long *out, int iters, int seed) {
    __shared__ int s[2][16];
    int v = threadIdx.x + seed;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    T0;
    for (int i = 0; i < iters; ++i) {
        if (lane == (v & 63)) s[i & 1][wave] = v;
        __syncthreads();
        v = s[i & 1][lane & (nw - 1)] + i;
    }
    T1(0);
    if (threadIdx.x == 0) out[7] = v;
}
// 5b. barrier only
__global__ void k_barrier_only(unsigned long long *out, int iters, int seed) {
    int v = threadIdx.x + seed;
    T0;
    for (int i = 0; i < iters; ++i) {
        __syncthreads();
        v += i;
    }
    T1(0);
    if (threadIdx.x == 0) out[7] = v;
}
// 6. dependent LDS read chain
__global__ void k_lds(unsigned long long *out, int iters, int seed) {
    __shared__ int s[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) s[i] = (i * 17 + seed) & 1023;
    __syncthreads();
    int v = threadIdx.x;
    T0;
    for (int i = 0; i < iters; ++i) v = s[v];
    T1(0);
    if (threadIdx.x == 0) out[7] = v;
}
// 7. s_memtime back to back
__global__ void k_memtime(unsigned long long *out, int iters, int seed) {
    unsigned long long acc = 0;
    T0;
    for (int i = 0; i < iters; ++i) {
        const unsigned long long a = __builtin_amdgcn_s_memtime();
        acc += a;
    }
    T1(0);
    if (threadIdx.x == 0) out[7] = acc;
}
// 8. readlane x4 from a uniform lane + v_med3 chain (phase D tail -> phase A head)
__global__ void k_readlane4(unsigned long long *out, int iters, int seed) {
    float x = threadIdx.x * 0.5f + seed, y = x + 1.f, z = x + 2.f;
    int l = seed & 63;
    T0;
    for (int i = 0; i < iters; ++i) {
        const float cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l));
        const float cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y), l));
        const float cz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z), l));
        x = __builtin_amdgcn_fmed3f(cx, x, y) - cx;
        y = __builtin_amdgcn_fmed3f(cy, y, z) - cy;
        z = __builtin_amdgcn_fmed3f(cz, z, x) - cz;
        l = (l + 7) & 63;
    }
    T1(0);
    if (threadIdx.x == 0) out[7] = (unsigned long long)(x + y + z);
}

template <typename K>
static void run(const char *name, K k, int threads, int iters, unsigned long long *d) {
    unsigned long long h[8];
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, d, iters, 3);
        hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    }
    printf("%-28s %4d threads: %.1f cycles / iteration\n", name, threads, (double)h[0] / iters);
}

int main() {
    unsigned long long *d;
    hipMalloc(&d, 64);
    const int it = 20000;
    run("gpr-idx read + readfirstlane", k_gpridx, 64, it, d);
    run("gpr-idx read+write", k_gpridx_rw, 64, it, d);
    run("4 dependent DPP max", k_dpp, 64, it, d);
    run("wave_max_all", k_wavemax, 64, it, d);
    run("ballot+ctz+readlane", k_ballot, 64, it, d);
    for (int w = 1; w <= 16; w *= 2) run("LDS write/barrier/read", k_barrier, 64 * w, it, d);
    for (int w = 1; w <= 16; w *= 2) run("barrier only", k_barrier_only, 64 * w, it, d);
    run("dependent LDS read", k_lds, 64, it, d);
    run("s_memtime", k_memtime, 64, it, d);
    run("3 readlane + med3/sub", k_readlane4, 64, it, d);
    return 0;
}
