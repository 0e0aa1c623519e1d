// scratch microbenchmarks (not product): clock probe + FPS timing through the C ABI
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "epnet_ops.h"

__global__ void clock_probe(unsigned long long *out, int spin) {
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float a = threadIdx.x;
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = (unsigned long long)a; }
}

// dependent-chain latency and independent-issue rate of v_add_f32 for ONE wave
__global__ void valu_probe(unsigned long long *out, int iters) {
    float a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3, e = a + 4, f = a + 5, g = a + 6, h = a + 7;
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) a = a + 1.0f;
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 2; ++k) { a += 1.f; b += 1.f; c += 1.f; d += 1.f; e += 1.f; f += 1.f; g += 1.f; h += 1.f; }
    }
    unsigned long long c2 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = c2 - c1; out[2] = (unsigned long long)(a+b+c+d+e+f+g+h); }
}

int main(int argc, char **argv) {
    unsigned long long *d, h[4];
    hipMalloc(&d, 64);
    for (int rep = 0; rep < 3; ++rep) {
        clock_probe<<<1, 64>>>(d, 2000000);
        hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
        printf("clock probe: %llu shader cycles in %llu x10ns -> %.1f MHz\n", h[0], h[1], (double)h[0] / h[1] * 100.0);
    }
    for (int waves = 1; waves <= 16; waves *= 2) {
        valu_probe<<<1, 64 * waves>>>(d, 10000);
        hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
        printf("valu probe %2d waves: dependent v_add %.2f cyc/instr, 8-way independent %.2f cyc/instr\n", waves, h[0] / 160000.0, h[1] / 160000.0);
    }
    // FPS timing
    struct Cfg { int b, n, m; } cfgs[] = {{1,16384,4096},{1,4096,1024},{1,1024,256},{1,256,64},{64,512,128}};
    for (auto c : cfgs) {
        std::vector<float> xyz((size_t)c.b * c.n * 3);
        srand(1); for (auto &v : xyz) v = (float)rand() / RAND_MAX * 40.f;
        float *dx, *dt; int *di;
        hipMalloc(&dx, xyz.size() * 4); hipMalloc(&dt, (size_t)c.b * c.n * 4); hipMalloc(&di, (size_t)c.b * c.m * 4);
        hipMemcpy(dx, xyz.data(), xyz.size() * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e9;
        for (int r = 0; r < 5; ++r) {
            hipEventRecord(e0, 0);
            int rc = epnet_furthest_point_sampling(c.b, c.n, c.m, dx, nullptr, di, 0);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            if (rc) printf("rc=%d\n", rc);
        }
        printf("fps b=%d n=%d m=%d: %.3f ms, %.3f us/iter\n", c.b, c.n, c.m, best, best * 1e3 / (c.m - 1));
        hipFree(dx); hipFree(dt); hipFree(di);
    }
    return 0;
}
