// experiment (not product): what slows the LDS row gather down beside the level-1 FPS?  A "squatter" kernel holds an
// FPS-like share of every CU (8 waves, ~200 VGPRs, 21 KB LDS) and either sleeps, spins on the VALU, or hammers LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "epnet_ops.h"

template <int MODE, int NV>
__global__ __launch_bounds__(512) void squatter(unsigned long long ticks, float *sink) {
    __shared__ float lds[5200];
    float r[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) r[i] = threadIdx.x * 0.5f + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int it = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
        if (MODE == 0) {
            __builtin_amdgcn_s_sleep(64);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < NV; ++i) r[i] = r[i] * 1.0001f + 0.5f;
        } else {
            lds[(threadIdx.x * 17 + it) % 5200] = r[it % NV];
            __syncthreads();
            r[0] += lds[(threadIdx.x * 31 + it) % 5200];
        }
        ++it;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += r[i];
    if (s == 12345.678f) sink[0] = s + lds[0];
}

int main() {
    const int b = 256, c = 96, n = 4096, m = 1024, ns = 16;
    const size_t p = (size_t)m * ns;
    float *feat, *out, *sink; int *idx;
    hipMalloc(&feat, (size_t)b * c * n * 4); hipMalloc(&out, (size_t)b * c * p * 4); hipMalloc(&idx, (size_t)b * p * 4); hipMalloc(&sink, 64);
    std::vector<int> h((size_t)b * p);
    srand(1); for (auto &v : h) v = rand() % n;
    hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(feat, 0, (size_t)b * c * n * 4);
    hipStream_t sa, sb; hipStreamCreate(&sa); hipStreamCreate(&sb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto gather_ms = [&]() {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, sb);
            epnet_group_points(b, c, n, m, ns, feat, idx, out, sb);
            hipEventRecord(e1, sb); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        return best;
    };
    printf("gather alone: %.3f ms\n", gather_ms());
    const unsigned long long ticks = 100000000ull / 1000 * 20;  // 100 MHz realtime clock: 20 ms
#define RUN(MODE, NV, BLOCKS, label) do { \
        hipLaunchKernelGGL((squatter<MODE, NV>), dim3(BLOCKS), dim3(512), 0, sa, ticks, sink); \
        hipDeviceSynchronize == nullptr; \
        float g = gather_ms(); hipStreamSynchronize(sa); \
        printf("gather beside %-44s (%d blocks): %.3f ms\n", label, BLOCKS, g); } while (0)
    RUN(0, 180, 256, "sleeping squatter, ~190 VGPRs");
    RUN(0, 16, 256, "sleeping squatter, few VGPRs");
    RUN(1, 180, 256, "VALU-spinning squatter, ~190 VGPRs");
    RUN(1, 16, 256, "VALU-spinning squatter, few VGPRs");
    RUN(2, 16, 256, "LDS + barrier squatter, few VGPRs");
    RUN(0, 180, 128, "sleeping squatter, ~190 VGPRs");
    return 0;
}
