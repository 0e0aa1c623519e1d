// scratch: group kernel ceiling study (not product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#include "epnet_ops.h"

__global__ __launch_bounds__(256) void write_only(int c, int p, float *out) {
    const int bs = blockIdx.z, c0 = blockIdx.y * 16;
    const int q4 = blockIdx.x * 256 + threadIdx.x;
    if (q4 * 4 >= p) return;
    float *dst = out + ((size_t)bs * c + c0) * p + (size_t)q4 * 4;
    for (int ci = 0; ci < 16; ++ci) { *reinterpret_cast<float4 *>(dst) = make_float4(ci, q4, 1.f, 2.f); dst += p; }
}
__global__ __launch_bounds__(256) void write_only_nt(int c, int p, float *out) {
    const int bs = blockIdx.z, c0 = blockIdx.y * 16;
    const int q4 = blockIdx.x * 256 + threadIdx.x;
    if (q4 * 4 >= p) return;
    float *dst = out + ((size_t)bs * c + c0) * p + (size_t)q4 * 4;
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (int ci = 0; ci < 16; ++ci) { f4 v = {(float)ci, (float)q4, 1.f, 2.f}; __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(dst)); dst += p; }
}
__global__ void copy_f4(const float4 *in, float4 *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

int main() {
    const int B = 64, C = 96, N = 4096, M = 1024, NS = 32;
    const size_t P = (size_t)M * NS;
    float *feat, *out; int *idx;
    hipMalloc(&feat, (size_t)B * C * N * 4); hipMalloc(&out, (size_t)B * C * P * 4); hipMalloc(&idx, (size_t)B * P * 4);
    std::vector<int> h((size_t)B * P); srand(1); for (auto &v : h) v = rand() % N;
    hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(feat, 0, (size_t)B * C * N * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double out_bytes = (double)B * C * P * 4, all_bytes = out_bytes + (double)B * C * N * 4 + (double)B * P * 4;
    auto timeit = [&](const char *name, auto fn, double bytes) {
        for (int i = 0; i < 3; ++i) fn();
        hipDeviceSynchronize();
        float best = 1e9;
        for (int r = 0; r < 10; ++r) { hipEventRecord(e0, 0); fn(); hipEventRecord(e1, 0); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
        printf("%-28s %.3f ms  %.0f GB/s\n", name, best, bytes / best / 1e6);
    };
    dim3 grid((unsigned)(P / 4 / 256), C / 16, B);
    timeit("product group_points", [&] { epnet_group_points(B, C, N, M, NS, feat, idx, out, 0); }, all_bytes);
    timeit("write_only (same grid)", [&] { write_only<<<grid, 256>>>(C, (int)P, out); }, out_bytes);
    timeit("write_only nontemporal", [&] { write_only_nt<<<grid, 256>>>(C, (int)P, out); }, out_bytes);
    float4 *a = (float4 *)out; size_t n4 = (size_t)B * C * P / 8;  // copy half onto the other half
    timeit("float4 copy (r+w)", [&] { copy_f4<<<2048, 256>>>(a, a + n4, n4); }, (double)n4 * 32);
    return 0;
}
