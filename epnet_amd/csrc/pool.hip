// pool.hip -- the neighbourhood max-pool of a set-abstraction level for gfx950.
//
// The reference pools the shared MLP's output (B, C, npoint, nsample) over the nsample axis with the stock
// F.max_pool2d(kernel_size=[1, nsample]) (pointnet2_lib/pointnet2/pointnet2_modules.py:61-68). That tensor is the
// largest one an SA level touches after the grouped tensor itself (level 1: 2 x 64 x 4096 x 32 floats = 67 MB for two
// scenes), and the stock kernel walks it with one thread per OUTPUT element, i.e. with a stride of nsample floats between
// neighbouring lanes: 0.24 ms for those 67 MB (0.3 TB/s). Here a row of nsample contiguous floats is read by nsample/4
// neighbouring lanes with 16-byte loads (a wave covers 1 KB of contiguous input per load instruction), reduced inside the
// lane group with DPP-free shuffles, and the row's maximum and its position are written by the group's first lane.
// Pure HBM-bound byte work: algorithmic bytes = rows * (nsample * 4 + 4 [+ 4 for the arg-max]).
//
// Values: the maximum under `>`; ties keep the lowest position (F.max_pool2d's scan order with strict `>`); a NaN in the
// row makes the output NaN (as the stock op). The backward routes each output gradient to the recorded position and
// writes zeros elsewhere, which is what the stock backward does with its recorded indices.
#include "common.h"

namespace epnet {

constexpr int kPoolThreads = 256;
constexpr int kPoolUnroll = 4;  // 16-byte loads in flight per thread

__device__ __forceinline__ bool pool_better(float v, int i, float best, int bi) {
    // NaN outranks everything; otherwise larger value, then lower position
    const bool vn = v != v, bn = best != best;
    if (vn != bn) return vn;
    if (vn) return i < bi;
    return v > best || (v == best && i < bi);
}

// NS = nsample (power of two, 4..256): NS/4 lanes per row
template <int NS>
__global__ __launch_bounds__(kPoolThreads) void pool_max_vec_kernel(long long rows, const float *__restrict__ x,
                                                                    float *__restrict__ out, int *__restrict__ arg) {
    constexpr int G = NS / 4;                       // lanes per row (1..64)
    constexpr int kRowsPerPass = kPoolThreads / G;  // rows per block per load instruction
    const int sub = threadIdx.x % G;
    const long long row0 = (long long)blockIdx.x * (kRowsPerPass * kPoolUnroll) + threadIdx.x / G;
    float4 v[kPoolUnroll];
#pragma unroll
    for (int u = 0; u < kPoolUnroll; ++u) {
        const long long r = row0 + (long long)u * kRowsPerPass;
        v[u] = r < rows ? *reinterpret_cast<const float4 *>(x + r * NS + sub * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < kPoolUnroll; ++u) {
        const long long r = row0 + (long long)u * kRowsPerPass;
        float best = v[u].x;
        int bi = sub * 4;
        if (pool_better(v[u].y, sub * 4 + 1, best, bi)) { best = v[u].y; bi = sub * 4 + 1; }
        if (pool_better(v[u].z, sub * 4 + 2, best, bi)) { best = v[u].z; bi = sub * 4 + 2; }
        if (pool_better(v[u].w, sub * 4 + 3, best, bi)) { best = v[u].w; bi = sub * 4 + 3; }
#pragma unroll
        for (int off = G / 2; off >= 1; off >>= 1) {  // groups are aligned runs of G lanes inside a wave
            const float ov = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (pool_better(ov, oi, best, bi)) { best = ov; bi = oi; }
        }
        if (sub == 0 && r < rows) {
            out[r] = best;
            if (arg) arg[r] = bi;
        }
    }
}

// any nsample: one wave per row
__global__ __launch_bounds__(kPoolThreads) void pool_max_row_kernel(long long rows, int ns, const float *__restrict__ x,
                                                                    float *__restrict__ out, int *__restrict__ arg) {
    const int lane = lane_id();
    const long long r = (long long)blockIdx.x * (kPoolThreads / 64) + (threadIdx.x >> 6);
    if (r >= rows) return;
    float best = 0.f;
    int bi = 0x7fffffff;  // "nothing yet": loses against any real element of equal value
    bool have = false;
    for (int i = lane; i < ns; i += 64) {
        const float v = x[r * ns + i];
        if (!have || pool_better(v, i, best, bi)) { best = v; bi = i; have = true; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        const bool oh = __shfl_xor((int)have, off, 64) != 0;
        if (oh && (!have || pool_better(ov, oi, best, bi))) { best = ov; bi = oi; have = true; }
    }
    if (lane == 0) {
        out[r] = best;
        if (arg) arg[r] = bi;
    }
}

// grad_x (rows, ns) = grad_out[row] at arg[row], zero elsewhere: every element written once, 16 bytes per thread
__global__ __launch_bounds__(kPoolThreads) void pool_max_grad_vec_kernel(long long rows, int ns, const float *__restrict__ grad_out,
                                                                         const int *__restrict__ arg, float *__restrict__ grad_x) {
    const long long quads = rows * (ns / 4);
    for (long long q = (long long)blockIdx.x * kPoolThreads + threadIdx.x; q < quads; q += (long long)gridDim.x * kPoolThreads) {
        const long long r = q / (ns / 4);
        const int i0 = (int)(q - r * (ns / 4)) * 4;
        const int a = arg[r] - i0;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a >= 0 && a < 4) {
            const float g = grad_out[r];
            o.x = a == 0 ? g : 0.f;
            o.y = a == 1 ? g : 0.f;
            o.z = a == 2 ? g : 0.f;
            o.w = a == 3 ? g : 0.f;
        }
        store_stream(grad_x + q * 4, o.x, o.y, o.z, o.w);
    }
}

__global__ __launch_bounds__(kPoolThreads) void pool_max_grad_kernel(long long rows, int ns, const float *__restrict__ grad_out,
                                                                     const int *__restrict__ arg, float *__restrict__ grad_x) {
    const long long total = rows * ns;
    for (long long e = (long long)blockIdx.x * kPoolThreads + threadIdx.x; e < total; e += (long long)gridDim.x * kPoolThreads) {
        const long long r = e / ns;
        grad_x[e] = (int)(e - r * ns) == arg[r] ? grad_out[r] : 0.f;
    }
}

}  // namespace epnet

using namespace epnet;

extern "C" int epnet_pool_max(long long rows, int nsample, const float *x, float *out, int *arg, epnet_stream_t stream) {
    EPNET_REQUIRE(rows >= 0 && nsample >= 1);
    if (rows == 0) return EPNET_OK;
    EPNET_REQUIRE(x && out);
    hipStream_t s = (hipStream_t)stream;
    const bool vec = (nsample & (nsample - 1)) == 0 && nsample >= 4 && nsample <= 256 && ((uintptr_t)x & 15) == 0;
    if (vec) {
        const int g = nsample / 4;
        const long long per_block = (long long)(kPoolThreads / g) * kPoolUnroll;
        const long long blocks = div_up64(rows, per_block);
        if (blocks > 0x7fffffffll) return EPNET_ELIMIT;
#define EPNET_POOL(NS_) \
    hipLaunchKernelGGL((pool_max_vec_kernel<NS_>), dim3((unsigned)blocks), dim3(kPoolThreads), 0, s, rows, x, out, arg)
        switch (nsample) {
            case 4: EPNET_POOL(4); break;
            case 8: EPNET_POOL(8); break;
            case 16: EPNET_POOL(16); break;
            case 32: EPNET_POOL(32); break;
            case 64: EPNET_POOL(64); break;
            case 128: EPNET_POOL(128); break;
            default: EPNET_POOL(256); break;
        }
#undef EPNET_POOL
    } else {
        const long long blocks = div_up64(rows, kPoolThreads / 64);
        if (blocks > 0x7fffffffll) return EPNET_ELIMIT;
        hipLaunchKernelGGL(pool_max_row_kernel, dim3((unsigned)blocks), dim3(kPoolThreads), 0, s, rows, nsample, x, out, arg);
    }
    return check_launch("pool_max");
}

extern "C" int epnet_pool_max_grad(long long rows, int nsample, const float *grad_out, const int *arg, float *grad_x,
                                   epnet_stream_t stream) {
    EPNET_REQUIRE(rows >= 0 && nsample >= 1);
    if (rows == 0) return EPNET_OK;
    EPNET_REQUIRE(grad_out && arg && grad_x);
    hipStream_t s = (hipStream_t)stream;
    const long long total = rows * nsample;
    if (nsample % 4 == 0 && ((uintptr_t)grad_x & 15) == 0) {
        const long long blocks = div_up64(total / 4, kPoolThreads);
        hipLaunchKernelGGL(pool_max_grad_vec_kernel, dim3((unsigned)(blocks > 1048576 ? 1048576 : blocks)), dim3(kPoolThreads), 0, s,
                           rows, nsample, grad_out, arg, grad_x);
    } else {
        const long long blocks = div_up64(total, kPoolThreads);
        hipLaunchKernelGGL(pool_max_grad_kernel, dim3((unsigned)(blocks > 1048576 ? 1048576 : blocks)), dim3(kPoolThreads), 0, s, rows,
                           nsample, grad_out, arg, grad_x);
    }
    return check_launch("pool_max_grad");
}
