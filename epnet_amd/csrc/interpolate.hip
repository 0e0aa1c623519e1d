// interpolate.hip -- three_nn, three_interpolate and its gradient for gfx950.
//
// Replaces three_nn_kernel_fast, three_interpolate_kernel_fast and three_interpolate_grad_kernel_fast
// (pointnet2_lib/pointnet2/src/interpolate_gpu.cu:9-52, 77-97, 120-142).
//
// three_nn: the reference gives each unknown point to one thread that scans all m known points
// (n/64 waves in flight -- one per CU at n = 16384). Here kSplit = 8 lanes share one unknown,
// each scans every 8th known point out of an LDS tile and keeps its own 3 best; the 8 partial
// lists are merged with xor-shuffles. The reference's strict '<' insertion in index order
// (:37-48) yields exactly the 3 smallest pairs under the lexicographic order (d, k) -- earlier
// index wins a distance tie -- so merging partial lists under that same order is bit-identical,
// whatever the split. The reference's double accumulators initialised to 1e40 (:30) behave like
// float +inf (a float d is never >= 1e40 unless it is inf/nan, and (float)1e40 == inf).
//
// three_interpolate: out = w0*p[i0] + w1*p[i1] + w2*p[i2], left to right, no contraction (:96).
// A thread owns 4 consecutive unknowns, keeps their 12 indices and weights in registers and
// loops over a chunk of channels (the reference re-reads idx and weight for every channel).
#include <math.h>

#include "common.h"

namespace epnet {

constexpr int kNnThreads = 256;
constexpr int kNnSplit = 8;                       // lanes per unknown
constexpr int kNnPerBlock = kNnThreads / kNnSplit;  // 32 unknowns per workgroup
constexpr int kNnTile = 4096;                     // known points per LDS tile (48 KiB)

struct Top3 {
    float d0, d1, d2;
    int i0, i1, i2;
};

__device__ __forceinline__ bool pair_less(float da, int ia, float db, int ib) {
    return da < db || (da == db && ia < ib);
}

// insert (d,k) into a list sorted by (d,k); used for merging (k may be smaller than entries)
__device__ __forceinline__ void top3_insert(Top3 &t, float d, int k) {
    if (pair_less(d, k, t.d0, t.i0)) {
        t.d2 = t.d1; t.i2 = t.i1;
        t.d1 = t.d0; t.i1 = t.i0;
        t.d0 = d; t.i0 = k;
    } else if (pair_less(d, k, t.d1, t.i1)) {
        t.d2 = t.d1; t.i2 = t.i1;
        t.d1 = d; t.i1 = k;
    } else if (pair_less(d, k, t.d2, t.i2)) {
        t.d2 = d; t.i2 = k;
    }
}

__global__ __launch_bounds__(kNnThreads) void three_nn_kernel(int n, int m, const float *__restrict__ unknown,
                                                              const float *__restrict__ known,
                                                              float *__restrict__ dist2, int *__restrict__ idx) {
    __shared__ float tile[kNnTile * 3];
    const int bs = blockIdx.y;
    unknown += (size_t)bs * n * 3;
    known += (size_t)bs * m * 3;
    dist2 += (size_t)bs * n * 3;
    idx += (size_t)bs * n * 3;

    const int part = threadIdx.x & (kNnSplit - 1);
    const int pt = blockIdx.x * kNnPerBlock + (threadIdx.x / kNnSplit);
    const bool ok = pt < n;
    const float ux = ok ? unknown[pt * 3 + 0] : 0.f;
    const float uy = ok ? unknown[pt * 3 + 1] : 0.f;
    const float uz = ok ? unknown[pt * 3 + 2] : 0.f;

    const float inf = __builtin_huge_valf();
    const int big = 0x7fffffff;
    Top3 t = {inf, inf, inf, big, big, big};

    for (int t0 = 0; t0 < m; t0 += kNnTile) {
        const int tn = min(kNnTile, m - t0);
        __syncthreads();
        for (int e = threadIdx.x; e < tn * 3; e += kNnThreads) tile[e] = known[(size_t)t0 * 3 + e];
        __syncthreads();
        for (int k = part; k < tn; k += kNnSplit) {
            const float x = tile[k * 3 + 0], y = tile[k * 3 + 1], z = tile[k * 3 + 2];
            const float dx = ux - x, dy = uy - y, dz = uz - z;
            const float d = dx * dx + dy * dy + dz * dz;
            // ascending k within a lane: strict '<' on d alone is the (d,k) order here
            if (d < t.d0) {
                t.d2 = t.d1; t.i2 = t.i1;
                t.d1 = t.d0; t.i1 = t.i0;
                t.d0 = d; t.i0 = t0 + k;
            } else if (d < t.d1) {
                t.d2 = t.d1; t.i2 = t.i1;
                t.d1 = d; t.i1 = t0 + k;
            } else if (d < t.d2) {
                t.d2 = d; t.i2 = t0 + k;
            }
        }
    }

    // merge the kNnSplit partial lists (lanes part = 0..7 are adjacent)
#pragma unroll
    for (int off = 1; off < kNnSplit; off <<= 1) {
        const float od0 = __shfl_xor(t.d0, off, 64), od1 = __shfl_xor(t.d1, off, 64), od2 = __shfl_xor(t.d2, off, 64);
        const int oi0 = __shfl_xor(t.i0, off, 64), oi1 = __shfl_xor(t.i1, off, 64), oi2 = __shfl_xor(t.i2, off, 64);
        top3_insert(t, od0, oi0);
        top3_insert(t, od1, oi1);
        top3_insert(t, od2, oi2);
    }

    if (ok && part == 0) {
        dist2[pt * 3 + 0] = t.d0;
        dist2[pt * 3 + 1] = t.d1;
        dist2[pt * 3 + 2] = t.d2;
        // slots never filled keep the reference's initial index 0 (:31)
        idx[pt * 3 + 0] = t.i0 == big ? 0 : t.i0;
        idx[pt * 3 + 1] = t.i1 == big ? 0 : t.i1;
        idx[pt * 3 + 2] = t.i2 == big ? 0 : t.i2;
    }
}

constexpr int kTiThreads = 256;
constexpr int kTiChan = 16;

__global__ __launch_bounds__(kTiThreads) void three_interpolate_kernel(int c, int m, int n,
                                                                       const float *__restrict__ points,
                                                                       const int *__restrict__ idx,
                                                                       const float *__restrict__ weight,
                                                                       float *__restrict__ out) {
    const int bs = blockIdx.z;
    const int c0 = blockIdx.y * kTiChan;
    const int i0 = (blockIdx.x * kTiThreads + threadIdx.x) * 4;
    if (i0 >= n) return;
    const int cnt = min(4, n - i0);
    int ix[4][3];
    float w[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const bool ok = u < cnt;
            ix[u][j] = ok ? idx[((size_t)bs * n + i0 + u) * 3 + j] : 0;
            w[u][j] = ok ? weight[((size_t)bs * n + i0 + u) * 3 + j] : 0.f;
        }
    const int cend = min(c, c0 + kTiChan);
    const bool vec = (cnt == 4) && (n % 4 == 0) && ((uintptr_t)out % 16 == 0);
    for (int ci = c0; ci < cend; ++ci) {
        const float *src = points + ((size_t)bs * c + ci) * m;
        float *dst = out + ((size_t)bs * c + ci) * n + i0;
        float r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = w[u][0] * src[ix[u][0]] + w[u][1] * src[ix[u][1]] + w[u][2] * src[ix[u][2]];
        if (vec) {
            *reinterpret_cast<float4 *>(dst) = make_float4(r[0], r[1], r[2], r[3]);
        } else {
            for (int u = 0; u < cnt; ++u) dst[u] = r[u];
        }
    }
}

// gradient: rows of grad_points (m floats each) accumulated in LDS, see group.hip
__global__ __launch_bounds__(kTiThreads) void three_interpolate_grad_lds_kernel(int c, int n, int m, int rows,
                                                                                const float *__restrict__ grad_out,
                                                                                const int *__restrict__ idx,
                                                                                const float *__restrict__ weight,
                                                                                float *__restrict__ grad_points) {
    extern __shared__ float acc[];
    const int bs = blockIdx.y;
    const int c0 = blockIdx.x * rows;
    const int nr = min(rows, c - c0);
    for (int e = threadIdx.x; e < nr * m; e += kTiThreads) acc[e] = 0.f;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += kTiThreads) {
        const int *ix = idx + ((size_t)bs * n + i) * 3;
        const float *w = weight + ((size_t)bs * n + i) * 3;
        const int i0 = ix[0], i1 = ix[1], i2 = ix[2];
        const float w0 = w[0], w1 = w[1], w2 = w[2];
        for (int r = 0; r < nr; ++r) {
            const float g = grad_out[((size_t)bs * c + c0 + r) * n + i];
            atomicAdd(&acc[r * m + i0], g * w0);
            atomicAdd(&acc[r * m + i1], g * w1);
            atomicAdd(&acc[r * m + i2], g * w2);
        }
    }
    __syncthreads();
    float *gp = grad_points + ((size_t)bs * c + c0) * m;
    for (int e = threadIdx.x; e < nr * m; e += kTiThreads) gp[e] += acc[e];
}

__global__ __launch_bounds__(kTiThreads) void three_interpolate_grad_atomic_kernel(int c, int n, int m,
                                                                                   const float *__restrict__ grad_out,
                                                                                   const int *__restrict__ idx,
                                                                                   const float *__restrict__ weight,
                                                                                   float *__restrict__ grad_points) {
    const int bs = blockIdx.z, ci = blockIdx.y;
    const int i = blockIdx.x * kTiThreads + threadIdx.x;
    if (i >= n) return;
    const float g = grad_out[((size_t)bs * c + ci) * n + i];
    const int *ix = idx + ((size_t)bs * n + i) * 3;
    const float *w = weight + ((size_t)bs * n + i) * 3;
    float *gp = grad_points + ((size_t)bs * c + ci) * m;
    atomicAdd(gp + ix[0], g * w[0]);
    atomicAdd(gp + ix[1], g * w[1]);
    atomicAdd(gp + ix[2], g * w[2]);
}

}  // namespace epnet

using namespace epnet;

extern "C" int epnet_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx,
                              epnet_stream_t stream) {
    EPNET_REQUIRE(b >= 0 && n >= 0 && m >= 0);
    if (b == 0 || n == 0) return EPNET_OK;
    EPNET_REQUIRE(unknown && dist2 && idx && (known || m == 0));
    EPNET_REQUIRE(b <= 65535);
    dim3 grid(div_up(n, kNnPerBlock), b);
    hipLaunchKernelGGL(three_nn_kernel, grid, dim3(kNnThreads), 0, (hipStream_t)stream, n, m, unknown, known, dist2, idx);
    return check_launch("three_nn");
}

extern "C" int epnet_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                                       const float *weight, float *out, epnet_stream_t stream) {
    EPNET_REQUIRE(b >= 0 && c >= 0 && m >= 0 && n >= 0);
    if (b == 0 || c == 0 || n == 0) return EPNET_OK;
    EPNET_REQUIRE(points && idx && weight && out);
    if (b > 65535 || div_up(c, kTiChan) > 65535) return EPNET_ELIMIT;
    dim3 grid(div_up(div_up(n, 4), kTiThreads), div_up(c, kTiChan), b);
    hipLaunchKernelGGL(three_interpolate_kernel, grid, dim3(kTiThreads), 0, (hipStream_t)stream, c, m, n, points, idx,
                       weight, out);
    return check_launch("three_interpolate");
}

extern "C" int epnet_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                            const float *weight, float *grad_points, epnet_stream_t stream) {
    EPNET_REQUIRE(b >= 0 && c >= 0 && m >= 0 && n >= 0);
    if (b == 0 || c == 0 || n == 0 || m == 0) return EPNET_OK;
    EPNET_REQUIRE(grad_out && idx && weight && grad_points);
    if (b > 65535) return EPNET_ELIMIT;
    constexpr int kLdsBudget = 64 * 1024;
    hipStream_t s = (hipStream_t)stream;
    if ((size_t)m * 4 <= kLdsBudget) {
        int rows = kLdsBudget / (m * 4);
        if (rows > 8) rows = 8;
        if (rows > c) rows = c;
        dim3 grid(div_up(c, rows), b);
        hipLaunchKernelGGL(three_interpolate_grad_lds_kernel, grid, dim3(kTiThreads), (size_t)rows * m * 4, s, c, n, m,
                           rows, grad_out, idx, weight, grad_points);
    } else {
        if (c > 65535) return EPNET_ELIMIT;
        dim3 grid(div_up(n, kTiThreads), c, b);
        hipLaunchKernelGGL(three_interpolate_grad_atomic_kernel, grid, dim3(kTiThreads), 0, s, c, n, m, grad_out, idx,
                           weight, grad_points);
    }
    return check_launch("three_interpolate_grad");
}
