// group.hip -- index gathers (gather_points, group_points) and their scatter-add gradients.
//
// Replaces gather_points{,_grad}_kernel_fast (pointnet2_lib/pointnet2/src/sampling_gpu.cu:8-83) and
// group_points{,_grad}_kernel_fast (group_points_gpu.cu:8-86). Both forward ops are the same 1-D
// gather per channel row: out[b,c,q] = points[b,c,idx[b,q]], q over M (gather) or M*nsample
// (group). These are the HBM-bound kernels of the SA stack (95 % of its compulsory bytes):
//
//   forward : a thread owns 4 consecutive output positions, loads their 4 indices ONCE (16-B
//             load) and reuses them for a chunk of channels -- the reference re-reads idx for
//             every channel (grid.y = C) -- and writes 16-B coalesced stores; the random 4-B
//             reads hit a channel row of N*4 B that stays in L1/L2.
//   backward: the reference issues one global atomicAdd per element. Here a workgroup owns
//             (scene, a few channel rows), accumulates the row in LDS with ds_add_f32 and adds it
//             to grad_points with plain coalesced stores: no global atomics, no cross-workgroup
//             races. Summation order within a row is still unspecified (as it is in the
//             reference), so gradients are compared to 1e-5, not bit-for-bit.
#include <stdlib.h>

#include "common.h"
#include "spatial.h"
#include "runsum.h"

namespace epnet {

constexpr int kGThreads = 256;
constexpr int kGChan = 16;  // channels per workgroup in the forward gather

// vectorised: p % 4 == 0 (so every row of idx/out is 16-B aligned given 16-B aligned bases)
// `ostride`: elements between consecutive scenes of `out` (c*p for a dense output; larger when the rows land
// in a channel slice of a wider tensor, see epnet_group_concat)
__global__ __launch_bounds__(kGThreads) void gather_rows_vec4_kernel(int c, int n, int p, size_t ostride,
                                                                     const float *__restrict__ points,
                                                                     const int *__restrict__ idx,
                                                                     float *__restrict__ out) {
    int wg_x, wg_y, bs;
    xcd_scene_map3(wg_x, wg_y, bs);   // a scene's rows and indices pass through one XCD's L2
    const int c0 = wg_y * kGChan;
    const int q4 = wg_x * kGThreads + threadIdx.x;  // group of 4 positions
    if (q4 * 4 >= p) return;
    const int4 id = *reinterpret_cast<const int4 *>(idx + (size_t)bs * p + (size_t)q4 * 4);
    const int cend = min(c, c0 + kGChan);
    const float *src = points + ((size_t)bs * c + c0) * n;
    float *dst = out + (size_t)bs * ostride + (size_t)c0 * p + (size_t)q4 * 4;
#pragma unroll 4
    for (int ci = c0; ci < cend; ++ci) {
        float4 v;
        v.x = src[id.x];
        v.y = src[id.y];
        v.z = src[id.z];
        v.w = src[id.w];
        *reinterpret_cast<float4 *>(dst) = v;  // plain store: streaming stores here slowed the kernel that follows
        src += n;
        dst += p;
    }
}

__global__ __launch_bounds__(kGThreads) void gather_rows_scalar_kernel(int c, int n, int p, size_t ostride,
                                                                       const float *__restrict__ points,
                                                                       const int *__restrict__ idx,
                                                                       float *__restrict__ out) {
    int wg_x, wg_y, bs;
    xcd_scene_map3(wg_x, wg_y, bs);   // a scene's rows and indices pass through one XCD's L2
    const int c0 = wg_y * kGChan;
    const int q = wg_x * kGThreads + threadIdx.x;
    if (q >= p) return;
    const int id = idx[(size_t)bs * p + q];
    const int cend = min(c, c0 + kGChan);
    for (int ci = c0; ci < cend; ++ci)
        out[(size_t)bs * ostride + (size_t)ci * p + q] = points[((size_t)bs * c + ci) * n + id];
}

// LDS-staged gather: a workgroup copies R whole channel rows (R*n floats) of one scene into LDS with
// coalesced 16-byte loads, then serves a tile of output positions out of LDS -- the random 4-byte reads
// become ds_read_b32 (bank conflicts cost a few cycles) instead of 64 scattered L1/L2 requests per wave
// instruction, and the kernel runs at the rate of its 16-byte coalesced stores. Measured on random
// indices (C=96, N=4096, P=32768, B=64): direct gather 1.9 TB/s, this kernel see profiles/.
// grid: (tiles, row chunks, scenes); dynamic LDS: R * n floats; p % 4 == 0.
#ifndef EPNET_GATHER_STAGE_UNROLL
#define EPNET_STAGE_PRAGMA
#else
#define EPNET_STAGE_STR2(x) #x
#define EPNET_STAGE_STR(x) EPNET_STAGE_STR2(unroll x)
#define EPNET_STAGE_PRAGMA _Pragma(EPNET_STAGE_STR(EPNET_GATHER_STAGE_UNROLL))
#endif
#ifndef EPNET_GATHER_LDS_BUDGET_KB
#define EPNET_GATHER_LDS_BUDGET_KB 64   // LDS per row-gather workgroup: two workgroups per CU (32: 3.91, 128: 4.63 ms per 256-scene step against 3.78)
#endif
#ifndef EPNET_GATHER_LDS_THREADS
#define EPNET_GATHER_LDS_THREADS 256
#endif
// rows in flight per thread in the serving loop (4 LDS reads each). Alone on the device the kernel does not care (1: 5.07-5.22,
// 4: 5.11-5.22 TB/s), but in the software-pipelined SA stack it shares every CU with the sampling kernels, whose rounds wait
// on four dependent LDS operations: the fewer gather reads queue in front of those, the shorter the round. 256-scene step:
// unroll 8 4.10, 4 3.97, 2 3.86, 1 3.82 ms
#ifndef EPNET_GATHER_LDS_UNROLL
#define EPNET_GATHER_LDS_UNROLL 1
#endif
constexpr int kGLdsThreads = EPNET_GATHER_LDS_THREADS;
// ---- four channel rows interleaved per LDS word ("quad" staging) ----------------------------------------------------
// [n][4] floats per group of four rows: one 16-byte LDS read then serves four channel rows of one position -- a quarter of
// the LDS instructions and less than half the LDS cycles of the row-major staging (random 4-byte reads: two 32-lane groups
// over 32 banks; random 16-byte reads: four 16-lane groups over 16 bank quads). The gather shares every CU with the
// sampling rounds of the pipelined stack, which wait on dependent LDS operations: what the gather does not put into the
// LDS queue shortens those rounds.
__device__ __forceinline__ void stage_quads(const float *__restrict__ src, int groups, int n, float4 *__restrict__ s_quad,
                                            int threads) {
    const int n4 = n >> 2;
    for (int e = threadIdx.x; e < groups * n4; e += threads) {
        const int g = e / n4, j4 = e - g * n4;
        const float4 *row = reinterpret_cast<const float4 *>(src + (size_t)g * 4 * n) + j4;
        const float4 a = row[0], b4 = row[n4], cc = row[2 * n4], d = row[3 * n4];
        float4 *dst = s_quad + (size_t)g * n + j4 * 4;
        dst[0] = make_float4(a.x, b4.x, cc.x, d.x);
        dst[1] = make_float4(a.y, b4.y, cc.y, d.y);
        dst[2] = make_float4(a.z, b4.z, cc.z, d.z);
        dst[3] = make_float4(a.w, b4.w, cc.w, d.w);
    }
}

// positions id.x..w of `groups` quad-staged row groups -> four channel rows of 16 bytes each per group
__device__ __forceinline__ void serve_quads(const float4 *__restrict__ quad, int groups, int n, int4 id, float *__restrict__ dst,
                                            int p) {
#pragma unroll 1
    for (int g = 0; g < groups; ++g) {
        const float4 v0 = quad[id.x], v1 = quad[id.y], v2 = quad[id.z], v3 = quad[id.w];
        store_stream(dst, v0.x, v1.x, v2.x, v3.x);
        store_stream(dst + p, v0.y, v1.y, v2.y, v3.y);
        store_stream(dst + 2 * (size_t)p, v0.z, v1.z, v2.z, v3.z);
        store_stream(dst + 3 * (size_t)p, v0.w, v1.w, v2.w, v3.w);
        quad += n;
        dst += 4 * (size_t)p;
    }
}

template <bool QUAD>  // QUAD: rows % 4 == 0 == c % 4 == n % 4, points 16-byte aligned
__global__ __launch_bounds__(kGLdsThreads) void gather_rows_lds_kernel(int c, int n, int p, int rows, int tile, size_t ostride,
                                                                    const float *__restrict__ points,
                                                                    const int *__restrict__ idx,
                                                                    float *__restrict__ out) {
    extern __shared__ float s_rows[];
    int wg_x, wg_y, bs;
    xcd_scene_map3(wg_x, wg_y, bs);   // a scene's rows and indices pass through one XCD's L2
    const int c0 = wg_y * rows;
    const int nr = min(rows, c - c0);
    const float *src = points + ((size_t)bs * c + c0) * n;
    const int total = nr * n;
    if (QUAD) {
        stage_quads(src, nr >> 2, n, reinterpret_cast<float4 *>(s_rows), kGLdsThreads);
    } else if ((n & 3) == 0 && ((uintptr_t)src & 15) == 0) {
        const float4 *src4 = reinterpret_cast<const float4 *>(src);
        float4 *dst4 = reinterpret_cast<float4 *>(s_rows);
        EPNET_STAGE_PRAGMA
        for (int e = threadIdx.x; e < total / 4; e += kGLdsThreads) dst4[e] = src4[e];
    } else {
        for (int e = threadIdx.x; e < total; e += kGLdsThreads) s_rows[e] = src[e];
    }
    __syncthreads();
    const int q_begin = wg_x * tile, q_end = min(p, q_begin + tile);
    const int *ix = idx + (size_t)bs * p;
    float *dst_base = out + (size_t)bs * ostride + (size_t)c0 * p;
    for (int q = q_begin + threadIdx.x * 4; q < q_end; q += kGLdsThreads * 4) {
        const int4 id = *reinterpret_cast<const int4 *>(ix + q);
        float *dst = dst_base + q;
        if (QUAD) {
            serve_quads(reinterpret_cast<const float4 *>(s_rows), nr >> 2, n, id, dst, p);
            continue;
        }
        const float *row = s_rows;
#pragma unroll EPNET_GATHER_LDS_UNROLL
        for (int r = 0; r < nr; ++r) {
            float4 v;
            v.x = row[id.x];
            v.y = row[id.y];
            v.z = row[id.z];
            v.w = row[id.w];
            store_stream(dst, v.x, v.y, v.z, v.w);
            row += n;
            dst += p;
        }
    }
}

#ifndef EPNET_GATHER_LDS2_UNROLL
#define EPNET_GATHER_LDS2_UNROLL 1   // see EPNET_GATHER_LDS_UNROLL
#endif
// the same for the TWO scales of an MSG level at once: both gather from the same feature rows, so the rows are
// staged once and then serve both index sets (saves one 64 KB staging pass per workgroup: 7-12 % of the traffic)
template <bool QUAD>
__global__ __launch_bounds__(kGLdsThreads) void gather_rows_lds2_kernel(int c, int n, int rows, const float *__restrict__ points,
                                                                        int p0, size_t ostride0, const int *__restrict__ idx0,
                                                                        float *__restrict__ out0, int p1, size_t ostride1,
                                                                        const int *__restrict__ idx1, float *__restrict__ out1) {
    extern __shared__ float s_rows[];
    int wg_x, bs;
    xcd_scene_map(wg_x, bs);   // the index tensors of a scene are read by all its row chunks: through one XCD's L2
    const int c0 = wg_x * rows;
    const int nr = min(rows, c - c0);
    const float *src = points + ((size_t)bs * c + c0) * n;
    const int total = nr * n;
    if (QUAD) {
        stage_quads(src, nr >> 2, n, reinterpret_cast<float4 *>(s_rows), kGLdsThreads);
    } else if ((n & 3) == 0 && ((uintptr_t)src & 15) == 0) {
        const float4 *src4 = reinterpret_cast<const float4 *>(src);
        float4 *dst4 = reinterpret_cast<float4 *>(s_rows);
        EPNET_STAGE_PRAGMA
        for (int e = threadIdx.x; e < total / 4; e += kGLdsThreads) dst4[e] = src4[e];
    } else {
        for (int e = threadIdx.x; e < total; e += kGLdsThreads) s_rows[e] = src[e];
    }
    __syncthreads();
#pragma unroll 1
    for (int set = 0; set < 2; ++set) {
        const int p = set ? p1 : p0;
        const int *ix = (set ? idx1 : idx0) + (size_t)bs * p;
        float *dst_base = (set ? out1 : out0) + (size_t)bs * (set ? ostride1 : ostride0) + (size_t)c0 * p;
        for (int q = threadIdx.x * 4; q < p; q += kGLdsThreads * 4) {
            const int4 id = *reinterpret_cast<const int4 *>(ix + q);
            float *dst = dst_base + q;
            if (QUAD) {
                serve_quads(reinterpret_cast<const float4 *>(s_rows), nr >> 2, n, id, dst, p);
                continue;
            }
            const float *row = s_rows;
#pragma unroll EPNET_GATHER_LDS2_UNROLL
            for (int r = 0; r < nr; ++r) {
                float4 v;
                v.x = row[id.x];
                v.y = row[id.y];
                v.z = row[id.z];
                v.w = row[id.w];
                store_stream(dst, v.x, v.y, v.z, v.w);
                row += n;
                dst += p;
            }
        }
    }
}

// scatter-add with the destination rows held in LDS. dynamic LDS: rows * n floats.
__global__ __launch_bounds__(kGThreads) void scatter_rows_lds_kernel(int c, int n, int p, int rows, size_t gstride,
                                                                     const float *__restrict__ grad_out,
                                                                     const int *__restrict__ idx,
                                                                     float *__restrict__ grad_points) {
    extern __shared__ float acc[];
    const int bs = blockIdx.y;
    const int c0 = blockIdx.x * rows;
    const int nr = min(rows, c - c0);
    for (int e = threadIdx.x; e < nr * n; e += kGThreads) acc[e] = 0.f;
    __syncthreads();
    const int *ix = idx + (size_t)bs * p;
    const float *go = grad_out + (size_t)bs * gstride + (size_t)c0 * p;
    for (int q = threadIdx.x; q < p; q += kGThreads) {
        const int id = ix[q];
        for (int r = 0; r < nr; ++r) atomicAdd(&acc[r * n + id], go[(size_t)r * p + q]);
    }
    __syncthreads();
    float *gp = grad_points + ((size_t)bs * c + c0) * n;
    for (int e = threadIdx.x; e < nr * n; e += kGThreads) gp[e] += acc[e];
}

// fallback for rows that do not fit LDS: global float atomics, as the reference does
__global__ __launch_bounds__(kGThreads) void scatter_rows_atomic_kernel(int c, int n, int p, size_t gstride,
                                                                        const float *__restrict__ grad_out,
                                                                        const int *__restrict__ idx,
                                                                        float *__restrict__ grad_points) {
    const int bs = blockIdx.z, ci = blockIdx.y;
    const int q = blockIdx.x * kGThreads + threadIdx.x;
    if (q >= p) return;
    atomicAdd(grad_points + ((size_t)bs * c + ci) * n + idx[(size_t)bs * p + q],
              grad_out[(size_t)bs * gstride + (size_t)ci * p + q]);
}

// grouped, centre-relative coordinates straight from the (B,N,3) layout:
// out[b, ch, i, s] = xyz[b, idx[b,i,s], ch] - new_xyz[b, i, ch], ch = 0..2  (pointnet2_utils.py:250-252)
__global__ __launch_bounds__(kGThreads) void group_xyz_centred_kernel(int n, int npoints, int nsample, size_t ostride,
                                                                      const float *__restrict__ xyz,
                                                                      const float *__restrict__ new_xyz,
                                                                      const int *__restrict__ idx, float *__restrict__ out) {
    int wg_x, bs;
    xcd_scene_map(wg_x, bs);   // a scene's coordinates pass through one XCD's L2
    const int p = npoints * nsample;
    const int q = wg_x * kGThreads + threadIdx.x;
    if (q >= p) return;
    const int id = idx[(size_t)bs * p + q];
    const int ci = q / nsample;
    const float *pt = xyz + ((size_t)bs * n + id) * 3;
    const float *ce = new_xyz + ((size_t)bs * npoints + ci) * 3;
    float *dst = out + (size_t)bs * ostride + q;
    dst[0] = pt[0] - ce[0];
    dst[(size_t)p] = pt[1] - ce[1];
    dst[(size_t)p * 2] = pt[2] - ce[2];
}

// the same for 4 consecutive samples of one centre per thread (nsample % 4 == 0): one 16-B index load, three 16-B stores
__global__ __launch_bounds__(kGThreads) void group_xyz_centred_vec4_kernel(int n, int npoints, int nsample, size_t ostride,
                                                                           const float *__restrict__ xyz,
                                                                           const float *__restrict__ new_xyz,
                                                                           const int *__restrict__ idx,
                                                                           float *__restrict__ out) {
    int wg_x, bs;
    xcd_scene_map(wg_x, bs);   // a scene's coordinates pass through one XCD's L2
    const int p = npoints * nsample;
    const int q = (wg_x * kGThreads + threadIdx.x) * 4;
    if (q >= p) return;
    const int4 id = *reinterpret_cast<const int4 *>(idx + (size_t)bs * p + q);
    const float *base = xyz + (size_t)bs * n * 3;
    const float *ce = new_xyz + ((size_t)bs * npoints + q / nsample) * 3;
    const float cx = ce[0], cy = ce[1], cz = ce[2];
    const float *p0 = base + (size_t)id.x * 3, *p1 = base + (size_t)id.y * 3, *p2 = base + (size_t)id.z * 3,
                *p3 = base + (size_t)id.w * 3;
    float *dst = out + (size_t)bs * ostride + q;
    *reinterpret_cast<float4 *>(dst) = make_float4(p0[0] - cx, p1[0] - cx, p2[0] - cx, p3[0] - cx);
    *reinterpret_cast<float4 *>(dst + (size_t)p) = make_float4(p0[1] - cy, p1[1] - cy, p2[1] - cy, p3[1] - cy);
    *reinterpret_cast<float4 *>(dst + (size_t)p * 2) = make_float4(p0[2] - cz, p1[2] - cz, p2[2] - cz, p3[2] - cz);
}

// ---- first shared-MLP layer folded into the grouping -----------------------------------------
// An SA level feeds [xyz[idx] - centre ; features[idx]] to a 1x1 convolution (pointnet2_utils.py:250-257 ->
// pointnet2_modules.py:61). A 1x1 convolution is linear and pointwise, so it commutes with the gather of its feature
// columns:  W . [dxyz ; F[:, idx]] = W_xyz . dxyz + (W_f . F)[:, idx].  The caller multiplies the N feature columns once
// (z = W_f . F, a dense GEMM on N instead of npoint * nsample columns) and this kernel writes the layer's
// pre-activations directly:
//     out[b, co, m, s] = z[b, co, idx[b, m, s]] + (wx[co][0] * dx + wx[co][1] * dy + wx[co][2] * dz) (+ bias[co]),
//     (dx, dy, dz) = xyz[b, idx[b, m, s]] - new_xyz[b, m]
// (products and sums in exactly this order, no contraction). The (3 + C_in, npoint, nsample) grouped tensor is never
// materialised, neither forwards nor for the backward. Same staging as gather_rows_lds_kernel: `rows` rows of z in LDS.
__global__ __launch_bounds__(kGLdsThreads) void group_linear_lds_kernel(int c, int n, int npoints, int nsample, int rows, int tile,
                                                                        const float *__restrict__ z, const float *__restrict__ xyz,
                                                                        const float *__restrict__ new_xyz, const int *__restrict__ idx,
                                                                        const float *__restrict__ wx, const float *__restrict__ bias,
                                                                        float *__restrict__ out) {
    extern __shared__ float s_rows[];
    const int bs = blockIdx.z;
    const int c0 = blockIdx.y * rows;
    const int nr = min(rows, c - c0);
    const int p = npoints * nsample;
    const float *src = z + ((size_t)bs * c + c0) * n;
    const int total = nr * n;
    if ((n & 3) == 0 && ((uintptr_t)src & 15) == 0) {
        const float4 *src4 = reinterpret_cast<const float4 *>(src);
        float4 *dst4 = reinterpret_cast<float4 *>(s_rows);
        EPNET_STAGE_PRAGMA
        for (int e = threadIdx.x; e < total / 4; e += kGLdsThreads) dst4[e] = src4[e];
    } else {
        for (int e = threadIdx.x; e < total; e += kGLdsThreads) s_rows[e] = src[e];
    }
    __syncthreads();
    const int q_begin = blockIdx.x * tile, q_end = min(p, q_begin + tile);
    const int *ix = idx + (size_t)bs * p;
    const float *pts = xyz + (size_t)bs * n * 3;
    float *dst_base = out + ((size_t)bs * c + c0) * p;
    for (int q = q_begin + threadIdx.x * 4; q < q_end; q += kGLdsThreads * 4) {
        const int4 id = *reinterpret_cast<const int4 *>(ix + q);
        const float *ce = new_xyz + ((size_t)bs * npoints + q / nsample) * 3;  // nsample % 4 == 0: one centre per thread
        const float cx = ce[0], cy = ce[1], cz = ce[2];
        const float *p0 = pts + (size_t)id.x * 3, *p1 = pts + (size_t)id.y * 3, *p2 = pts + (size_t)id.z * 3, *p3 = pts + (size_t)id.w * 3;
        const float dx0 = p0[0] - cx, dy0 = p0[1] - cy, dz0 = p0[2] - cz;
        const float dx1 = p1[0] - cx, dy1 = p1[1] - cy, dz1 = p1[2] - cz;
        const float dx2 = p2[0] - cx, dy2 = p2[1] - cy, dz2 = p2[2] - cz;
        const float dx3 = p3[0] - cx, dy3 = p3[1] - cy, dz3 = p3[2] - cz;
        float *dst = dst_base + q;
        const float *row = s_rows;
#pragma unroll 2
        for (int r = 0; r < nr; ++r) {
            const float w0 = wx[(c0 + r) * 3], w1 = wx[(c0 + r) * 3 + 1], w2 = wx[(c0 + r) * 3 + 2];  // wave-uniform
            float4 v;
            v.x = row[id.x] + (w0 * dx0 + w1 * dy0 + w2 * dz0);
            v.y = row[id.y] + (w0 * dx1 + w1 * dy1 + w2 * dz1);
            v.z = row[id.z] + (w0 * dx2 + w1 * dy2 + w2 * dz2);
            v.w = row[id.w] + (w0 * dx3 + w1 * dy3 + w2 * dz3);
            if (bias) {
                const float bv = bias[c0 + r];
                v.x += bv; v.y += bv; v.z += bv; v.w += bv;
            }
            store_stream(dst, v.x, v.y, v.z, v.w);
            row += n;
            dst += p;
        }
    }
}

// any shape: one thread per position, kGChan channels per block row, z read through L1 / L2
__global__ __launch_bounds__(kGThreads) void group_linear_scalar_kernel(int c, int n, int npoints, int nsample,
                                                                        const float *__restrict__ z, const float *__restrict__ xyz,
                                                                        const float *__restrict__ new_xyz, const int *__restrict__ idx,
                                                                        const float *__restrict__ wx, const float *__restrict__ bias,
                                                                        float *__restrict__ out) {
    const int bs = blockIdx.z;
    const int c0 = blockIdx.y * kGChan;
    const int p = npoints * nsample;
    const int q = blockIdx.x * kGThreads + threadIdx.x;
    if (q >= p) return;
    const int id = idx[(size_t)bs * p + q];
    const float *pt = xyz + ((size_t)bs * n + id) * 3;
    const float *ce = new_xyz + ((size_t)bs * npoints + q / nsample) * 3;
    const float dx = pt[0] - ce[0], dy = pt[1] - ce[1], dz = pt[2] - ce[2];
    const int cend = min(c, c0 + kGChan);
    for (int ci = c0; ci < cend; ++ci) {
        float v = z[((size_t)bs * c + ci) * n + id] + (wx[ci * 3] * dx + wx[ci * 3 + 1] * dy + wx[ci * 3 + 2] * dz);
        if (bias) v += bias[ci];
        out[((size_t)bs * c + ci) * p + q] = v;
    }
}

// gradient of group_linear w.r.t. w_xyz: grad_w[co][k] += sum over (b, m, s) of grad_out[b,co,m,s] * (xyz[b,idx] - new_xyz[b,m])[k].
// The centred coordinates are rebuilt per position (never stored); a block owns kGwRows channel rows x a tile of positions
// of one scene, every thread keeps kGwRows x 3 partial sums, reduced by shuffles + LDS and added to grad_w with 3 * kGwRows
// float atomics per block (a dense-product formulation has K = b * npoint * nsample and 3 output columns: rocBLAS takes
// 4.4 ms for the 537 MB of the RCNN stage's first level; this kernel reads them once).
constexpr int kGwRows = 8;
__global__ __launch_bounds__(kGThreads) void group_linear_grad_w_kernel(int c, int n, int npoints, int nsample, int tile,
                                                                        const float *__restrict__ grad_out,
                                                                        const float *__restrict__ xyz,
                                                                        const float *__restrict__ new_xyz,
                                                                        const int *__restrict__ idx, float *__restrict__ grad_w) {
    __shared__ float s_part[kGThreads / 64][kGwRows * 3];
    const int bs = blockIdx.z;
    const int c0 = blockIdx.y * kGwRows;
    const int nr = min(kGwRows, c - c0);
    const int p = npoints * nsample;
    const int q_begin = blockIdx.x * tile, q_end = min(p, q_begin + tile);
    const int *ix = idx + (size_t)bs * p;
    const float *pts = xyz + (size_t)bs * n * 3;
    const float *g = grad_out + ((size_t)bs * c + c0) * p;
    float acc[kGwRows][3];
#pragma unroll
    for (int r = 0; r < kGwRows; ++r) acc[r][0] = acc[r][1] = acc[r][2] = 0.f;
    // (a full chunk of rows reads them without a guard: a load under `r < nr` is waited for before the next one is issued --
    // eight serial round trips per position; the last, partial chunk of a channel count that is no multiple of kGwRows reads
    // its last row again instead and drops the products)
    for (int q = q_begin + threadIdx.x; q < q_end; q += kGThreads) {
        const int id = ix[q];
        const float *pt = pts + (size_t)id * 3;
        const float *ce = new_xyz + ((size_t)bs * npoints + q / nsample) * 3;
        float gv[kGwRows];
#pragma unroll
        for (int r = 0; r < kGwRows; ++r) gv[r] = g[(size_t)min(r, nr - 1) * p + q];
        const float dx = pt[0] - ce[0], dy = pt[1] - ce[1], dz = pt[2] - ce[2];
#pragma unroll
        for (int r = 0; r < kGwRows; ++r) {
            const float gr = r < nr ? gv[r] : 0.f;
            acc[r][0] += gr * dx;
            acc[r][1] += gr * dy;
            acc[r][2] += gr * dz;
        }
    }
    const int lane = lane_id(), wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < kGwRows; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float v = wave_sum_f32(acc[r][k]);
            if (lane == 0) s_part[wave][r * 3 + k] = v;
        }
    __syncthreads();
    if (threadIdx.x < nr * 3) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < kGThreads / 64; ++w) v += s_part[w][threadIdx.x];
        atomicAdd(grad_w + (size_t)c0 * 3 + threadIdx.x, v);
    }
}

// ---- rows too long for LDS (n * 4 > 64 KB: the 65536-point scenes of BASELINE config 5) -----------------------------
// The channel-major gather above then reads 4 bytes per lane from 64 different cache lines per instruction and is bound by
// the texture addresser (2.2 TB/s on C = 64, N = 65536). With caller scratch the features are turned point-major once
// (b, n, c): a position's c channels are one contiguous row, fetched with 16-byte loads into an LDS tile of 256 positions
// and written back out transposed, 256 contiguous bytes per channel row and wave. Extra traffic: the features once more
// (c * n * 8 bytes per scene against c * p * 4 of output, p = npoints * nsample >> n).
constexpr int kPmThreads = 256;
constexpr int kPmTile = 128;  // positions per workgroup: 32 KB of LDS at c = 64, four workgroups per CU

__global__ __launch_bounds__(256) void transpose_cn_kernel(int c, int n, const float *__restrict__ src, float *__restrict__ dst) {
    __shared__ float tile[64][65];
    const int bs = blockIdx.z, n0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    src += (size_t)bs * c * n;
    dst += (size_t)bs * c * n;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4)
        if (c0 + r < c && n0 + tx < n) tile[r][tx] = src[(size_t)(c0 + r) * n + n0 + tx];
    __syncthreads();
    for (int r = ty; r < 64; r += 4)
        if (n0 + r < n && c0 + tx < c) dst[(size_t)(n0 + r) * c + c0 + tx] = tile[tx][r];
}

// points_t (b, n, c) point-major; out rows (b, c, p) with row stride ostride between scenes; c % 4 == 0, 16 <= c <= 128.
// LDS tile [c][kPmTile], channel-major, with the 16-byte groups of a channel row swizzled by the channel
// (position r of channel ch sits at r ^ (((ch >> 2) & 7) << 2)): the staging writes of a wave (4 rows x 16 lanes x 4 channels)
// fall on 32 banks two at a time (free for ds_write_b32), and the read-back is one conflict-free ds_read_b128 per lane --
// four consecutive positions of one channel, stored with ONE 16-byte streaming store (the 4-byte stores of the first
// version were store-issue bound: 256 B per wave-instruction).
__device__ __forceinline__ int pm_slot(int ch, int r) { return ch * kPmTile + (r ^ (((ch >> 2) & 7) << 2)); }

__global__ __launch_bounds__(kPmThreads) void gather_rows_pm_kernel(int c, int n, int p, size_t ostride,
                                                                    const float *__restrict__ points_t,
                                                                    const int *__restrict__ idx, float *__restrict__ out) {
    extern __shared__ float s_tile[];  // [c][kPmTile]
    int wg_x, bs;
    xcd_scene_map(wg_x, bs);   // an XCD gathers from ONE scene's point-major copy at a time (a quarter of it fits its L2) instead of from all the chunk's
    const int q0 = wg_x * kPmTile;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lanes_per_row = c >> 2, rows_per_pass = kPmThreads / lanes_per_row;
    points_t += (size_t)bs * n * c;
    idx += (size_t)bs * p;
    out += (size_t)bs * ostride;
    const int cnt = min(kPmTile, p - q0);
    const int r_in = t / lanes_per_row, l_in = t - r_in * lanes_per_row;
    // a thread's rows r_in, r_in + rows_per_pass, ...: all indices first, then all rows in flight together (kMaxPass x 16 bytes)
    constexpr int kMaxPass = 8;   // c >= 16: at least 64 rows per pass of 256 threads... 128 / 16 = 8 passes at c = 64
    if (r_in < rows_per_pass) {
        for (int r0 = r_in; r0 < cnt; r0 += rows_per_pass * kMaxPass) {
            int j[kMaxPass];
            float4 v[kMaxPass];
#pragma unroll
            for (int k = 0; k < kMaxPass; ++k) {
                const int r = r0 + k * rows_per_pass;
                j[k] = r < cnt ? idx[q0 + r] : 0;
            }
#pragma unroll
            for (int k = 0; k < kMaxPass; ++k) v[k] = reinterpret_cast<const float4 *>(points_t + (size_t)j[k] * c)[l_in];
#pragma unroll
            for (int k = 0; k < kMaxPass; ++k) {
                const int r = r0 + k * rows_per_pass;
                if (r < cnt) {
                    const int ch = l_in * 4;   // four channels of one swizzle group: the same position offset for all four
                    float *d = s_tile + pm_slot(ch, r);
                    d[0] = v[k].x; d[kPmTile] = v[k].y; d[2 * kPmTile] = v[k].z; d[3 * kPmTile] = v[k].w;
                }
            }
        }
    }
    __syncthreads();
    const bool wide = cnt == kPmTile && (p & 3) == 0 && (((uintptr_t)out) & 15) == 0;   // (workgroup-uniform)
    if (wide) {
        // a wave-instruction covers two channel rows: lanes 0-31 the 128 positions of channel ch, lanes 32-63 those of ch + 1
        const int half = lane >> 5, l = lane & 31;
        for (int ch = wave * 2 + half; ch < c; ch += (kPmThreads / 64) * 2) {
            const float4 v = *reinterpret_cast<const float4 *>(s_tile + pm_slot(ch, l * 4));
            store_stream(out + (size_t)ch * p + q0 + l * 4, v.x, v.y, v.z, v.w);
        }
    } else {
        for (int ch = wave; ch < c; ch += kPmThreads / 64) {
            float *dst = out + (size_t)ch * p + q0;
#pragma unroll
            for (int k = 0; k < kPmTile / 64; ++k) {
                const int r = k * 64 + lane;
                if (r < cnt) __builtin_nontemporal_store(s_tile[pm_slot(ch, r)], dst + r);
            }
        }
    }
}

static size_t gather_pm_ws_bytes(int b, int c, int n, long long p) {
    constexpr int kLdsBudget = EPNET_GATHER_LDS_BUDGET_KB * 1024;
    if (b <= 0 || c < 16 || c > 128 || (c & 3) || (size_t)n * 4 <= (size_t)kLdsBudget || p < (long long)n) return 0;  // (few positions: the copy would cost more than it saves)
    return (size_t)b * c * n * sizeof(float);
}

static int launch_gather_rows_pm(int b, int c, int n, long long p, const float *points, const int *idx, float *out, size_t ostride,
                                 void *workspace, hipStream_t s, const char *what) {
    if (b > 65535 || p > 0x7fffffffll) return EPNET_ELIMIT;
    float *pt = (float *)workspace;
    // a few scenes at a time, so that the point-major copies are still on the chip (L2 / Infinity Cache) when they are gathered
    // (16 scenes of C = 64, N = 65536 at once: 268 MB of copies, 3.7 TB/s; one scene: 4.0 TB/s)
    int chunk = (int)((64ll << 20) / ((long long)c * n * 4));
    if (chunk < 1) chunk = 1;
    for (int b0 = 0; b0 < b; b0 += chunk) {
        const int nb = min(chunk, b - b0);
        const float *src = points + (size_t)b0 * c * n;
        float *dst = pt + (size_t)b0 * c * n;
        hipLaunchKernelGGL(transpose_cn_kernel, dim3(div_up(n, 64), div_up(c, 64), nb), dim3(256), 0, s, c, n, src, dst);
        int rc = check_launch("feature transpose");
        if (rc) return rc;
        hipLaunchKernelGGL(gather_rows_pm_kernel, dim3((unsigned)div_up64(p, kPmTile), nb), dim3(kPmThreads),
                           (size_t)kPmTile * c * sizeof(float), s, c, n, (int)p, ostride, dst, idx + (size_t)b0 * p,
                           out + (size_t)b0 * ostride);
        rc = check_launch(what);
        if (rc) return rc;
    }
    return EPNET_OK;
}

static int launch_gather_rows(int b, int c, int n, long long p, const float *points, const int *idx, float *out,
                              hipStream_t s, const char *what, size_t ostride = 0) {
    if (ostride == 0) ostride = (size_t)c * (size_t)p;
    if (b == 0 || c == 0 || p == 0) return EPNET_OK;
    if (!(points && idx && out)) return EPNET_EINVAL;
    if (n <= 0) return EPNET_EINVAL;  // positions to fill from an empty cloud: no index can be valid
    if (p > 0x7fffffffll || b > 65535 || div_up(c, kGChan) > 65535) return EPNET_ELIMIT;
    const bool vec = (p % 4 == 0) && (((uintptr_t)idx | (uintptr_t)out) % 16 == 0);
    constexpr int kLdsBudget = EPNET_GATHER_LDS_BUDGET_KB * 1024;  // two workgroups per CU
    if (vec && c >= 8 && (size_t)n * 4 <= kLdsBudget && p >= 2048) {  // few channels (xyz): the rows stay in L2 anyway
        int rows = kLdsBudget / (n * 4);
        if (rows > c) rows = c;
        if (rows > 32) rows = 32;
        const bool quad = (c & 3) == 0 && (n & 3) == 0 && rows >= 4 && ((uintptr_t)points & 15) == 0 && !getenv("EPNET_GATHER_NO_QUAD");
        if (quad) rows &= ~3;  // whole groups of four channel rows
        const int chunks = div_up(c, rows);
        // enough workgroups to fill the chip; every tile re-stages its rows, so keep tiles >= 2048 positions
        int tiles = div_up(1024, b * chunks);
        const int max_tiles = (int)(p / 2048);
        if (tiles > max_tiles) tiles = max_tiles;
        if (tiles < 1) tiles = 1;
        int tile = (int)div_up64(p, tiles);
        tile = (tile + kGLdsThreads * 4 - 1) / (kGLdsThreads * 4) * (kGLdsThreads * 4);  // whole passes of the workgroup (threads x 4 positions)
        tiles = (int)div_up64(p, tile);
        if (chunks <= 65535) {
            dim3 grid(tiles, chunks, b);
            if (quad)
                hipLaunchKernelGGL(gather_rows_lds_kernel<true>, grid, dim3(kGLdsThreads), (size_t)rows * n * 4, s, c, n, (int)p, rows,
                                   tile, ostride, points, idx, out);
            else
                hipLaunchKernelGGL(gather_rows_lds_kernel<false>, grid, dim3(kGLdsThreads), (size_t)rows * n * 4, s, c, n, (int)p, rows,
                                   tile, ostride, points, idx, out);
            return check_launch(what);
        }
    }
    if (vec) {
        dim3 grid((unsigned)div_up64(p / 4, kGThreads), div_up(c, kGChan), b);
        hipLaunchKernelGGL(gather_rows_vec4_kernel, grid, dim3(kGThreads), 0, s, c, n, (int)p, ostride, points, idx, out);
    } else {
        dim3 grid((unsigned)div_up64(p, kGThreads), div_up(c, kGChan), b);
        hipLaunchKernelGGL(gather_rows_scalar_kernel, grid, dim3(kGThreads), 0, s, c, n, (int)p, ostride, points, idx, out);
    }
    return check_launch(what);
}

// ---- scatter-add without atomics (runsum.h): the positions grouped by target once, then equal shares of the sorted
// entries summed per thread out of LDS-staged grad_out rows
static size_t scatter_ws_bytes(int b, int n, long long p) {
    if (b <= 0 || !runsum::usable(n, 1, p, p)) return 0;
    return runsum::workspace_bytes(b, n, 1, p, false);
}

static int launch_scatter_rows_csr(int b, int c, int n, long long p, const float *grad_out, const int *idx,
                                   float *grad_points, void *workspace, size_t workspace_bytes, hipStream_t s,
                                   const char *what, size_t gstride) {
    if (gstride == 0) gstride = (size_t)c * (size_t)p;
    if (b == 0 || c == 0 || p == 0 || n == 0) return EPNET_OK;
    if (!(grad_out && idx && grad_points && workspace)) return EPNET_EINVAL;
    return runsum::launch<false>(b, c, n, 1, (int)p, grad_out, gstride, idx, nullptr, grad_points, workspace, workspace_bytes, s,
                                 what);
}

static int launch_scatter_rows(int b, int c, int n, long long p, const float *grad_out, const int *idx,
                               float *grad_points, hipStream_t s, const char *what, size_t gstride = 0) {
    if (gstride == 0) gstride = (size_t)c * (size_t)p;
    if (b == 0 || c == 0 || p == 0 || n == 0) return EPNET_OK;
    if (!(grad_out && idx && grad_points)) return EPNET_EINVAL;
    if (p > 0x7fffffffll || b > 65535) return EPNET_ELIMIT;
    constexpr int kLdsBudget = EPNET_GATHER_LDS_BUDGET_KB * 1024;  // two workgroups per CU
    if ((size_t)n * 4 <= kLdsBudget) {
        int rows = kLdsBudget / (n * 4);
        if (rows > 8) rows = 8;
        if (rows > c) rows = c;
        dim3 grid(div_up(c, rows), b);
        hipLaunchKernelGGL(scatter_rows_lds_kernel, grid, dim3(kGThreads), (size_t)rows * n * 4, s, c, n, (int)p, rows, gstride,
                           grad_out, idx, grad_points);
    } else {
        if (c > 65535) return EPNET_ELIMIT;
        dim3 grid((unsigned)div_up64(p, kGThreads), c, b);
        hipLaunchKernelGGL(scatter_rows_atomic_kernel, grid, dim3(kGThreads), 0, s, c, n, (int)p, gstride, grad_out, idx,
                           grad_points);
    }
    return check_launch(what);
}

}  // namespace epnet

using namespace epnet;

extern "C" int epnet_gather_points(int b, int c, int n, int npoints, const float *points, const int *idx, float *out,
                                   epnet_stream_t stream) {
    EPNET_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0);
    return launch_gather_rows(b, c, n, npoints, points, idx, out, (hipStream_t)stream, "gather_points");
}

extern "C" int epnet_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out, const int *idx,
                                        float *grad_points, epnet_stream_t stream) {
    EPNET_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0);
    return launch_scatter_rows(b, c, n, npoints, grad_out, idx, grad_points, (hipStream_t)stream, "gather_points_grad");
}

extern "C" int epnet_group_points(int b, int c, int n, int npoints, int nsample, const float *points, const int *idx,
                                  float *out, epnet_stream_t stream) {
    EPNET_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0);
    return launch_gather_rows(b, c, n, (long long)npoints * nsample, points, idx, out, (hipStream_t)stream,
                              "group_points");
}

extern "C" int epnet_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                                       const int *idx, float *grad_points, epnet_stream_t stream) {
    EPNET_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0);
    return launch_scatter_rows(b, c, n, (long long)npoints * nsample, grad_out, idx, grad_points, (hipStream_t)stream,
                               "group_points_grad");
}

// the three centred coordinate rows of a grouped tensor whose scenes are ostride floats apart
static int launch_group_xyz(int b, int n, int npoints, int nsample, long long p, size_t ostride, const float *xyz,
                            const float *new_xyz, const int *idx, float *out, hipStream_t s) {
    if (nsample % 4 == 0 && (((uintptr_t)idx | (uintptr_t)out) % 16 == 0) && (ostride % 4 == 0))
        hipLaunchKernelGGL(group_xyz_centred_vec4_kernel, dim3((unsigned)div_up64(p / 4, kGThreads), b), dim3(kGThreads), 0, s, n,
                           npoints, nsample, ostride, xyz, new_xyz, idx, out);
    else
        hipLaunchKernelGGL(group_xyz_centred_kernel, dim3((unsigned)div_up64(p, kGThreads), b), dim3(kGThreads), 0, s, n, npoints,
                           nsample, ostride, xyz, new_xyz, idx, out);
    return check_launch("group_concat xyz");
}

extern "C" int epnet_group_concat(int b, int c, int n, int npoints, int nsample, const float *xyz, const float *new_xyz,
                                  const float *features, const int *idx, float *out, int use_xyz, epnet_stream_t stream) {
    EPNET_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0);
    EPNET_REQUIRE(use_xyz || c > 0);
    const long long p = (long long)npoints * nsample;
    if (b == 0 || p == 0) return EPNET_OK;
    EPNET_REQUIRE(idx && out && (c == 0 || features) && (!use_xyz || (xyz && new_xyz)));
    EPNET_REQUIRE(n > 0);  // positions to fill from an empty cloud: no index can be valid
    if (p > 0x7fffffffll || b > 65535) return EPNET_ELIMIT;
    hipStream_t s = (hipStream_t)stream;
    const int ch0 = use_xyz ? 3 : 0;
    const size_t ostride = (size_t)(ch0 + c) * (size_t)p;
    if (use_xyz) {
        int rc = launch_group_xyz(b, n, npoints, nsample, p, ostride, xyz, new_xyz, idx, out, s);
        if (rc) return rc;
    }
    if (c == 0) return EPNET_OK;
    return launch_gather_rows(b, c, n, p, features, idx, out + (size_t)ch0 * p, s, "group_concat features", ostride);
}

extern "C" size_t epnet_group_concat_workspace_bytes(int b, int c, int n, int npoints, int nsample) {
    return gather_pm_ws_bytes(b, c, n, (long long)npoints * nsample);
}

// epnet_group_concat with caller scratch of epnet_group_concat_workspace_bytes bytes (0: no scratch helps this shape)
extern "C" int epnet_group_concat_ws(int b, int c, int n, int npoints, int nsample, const float *xyz, const float *new_xyz,
                                     const float *features, const int *idx, float *out, int use_xyz, void *workspace,
                                     size_t workspace_bytes, epnet_stream_t stream) {
    const long long p = (long long)npoints * nsample;
    const size_t need = gather_pm_ws_bytes(b, c, n, p);
    if (need == 0 || !workspace || ((uintptr_t)workspace & 15))
        return epnet_group_concat(b, c, n, npoints, nsample, xyz, new_xyz, features, idx, out, use_xyz, stream);
    if (workspace_bytes < need) return EPNET_ENOMEM;
    EPNET_REQUIRE(b >= 0 && n > 0 && npoints >= 0 && nsample >= 0 && features && idx && out && (!use_xyz || (xyz && new_xyz)));
    if (p > 0x7fffffffll || b > 65535) return EPNET_ELIMIT;
    // the coordinate rows as usual, then the feature rows through the point-major copy
    const int ch0 = use_xyz ? 3 : 0;
    const size_t ostride = (size_t)(ch0 + c) * (size_t)p;
    if (use_xyz) {
        int rc = launch_group_xyz(b, n, npoints, nsample, p, ostride, xyz, new_xyz, idx, out, (hipStream_t)stream);
        if (rc) return rc;
    }
    return launch_gather_rows_pm(b, c, n, p, features, idx, out + (size_t)ch0 * p, ostride, workspace, (hipStream_t)stream,
                                 "group_concat features");
}

// the groupings of the nscales scales of an MSG level (same points, same features) in one call: out[k] (b, 3+c | c,
// npoints, nsamples[k]). With two scales the feature rows are staged in LDS once for both. Same results as nscales calls
// of epnet_group_concat. nsamples / idx / out are HOST arrays of nscales entries.
extern "C" int epnet_group_concat_multi(int b, int c, int n, int npoints, int nscales, const int *nsamples, const float *xyz,
                                        const float *new_xyz, const float *features, const int *const *idx, float *const *out,
                                        int use_xyz, epnet_stream_t stream) {
    EPNET_REQUIRE(nscales >= 0 && (nscales == 0 || (nsamples && idx && out)));
    hipStream_t s = (hipStream_t)stream;
    const int ch0 = use_xyz ? 3 : 0;
    constexpr int kLdsBudget = EPNET_GATHER_LDS_BUDGET_KB * 1024;
    bool fused = nscales == 2 && c >= 8 && b > 0 && n > 0 && npoints > 0 && (size_t)n * 4 <= (size_t)kLdsBudget && features && b <= 65535;
    int rows = 0;
    bool quad = false;
    if (fused) {
        rows = kLdsBudget / (n * 4);
        if (rows > c) rows = c;
        if (rows > 32) rows = 32;
        quad = (c & 3) == 0 && (n & 3) == 0 && rows >= 4 && ((uintptr_t)features & 15) == 0 && !getenv("EPNET_GATHER_NO_QUAD");
        if (quad) rows &= ~3;
        for (int k = 0; k < 2 && fused; ++k) {
            const long long p = (long long)npoints * nsamples[k];
            fused = p >= 2048 && p <= 0x7fffffffll && p % 4 == 0 && idx[k] && out[k] &&
                    (((uintptr_t)idx[k] | (uintptr_t)out[k]) % 16 == 0) && ((size_t)(ch0 + c) * (size_t)p) % 4 == 0;
        }
        fused = fused && (long long)b * div_up(c, rows) >= 512;  // one workgroup serves ALL positions: needs a full chip
    }
    if (!fused) {
        for (int k = 0; k < nscales; ++k) {
            const int rc = epnet_group_concat(b, c, n, npoints, nsamples[k], xyz, new_xyz, features, idx[k], out[k], use_xyz, stream);
            if (rc) return rc;
        }
        return EPNET_OK;
    }
    size_t ostride[2];
    int pk[2];
    for (int k = 0; k < 2; ++k) {
        pk[k] = npoints * nsamples[k];
        ostride[k] = (size_t)(ch0 + c) * (size_t)pk[k];
        if (!use_xyz) continue;
        EPNET_REQUIRE(xyz && new_xyz);
        const long long p = pk[k];
        if (nsamples[k] % 4 == 0)
            hipLaunchKernelGGL(group_xyz_centred_vec4_kernel, dim3((unsigned)div_up64(p / 4, kGThreads), b), dim3(kGThreads), 0, s,
                               n, npoints, nsamples[k], ostride[k], xyz, new_xyz, idx[k], out[k]);
        else
            hipLaunchKernelGGL(group_xyz_centred_kernel, dim3((unsigned)div_up64(p, kGThreads), b), dim3(kGThreads), 0, s, n,
                               npoints, nsamples[k], ostride[k], xyz, new_xyz, idx[k], out[k]);
        const int rc = check_launch("group_concat_multi xyz");
        if (rc) return rc;
    }
    if (quad)
        hipLaunchKernelGGL(gather_rows_lds2_kernel<true>, dim3(div_up(c, rows), b), dim3(kGLdsThreads), (size_t)rows * n * 4, s, c, n,
                           rows, features, pk[0], ostride[0], idx[0], out[0] + (size_t)ch0 * pk[0], pk[1], ostride[1], idx[1],
                           out[1] + (size_t)ch0 * pk[1]);
    else
        hipLaunchKernelGGL(gather_rows_lds2_kernel<false>, dim3(div_up(c, rows), b), dim3(kGLdsThreads), (size_t)rows * n * 4, s, c, n,
                           rows, features, pk[0], ostride[0], idx[0], out[0] + (size_t)ch0 * pk[0], pk[1], ostride[1], idx[1],
                           out[1] + (size_t)ch0 * pk[1]);
    return check_launch("group_concat_multi");
}

extern "C" int epnet_group_concat_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out, const int *idx,
                                       float *grad_features, int use_xyz, epnet_stream_t stream) {
    EPNET_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0);
    const long long p = (long long)npoints * nsample;
    if (b == 0 || c == 0 || p == 0) return EPNET_OK;
    const int ch0 = use_xyz ? 3 : 0;
    const size_t gstride = (size_t)(ch0 + c) * (size_t)p;
    return launch_scatter_rows(b, c, n, p, grad_out + (size_t)ch0 * p, idx, grad_features, (hipStream_t)stream,
                               "group_concat_grad", gstride);
}

extern "C" size_t epnet_group_points_grad_workspace_bytes(int b, int n, int npoints, int nsample) {
    return scatter_ws_bytes(b, n, (long long)npoints * nsample);
}

extern "C" int epnet_group_points_grad_ws(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                                          const int *idx, float *grad_points, void *workspace, size_t workspace_bytes,
                                          epnet_stream_t stream) {
    EPNET_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0);
    const long long p = (long long)npoints * nsample;
    if (scatter_ws_bytes(b, n, p) == 0 || ((uintptr_t)grad_out & 15))
        return launch_scatter_rows(b, c, n, p, grad_out, idx, grad_points, (hipStream_t)stream, "group_points_grad");
    return launch_scatter_rows_csr(b, c, n, p, grad_out, idx, grad_points, workspace, workspace_bytes, (hipStream_t)stream,
                                   "group_points_grad", 0);
}

extern "C" int epnet_group_concat_grad_ws(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                                          const int *idx, float *grad_features, int use_xyz, void *workspace,
                                          size_t workspace_bytes, epnet_stream_t stream) {
    EPNET_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0);
    const long long p = (long long)npoints * nsample;
    if (b == 0 || c == 0 || p == 0) return EPNET_OK;
    const int ch0 = use_xyz ? 3 : 0;
    const size_t gstride = (size_t)(ch0 + c) * (size_t)p;
    if (scatter_ws_bytes(b, n, p) == 0 || ((uintptr_t)grad_out & 15))
        return launch_scatter_rows(b, c, n, p, grad_out + (size_t)ch0 * p, idx, grad_features, (hipStream_t)stream,
                                   "group_concat_grad", gstride);
    return launch_scatter_rows_csr(b, c, n, p, grad_out + (size_t)ch0 * p, idx, grad_features, workspace, workspace_bytes,
                                   (hipStream_t)stream, "group_concat_grad", gstride);
}

extern "C" int epnet_group_linear(int b, int c, int n, int npoints, int nsample, const float *xyz, const float *new_xyz,
                                  const float *z, const int *idx, const float *w_xyz, const float *bias, float *out,
                                  epnet_stream_t stream) {
    EPNET_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0);
    const long long p = (long long)npoints * nsample;
    if (b == 0 || c == 0 || p == 0) return EPNET_OK;
    EPNET_REQUIRE(xyz && new_xyz && z && idx && w_xyz && out && n > 0);
    if (p > 0x7fffffffll || b > 65535 || div_up(c, kGChan) > 65535) return EPNET_ELIMIT;
    hipStream_t s = (hipStream_t)stream;
    constexpr int kLdsBudget = EPNET_GATHER_LDS_BUDGET_KB * 1024;
    const bool vec = nsample % 4 == 0 && (((uintptr_t)idx | (uintptr_t)out) % 16 == 0);
    if (vec && c >= 8 && (size_t)n * 4 <= (size_t)kLdsBudget && p >= 1024) {
        int rows = kLdsBudget / (n * 4);
        if (rows > c) rows = c;
        if (rows > 32) rows = 32;
        const int chunks = div_up(c, rows);
        int tiles = div_up(1024, b * chunks);  // enough workgroups to fill the chip, tiles of at least 1024 positions
        const int max_tiles = (int)(p / 1024);
        if (tiles > max_tiles) tiles = max_tiles;
        if (tiles < 1) tiles = 1;
        int tile = (int)div_up64(p, tiles);
        tile = (tile + kGLdsThreads * 4 - 1) / (kGLdsThreads * 4) * (kGLdsThreads * 4);
        tiles = (int)div_up64(p, tile);
        if (chunks <= 65535) {
            hipLaunchKernelGGL(group_linear_lds_kernel, dim3(tiles, chunks, b), dim3(kGLdsThreads), (size_t)rows * n * 4, s, c, n, npoints,
                               nsample, rows, tile, z, xyz, new_xyz, idx, w_xyz, bias, out);
            return check_launch("group_linear");
        }
    }
    hipLaunchKernelGGL(group_linear_scalar_kernel, dim3((unsigned)div_up64(p, kGThreads), div_up(c, kGChan), b), dim3(kGThreads), 0, s, c,
                       n, npoints, nsample, z, xyz, new_xyz, idx, w_xyz, bias, out);
    return check_launch("group_linear");
}

extern "C" int epnet_group_linear_grad_w(int b, int c, int n, int npoints, int nsample, const float *grad_out, const float *xyz,
                                         const float *new_xyz, const int *idx, float *grad_w, epnet_stream_t stream) {
    EPNET_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0);
    const long long p = (long long)npoints * nsample;
    if (b == 0 || c == 0 || p == 0) return EPNET_OK;
    EPNET_REQUIRE(grad_out && xyz && new_xyz && idx && grad_w && n > 0);
    const int chunks = div_up(c, kGwRows);
    if (p > 0x7fffffffll || b > 65535 || chunks > 65535) return EPNET_ELIMIT;
    int tiles = div_up(2048, b * chunks);
    const int max_tiles = (int)div_up64(p, 1024);
    if (tiles > max_tiles) tiles = max_tiles;
    if (tiles < 1) tiles = 1;
    const int tile = (int)div_up64(p, tiles);
    tiles = (int)div_up64(p, tile);
    hipLaunchKernelGGL(group_linear_grad_w_kernel, dim3(tiles, chunks, b), dim3(kGThreads), 0, (hipStream_t)stream, c, n, npoints,
                       nsample, tile, grad_out, xyz, new_xyz, idx, grad_w);
    return check_launch("group_linear_grad_w");
}

