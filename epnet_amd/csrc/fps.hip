// fps.hip -- furthest point sampling for gfx950.
//
// Replaces furthest_point_sampling_kernel / _launcher of the reference
// (pointnet2_lib/pointnet2/src/sampling_gpu.cu:94-253). One workgroup per scene, as there, but:
//
//   * the scene lives in VGPRs for the whole kernel (x, y, z and the running min distance of up
//     to 16 points per thread) -- the reference re-reads xyz and temp from memory on each of the
//     M-1 dependent iterations; here HBM sees N*12 B once and M*4 B of indices;
//   * the arg-max is a wave-level butterfly on a 64-bit key followed by ONE barrier per
//     iteration (double-buffered 16-entry LDS exchange) instead of an 11-barrier shared-memory
//     tree;
//   * tie-breaking is made independent of the reduction order. The reference's result depends on
//     its block size bs = opt_n_threads(N) (cuda_utils.h:10-14): thread tid scans k = tid,
//     tid+bs, ... keeping the FIRST maximum (strict '>', :136-137), and at every level of the
//     tree the LOWER slot wins a tie (:86-91), i.e. among equal distances the winner is the one
//     with the smallest (bitreverse(k mod bs), k div bs). Squared distances are >= +0, so their
//     IEEE bit patterns order like the floats; the key
//         (bits(d2) << 32) | (0x7fffffff - ((bitrev(k mod bs) << 20) | (k div bs)))
//     therefore has a unique maximum, which is exactly the reference's winner.
//
// Arithmetic: d = dx*dx + dy*dy + dz*dz evaluated left to right in fp32 without contraction
// (the file is compiled with -ffp-contract=off), min with the running distance, as :133-135.
#include <math.h>

#include "common.h"

namespace epnet {

__device__ __forceinline__ unsigned bitrev_lg(unsigned v, int lg) {
    return lg == 0 ? 0u : (__brev(v) >> (32 - lg));
}

__device__ __forceinline__ unsigned fps_rank(int k, int lg) {
    return (bitrev_lg((unsigned)k & ((1u << lg) - 1u), lg) << 20) | ((unsigned)k >> lg);
}

__device__ __forceinline__ int fps_unrank(unsigned rank, int lg) {
    return (int)(bitrev_lg(rank >> 20, lg) + ((rank & 0xFFFFFu) << lg));
}

__device__ __forceinline__ long long wave_max_i64(long long v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const long long o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// Register-resident kernel: blockDim.x * PPT >= n, blockDim.x a multiple of the reference block
// size 2^lg_bs (so every point held by one thread shares k mod bs and slot order == rank order).
template <int PPT>
__global__ __launch_bounds__(1024) void fps_reg_kernel(int n, int m, int lg_bs, const float *__restrict__ xyz,
                                                       float *__restrict__ temp, int *__restrict__ idxs) {
    __shared__ long long red[2][16];
    const int BS = blockDim.x;
    const int q = threadIdx.x;
    const int lane = q & 63, wave = q >> 6;
    xyz += (size_t)blockIdx.x * n * 3;
    if (temp) temp += (size_t)blockIdx.x * n;
    idxs += (size_t)blockIdx.x * m;

    float x[PPT], y[PPT], z[PPT], t[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int k = q + j * BS;
        if (k < n) {
            x[j] = xyz[k * 3 + 0];
            y[j] = xyz[k * 3 + 1];
            z[j] = xyz[k * 3 + 2];
            t[j] = temp ? temp[k] : 1e10f;
        } else {  // padding slot: distance pinned at -1, can never be a maximum
            x[j] = y[j] = z[j] = 0.f;
            t[j] = -1.f;
        }
    }
    if (q < 32) red[q >> 4][q & 15] = (long long)0x8000000000000000ull;
    if (q == 0) idxs[0] = 0;
    float x1 = xyz[0], y1 = xyz[1], z1 = xyz[2];
    __syncthreads();

    for (int it = 1; it < m; ++it) {
        float best = -1.f;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const float dx = x[j] - x1, dy = y[j] - y1, dz = z[j] - z1;
            const float d = dx * dx + dy * dy + dz * dz;
            t[j] = fminf(d, t[j]);
            best = fmaxf(best, t[j]);
        }
        int bj = PPT - 1;  // first slot holding the thread maximum
#pragma unroll
        for (int j = PPT - 2; j >= 0; --j) bj = (t[j] == best) ? j : bj;
        const unsigned rank = fps_rank(q + bj * BS, lg_bs);
        long long key = ((long long)__float_as_int(best) << 32) | (long long)(0x7FFFFFFFu - rank);
        key = wave_max_i64(key);
        if (lane == 0) red[it & 1][wave] = key;
        __syncthreads();
        long long kk = red[it & 1][lane & 15];
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) {
            const long long o = __shfl_xor(kk, off, 64);
            kk = o > kk ? o : kk;
        }
        const unsigned wr = 0x7FFFFFFFu - (unsigned)(kk & 0xFFFFFFFFll);
        const int old = __builtin_amdgcn_readfirstlane(fps_unrank(wr, lg_bs));
        x1 = xyz[old * 3 + 0];
        y1 = xyz[old * 3 + 1];
        z1 = xyz[old * 3 + 2];
        if (q == 0) idxs[it] = old;
    }

    if (temp) {
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int k = q + j * BS;
            if (k < n) temp[k] = t[j];
        }
    }
}

// Streaming kernel for scenes too large for the register file (n > 16 * 1024): xyz and temp are
// re-read through L2 each iteration, blockDim.x == the reference block size.
__global__ __launch_bounds__(1024) void fps_stream_kernel(int n, int m, int lg_bs, const float *__restrict__ xyz,
                                                          float *__restrict__ temp, int *__restrict__ idxs) {
    __shared__ long long red[2][16];
    const int BS = blockDim.x;
    const int q = threadIdx.x;
    const int lane = q & 63, wave = q >> 6;
    xyz += (size_t)blockIdx.x * n * 3;
    temp += (size_t)blockIdx.x * n;
    idxs += (size_t)blockIdx.x * m;
    if (q < 32) red[q >> 4][q & 15] = (long long)0x8000000000000000ull;
    if (q == 0) idxs[0] = 0;
    float x1 = xyz[0], y1 = xyz[1], z1 = xyz[2];
    __syncthreads();
    for (int it = 1; it < m; ++it) {
        float best = -1.f;
        int bestk = q;
        for (int k = q; k < n; k += BS) {
            const float dx = xyz[k * 3 + 0] - x1, dy = xyz[k * 3 + 1] - y1, dz = xyz[k * 3 + 2] - z1;
            const float d = dx * dx + dy * dy + dz * dz;
            const float d2 = fminf(d, temp[k]);
            temp[k] = d2;
            bestk = d2 > best ? k : bestk;
            best = d2 > best ? d2 : best;
        }
        const unsigned rank = fps_rank(bestk, lg_bs);
        long long key = ((long long)__float_as_int(best) << 32) | (long long)(0x7FFFFFFFu - rank);
        key = wave_max_i64(key);
        if (lane == 0) red[it & 1][wave] = key;
        __syncthreads();
        long long kk = red[it & 1][lane & 15];
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) {
            const long long o = __shfl_xor(kk, off, 64);
            kk = o > kk ? o : kk;
        }
        const unsigned wr = 0x7FFFFFFFu - (unsigned)(kk & 0xFFFFFFFFll);
        const int old = __builtin_amdgcn_readfirstlane(fps_unrank(wr, lg_bs));
        x1 = xyz[old * 3 + 0];
        y1 = xyz[old * 3 + 1];
        z1 = xyz[old * 3 + 2];
        if (q == 0) idxs[it] = old;
    }
}

// opt_n_threads, pointnet2_lib/pointnet2/src/cuda_utils.h:10-14 -- same double-precision libm
// formula as the reference's host code, so the block size that defines the tie-break is the same.
static int ref_block_lg(int work_size) {
    int pow_2 = (int)(std::log(static_cast<double>(work_size)) / std::log(2.0));
    if (pow_2 > 10) pow_2 = 10;
    if (pow_2 < 0) pow_2 = 0;
    return pow_2;
}

}  // namespace epnet

using namespace epnet;

extern "C" int epnet_furthest_point_sampling(int b, int n, int m, const float *xyz, float *temp, int *idx,
                                             epnet_stream_t stream) {
    EPNET_REQUIRE(b >= 0 && n >= 1 && m >= 0);
    if (b == 0 || m == 0) return EPNET_OK;  // the reference kernel returns at once for m <= 0
    EPNET_REQUIRE(xyz && idx);
    if ((long long)n > (1ll << 20) * 1024) return EPNET_ELIMIT;  // rank field is 20 bits of k div bs
    hipStream_t s = (hipStream_t)stream;
    const int lg = ref_block_lg(n);
    const int bs_ref = 1 << lg;
    const int bs = bs_ref < 64 ? 64 : bs_ref;  // physical block: at least one wave
    const int ppt = div_up(n, bs);
    dim3 grid(b), block(bs);
    if (ppt <= 1)
        hipLaunchKernelGGL(fps_reg_kernel<1>, grid, block, 0, s, n, m, lg, xyz, temp, idx);
    else if (ppt <= 2)
        hipLaunchKernelGGL(fps_reg_kernel<2>, grid, block, 0, s, n, m, lg, xyz, temp, idx);
    else if (ppt <= 4)
        hipLaunchKernelGGL(fps_reg_kernel<4>, grid, block, 0, s, n, m, lg, xyz, temp, idx);
    else if (ppt <= 8)
        hipLaunchKernelGGL(fps_reg_kernel<8>, grid, block, 0, s, n, m, lg, xyz, temp, idx);
    else if (ppt <= 16)
        hipLaunchKernelGGL(fps_reg_kernel<16>, grid, block, 0, s, n, m, lg, xyz, temp, idx);
    else {
        EPNET_REQUIRE(temp != nullptr);  // the streaming path keeps the distances in the caller's buffer
        hipLaunchKernelGGL(fps_stream_kernel, grid, block, 0, s, n, m, lg, xyz, temp, idx);
    }
    return check_launch("furthest_point_sampling");
}
