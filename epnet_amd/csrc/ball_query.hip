// ball_query.hip -- radius neighbour search for gfx950.
//
// Replaces ball_query_kernel_fast / _launcher (pointnet2_lib/pointnet2/src/ball_query_gpu.cu:9-66).
// The reference walks all N points with ONE THREAD per centre (stride-3 scalar loads, divergent
// early exit). Here one 64-lane wave owns CPW centres: the block stages a tile of the interleaved
// (N,3) xyz array in LDS with coalesced loads, each lane tests one point of a 64-point chunk
// against the wave's centres (held in scalar registers), and __ballot + mbcnt give every hit its
// slot in index order -- so "the first nsample points with d2 < r2, in index order, padded with
// the first hit" (:29-43) is preserved exactly while the scan is 64 points wide. A wave stops
// scanning as soon as all its centres are full; the block leaves when all its waves have.
//
// d2 = (cx-x)*(cx-x) + (cy-y)*(cy-y) + (cz-z)*(cz-z) in source order, no contraction; strict '<'.
#include "common.h"

namespace epnet {

constexpr int kBqTile = 2048;     // points per LDS tile (24 KiB)
constexpr int kBqThreads = 256;   // 4 waves

template <int CPW>
__global__ __launch_bounds__(kBqThreads) void ball_query_kernel(int n, int m, float radius2, int nsample,
                                                                const float *__restrict__ new_xyz,
                                                                const float *__restrict__ xyz,
                                                                int *__restrict__ idx) {
    __shared__ float tile[kBqTile * 3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bs = blockIdx.y;
    xyz += (size_t)bs * n * 3;
    new_xyz += (size_t)bs * m * 3;
    idx += (size_t)bs * m * nsample;

    const int c0 = (blockIdx.x * (kBqThreads / 64) + wave) * CPW;
    float cx[CPW], cy[CPW], cz[CPW];
    int cnt[CPW], first[CPW];
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
        const int ci = c0 + i;
        const bool ok = ci < m;
        cx[i] = ok ? new_xyz[ci * 3 + 0] : 0.f;
        cy[i] = ok ? new_xyz[ci * 3 + 1] : 0.f;
        cz[i] = ok ? new_xyz[ci * 3 + 2] : 0.f;
        cnt[i] = ok ? 0 : nsample;  // out-of-range centres count as already full
        first[i] = 0;
    }

    for (int t0 = 0; t0 < n; t0 += kBqTile) {
        const int tn = min(kBqTile, n - t0);
        for (int e = threadIdx.x; e < tn * 3; e += kBqThreads) tile[e] = xyz[(size_t)t0 * 3 + e];
        __syncthreads();

        bool wave_done = true;
#pragma unroll
        for (int i = 0; i < CPW; ++i) wave_done = wave_done && (cnt[i] >= nsample);

        if (!wave_done) {
            for (int p0 = 0; p0 < tn; p0 += 64) {
                const int p = p0 + lane;
                const bool valid = p < tn;
                const float x = valid ? tile[p * 3 + 0] : 0.f;
                const float y = valid ? tile[p * 3 + 1] : 0.f;
                const float z = valid ? tile[p * 3 + 2] : 0.f;
                bool all_full = true;
#pragma unroll
                for (int i = 0; i < CPW; ++i) {
                    if (cnt[i] < nsample) {  // wave-uniform
                        const float dx = cx[i] - x, dy = cy[i] - y, dz = cz[i] - z;
                        const float d2 = dx * dx + dy * dy + dz * dz;
                        const bool hit = valid && (d2 < radius2);
                        const unsigned long long mask = __ballot(hit);
                        if (mask) {
                            const int pos = cnt[i] + popc_below(mask);
                            if (hit && pos < nsample) idx[(size_t)(c0 + i) * nsample + pos] = t0 + p;
                            if (cnt[i] == 0) first[i] = t0 + p0 + (int)__builtin_ctzll(mask);
                            cnt[i] += (int)__popcll(mask);
                        }
                        all_full = all_full && (cnt[i] >= nsample);
                    }
                }
                if (all_full) {
                    wave_done = true;
                    break;
                }
            }
        }
        // barrier before the tile is overwritten; doubles as the block-wide "everyone full" vote
        if (__syncthreads_and(wave_done ? 1 : 0)) break;
    }

    // padding: slots [cnt, nsample) repeat the first hit (:35-39); an empty ball is all zeros
    // (the value the caller's zero fill leaves in the reference, pointnet2_utils.py:218)
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
        const int ci = c0 + i;
        if (ci < m && cnt[i] < nsample) {
            const int fill = cnt[i] > 0 ? first[i] : 0;
            for (int l = cnt[i] + lane; l < nsample; l += 64) idx[(size_t)ci * nsample + l] = fill;
        }
    }
}

}  // namespace epnet

using namespace epnet;

extern "C" int epnet_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                                const float *xyz, int *idx, epnet_stream_t stream) {
    EPNET_REQUIRE(b >= 0 && n >= 0 && m >= 0 && nsample >= 0);
    if (b == 0 || m == 0 || nsample == 0) return EPNET_OK;
    EPNET_REQUIRE(new_xyz && xyz && idx);
    EPNET_REQUIRE(b <= 65535);
    const float radius2 = radius * radius;  // ball_query_gpu.cu:23
    hipStream_t s = (hipStream_t)stream;
    constexpr int waves = kBqThreads / 64;
    // few centres: one per wave so that the grid still covers the chip; many: two per wave to
    // halve the LDS reads per pair test
    if ((long long)m * b >= 8192) {
        dim3 grid(div_up(m, waves * 2), b);
        hipLaunchKernelGGL(ball_query_kernel<2>, grid, dim3(kBqThreads), 0, s, n, m, radius2, nsample, new_xyz, xyz, idx);
    } else {
        dim3 grid(div_up(m, waves), b);
        hipLaunchKernelGGL(ball_query_kernel<1>, grid, dim3(kBqThreads), 0, s, n, m, radius2, nsample, new_xyz, xyz, idx);
    }
    return check_launch("ball_query");
}
