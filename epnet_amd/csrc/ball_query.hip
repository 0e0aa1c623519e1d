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
#include "spatial.h"

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


// ---- indexed path ----------------------------------------------------------------------------------
//
// For the big levels (N from 2048 up to 65536) the scan is N*M pair tests although a ball only ever contains a
// handful of points. With caller-supplied scratch the query is split in two launches:
//   bq_index_kernel  one workgroup per scene sorts the points by grid cell (counting sort on an interleaved cell code)
//                    and writes them, with their ORIGINAL index, as float4 plus one bounding box per
//                    bucket of 64 consecutive sorted points;
//   bq_query_kernel  one wave per centre: the lanes test the buckets' boxes against the ball (the lower
//                    bound |clamp(c, box) - c|^2 uses the point-test expression, so by monotonicity of
//                    fp32 arithmetic no bucket holding a hit is ever skipped), the few surviving
//                    buckets are tested point by point, and every hit sets bit `original index` in an
//                    N-bit LDS bitmap. Reading the bitmap back in order yields exactly the reference's
//                    result: the first nsample hits in index order, padded with the first one.
// The point test is the same expression as above: d2 = (cx-x)*(cx-x) + (cy-y)*(cy-y) + (cz-z)*(cz-z).

constexpr int kIxThreads = 1024;
constexpr int kIxMaxPoints = 65536;

__global__ __launch_bounds__(kIxThreads) void bq_index_kernel(int n, int np, const float *__restrict__ xyz,
                                                              float4 *__restrict__ sorted, float *__restrict__ boxes) {
    extern __shared__ int s_hist[];  // cell histogram / running offsets (spatial.h)
    __shared__ float s_box[6][16];
    __shared__ int s_part[16];
    const int q = threadIdx.x, lane = q & 63, wave = q >> 6;
    xyz += (size_t)blockIdx.x * n * 3;
    sorted += (size_t)blockIdx.x * np;
    boxes += (size_t)blockIdx.x * (np / 64) * 6;
    float lo[3], ext[3];
    block_bbox3(xyz, n, s_box, lo, ext);
    const CellGrid g = make_cell_grid(lo, ext);
    // counting sort by cell, scattering the points (with their original index) straight to global memory
    constexpr int per = kCells / kIxThreads;
    constexpr int per_shift = per == 16 ? 4 : per == 8 ? 3 : per == 4 ? 2 : -1;
    static_assert(per_shift > 0, "scan layout");
    for (int i = q; i < cell_hist_words(kIxThreads); i += kIxThreads) s_hist[i] = 0;
    __syncthreads();
    for (int k = q; k < n; k += kIxThreads)
        atomicAdd(&s_hist[hist_at((int)cell_code(g, xyz[k * 3 + 0], xyz[k * 3 + 1], xyz[k * 3 + 2]), per_shift)], 1);
    __syncthreads();
    int sum = 0;
    for (int i = 0; i < per; ++i) sum += s_hist[hist_at(q * per + i, per_shift)];
    const int incl = wave_inclusive_scan(sum);
    if (lane == 63) s_part[wave] = incl;
    __syncthreads();
    int base = incl - sum;
    for (int w = 0; w < wave; ++w) base += s_part[w];
    for (int i = 0; i < per; ++i) {
        const int at = hist_at(q * per + i, per_shift);
        const int c = s_hist[at];
        s_hist[at] = base;
        base += c;
    }
    __syncthreads();
    for (int k = q; k < n; k += kIxThreads) {
        const float x = xyz[k * 3 + 0], y = xyz[k * 3 + 1], z = xyz[k * 3 + 2];
        const int pos = atomicAdd(&s_hist[hist_at((int)cell_code(g, x, y, z), per_shift)], 1);
        sorted[pos] = make_float4(x, y, z, __int_as_float(k));
    }
    for (int p = n + q; p < np; p += kIxThreads)  // padding: never inside a ball
        sorted[p] = make_float4(3.0e38f, 3.0e38f, 3.0e38f, __int_as_float(-1));
    __threadfence_block();
    __syncthreads();  // the scattered points are read back by other waves of this workgroup below
    for (int p = q; p < np; p += kIxThreads) {  // one wave handles one bucket at a time
        const float4 v = sorted[p];
        const bool real = p < n;
        float mn[3] = {real ? v.x : 3.4e38f, real ? v.y : 3.4e38f, real ? v.z : 3.4e38f};
        float mx[3] = {real ? v.x : -3.4e38f, real ? v.y : -3.4e38f, real ? v.z : -3.4e38f};
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, 64));
                mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, 64));
            }
        if (lane == 0) {
            float *bx = boxes + (p >> 6) * 6;
            const bool any = mn[0] <= mx[0];  // an all-padding bucket gets a box no ball can reach
            bx[0] = any ? mn[0] : 3.0e38f; bx[1] = any ? mx[0] : 3.0e38f;
            bx[2] = any ? mn[1] : 3.0e38f; bx[3] = any ? mx[1] : 3.0e38f;
            bx[4] = any ? mn[2] : 3.0e38f; bx[5] = any ? mx[2] : 3.0e38f;
        }
    }
}

constexpr int kQThreads = 256;

// DPL = bitmap dwords per lane = np / 2048 (np >= 2048)
template <int DPL>
__global__ __launch_bounds__(kQThreads) void bq_query_kernel(int np, int m, float radius2, int nsample,
                                                             const float *__restrict__ new_xyz,
                                                             const float4 *__restrict__ sorted,
                                                             const float *__restrict__ boxes, int *__restrict__ idx) {
    __shared__ unsigned s_bits[kQThreads / 64][64 * DPL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bs = blockIdx.y;
    const int ci = blockIdx.x * (kQThreads / 64) + wave;
    if (ci >= m) return;  // wave-uniform; no block-level barrier below
    sorted += (size_t)bs * np;
    boxes += (size_t)bs * (np / 64) * 6;
    const float *c = new_xyz + ((size_t)bs * m + ci) * 3;
    const float cx = c[0], cy = c[1], cz = c[2];
    int *out = idx + ((size_t)bs * m + ci) * nsample;
    unsigned *bits = s_bits[wave];
#pragma unroll
    for (int w = 0; w < DPL; ++w) bits[lane * DPL + w] = 0u;

    const int nb = np >> 6;  // buckets, a multiple of 32 (np >= 2048)
    for (int b0 = 0; b0 < nb; b0 += 64) {
        const int b = b0 + lane;
        bool near = false;
        if (b < nb) {
            const float *bx = boxes + b * 6;
            const float px = __builtin_amdgcn_fmed3f(cx, bx[0], bx[1]), py = __builtin_amdgcn_fmed3f(cy, bx[2], bx[3]),
                        pz = __builtin_amdgcn_fmed3f(cz, bx[4], bx[5]);
            const float dx = cx - px, dy = cy - py, dz = cz - pz;
            near = (dx * dx + dy * dy + dz * dz) < radius2;
        }
        unsigned long long cand = __ballot(near);
        while (cand) {
            const int bb = b0 + (int)__builtin_ctzll(cand);
            cand &= cand - 1ull;
            const float4 p = sorted[(bb << 6) + lane];
            const float dx = cx - p.x, dy = cy - p.y, dz = cz - p.z;
            const float d2 = dx * dx + dy * dy + dz * dz;
            if (d2 < radius2) {
                const int k = __float_as_int(p.w);
                atomicOr(&bits[k >> 5], 1u << (k & 31));
            }
        }
    }
    // read the bitmap back in index order: lane l owns bits [l*32*DPL, (l+1)*32*DPL)
    int cnt = 0;
#pragma unroll 8
    for (int i = 0; i < DPL; ++i) cnt += __popc(bits[lane * DPL + i]);
    const int incl = wave_inclusive_scan(cnt);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    int pos = incl - cnt;
    int mine_first = 0x7fffffff;
    if (cnt > 0 && (pos < nsample || pos == 0)) {
        for (int i = 0; i < DPL; ++i) {
            unsigned ww = bits[lane * DPL + i];
            if (ww && mine_first == 0x7fffffff) mine_first = (lane * DPL + i) * 32 + (int)__builtin_ctz(ww);
            while (ww && pos < nsample) {
                const int bit = (int)__builtin_ctz(ww);
                ww &= ww - 1u;
                out[pos++] = (lane * DPL + i) * 32 + bit;
            }
            if (pos >= nsample) break;
        }
    }
    // padding with the first hit (ball_query_gpu.cu:35-39); an empty ball is all zeros
    const unsigned long long have = __ballot(cnt > 0);
    int first = 0;
    if (have) first = __builtin_amdgcn_readlane(mine_first, (int)__builtin_ctzll(have));
    for (int l = total + lane; l < nsample; l += 64) out[l] = first;
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

static size_t bq_index_lds(int) { return (size_t)(kCells + kCells / (kCells / kIxThreads) + 64) * sizeof(int); }

// shared with three_nn (interpolate.hip): cell-sorted float4 copy (x, y, z, original index) of n points padded
// to np (a multiple of 64) per scene, plus one box (6 floats) per 64 sorted points
int epnet::spatial_index_launch(int b, int n, int np, const float *xyz, float4 *sorted, float *boxes, hipStream_t s) {
    hipLaunchKernelGGL(bq_index_kernel, dim3(b), dim3(kIxThreads), bq_index_lds(np), s, n, np, xyz, sorted, boxes);
    return check_launch("spatial index");
}

static int bq_query_launch(int b, int np, int m, float radius, int nsample, const float *new_xyz, const float4 *sorted,
                           const float *boxes, int *idx, hipStream_t s) {
    const float radius2 = radius * radius;  // ball_query_gpu.cu:23
    dim3 grid(div_up(m, kQThreads / 64), b);
    switch (np / 2048) {
        case 1: hipLaunchKernelGGL(bq_query_kernel<1>, grid, dim3(kQThreads), 0, s, np, m, radius2, nsample, new_xyz, sorted, boxes, idx); break;
        case 2: hipLaunchKernelGGL(bq_query_kernel<2>, grid, dim3(kQThreads), 0, s, np, m, radius2, nsample, new_xyz, sorted, boxes, idx); break;
        case 4: hipLaunchKernelGGL(bq_query_kernel<4>, grid, dim3(kQThreads), 0, s, np, m, radius2, nsample, new_xyz, sorted, boxes, idx); break;
        case 8: hipLaunchKernelGGL(bq_query_kernel<8>, grid, dim3(kQThreads), 0, s, np, m, radius2, nsample, new_xyz, sorted, boxes, idx); break;
        case 16: hipLaunchKernelGGL(bq_query_kernel<16>, grid, dim3(kQThreads), 0, s, np, m, radius2, nsample, new_xyz, sorted, boxes, idx); break;
        default: hipLaunchKernelGGL(bq_query_kernel<32>, grid, dim3(kQThreads), 0, s, np, m, radius2, nsample, new_xyz, sorted, boxes, idx); break;
    }
    return check_launch("ball_query query");
}

extern "C" size_t epnet_scene_index_bytes(int b, int n) { return scene_index_bytes(b, n); }

extern "C" int epnet_scene_index_build(int b, int n, const float *xyz, void *index, size_t index_bytes,
                                       epnet_stream_t stream) {
    const size_t need = scene_index_bytes(b, n);
    EPNET_REQUIRE(need != 0 && xyz && index);
    if (index_bytes < need) return EPNET_ENOMEM;
    if (((uintptr_t)index & 15) != 0) return EPNET_EINVAL;
    EPNET_REQUIRE(b <= 65535);
    const int np = scene_index_np(n);
    float4 *sorted = (float4 *)index;
    return spatial_index_launch(b, n, np, xyz, sorted, (float *)(sorted + (size_t)b * np), (hipStream_t)stream);
}

extern "C" int epnet_ball_query_indexed(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                                        const float *xyz, const void *index, size_t index_bytes, int *idx,
                                        epnet_stream_t stream) {
    const size_t need = scene_index_bytes(b, n);
    if (need == 0 || !index || nsample <= 0 || m <= 0)
        return epnet_ball_query(b, n, m, radius, nsample, new_xyz, xyz, idx, stream);
    EPNET_REQUIRE(new_xyz && idx);
    if (index_bytes < need) return EPNET_ENOMEM;
    EPNET_REQUIRE(b <= 65535);
    const int np = scene_index_np(n);
    const float4 *sorted = (const float4 *)index;
    return bq_query_launch(b, np, m, radius, nsample, new_xyz, sorted, (const float *)(sorted + (size_t)b * np), idx,
                           (hipStream_t)stream);
}

extern "C" size_t epnet_ball_query_workspace_bytes(int b, int n, int m) {
    if (n < 2048 || m <= 0) return 0;  // small or huge scenes: direct scan, no scratch
    return scene_index_bytes(b, n);
}

extern "C" int epnet_ball_query_ws(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                                   const float *xyz, int *idx, void *workspace, size_t workspace_bytes,
                                   epnet_stream_t stream) {
    const size_t need = epnet_ball_query_workspace_bytes(b, n, m);
    if (need == 0 || nsample <= 0) return epnet_ball_query(b, n, m, radius, nsample, new_xyz, xyz, idx, stream);
    EPNET_REQUIRE(new_xyz && xyz && idx && workspace);
    int rc = epnet_scene_index_build(b, n, xyz, workspace, workspace_bytes, stream);
    if (rc) return rc;
    return epnet_ball_query_indexed(b, n, m, radius, nsample, new_xyz, xyz, workspace, workspace_bytes, idx, stream);
}
