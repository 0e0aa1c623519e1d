// spatial.h -- Morton keys and an in-LDS bitonic sort, shared by the pruned FPS and ball-query kernels.
// The spatial order only decides WHICH work can be skipped; results never depend on it.
#pragma once
#include "common.h"

namespace epnet {

__device__ __forceinline__ unsigned spread10(unsigned v) {  // ..9876543210 -> ..9__8__7__6__5__4__3__2__1__0
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// 30-bit Morton code of p inside the cube [lo, lo + 1023/scale]^3 (one cell size for all axes)
__device__ __forceinline__ unsigned morton30(float px, float py, float pz, const float (&lo)[3], float scale) {
    const unsigned ix = (unsigned)fminf((px - lo[0]) * scale, 1023.f);
    const unsigned iy = (unsigned)fminf((py - lo[1]) * scale, 1023.f);
    const unsigned iz = (unsigned)fminf((pz - lo[2]) * scale, 1023.f);
    return spread10(ix) | (spread10(iy) << 1) | (spread10(iz) << 2);
}

// ascending bitonic sort of np (a power of two) 64-bit keys in LDS by all threads of the block;
// ends with a barrier
__device__ __forceinline__ void bitonic_sort_lds(unsigned long long *keys, int np) {
    const int q = threadIdx.x, T = blockDim.x;
    for (int k2 = 2; k2 <= np; k2 <<= 1)
        for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
            for (int i = q; i < np / 2; i += T) {
                const int a = ((i & ~(j2 - 1)) << 1) | (i & (j2 - 1));
                const int b = a | j2;
                const unsigned long long ka = keys[a], kb = keys[b];
                const bool up = (a & k2) == 0;
                if ((ka > kb) == up) {
                    keys[a] = kb;
                    keys[b] = ka;
                }
            }
            __syncthreads();
        }
}

// inclusive prefix sum over the 64 lanes (DPP row shifts + row broadcasts)
__device__ __forceinline__ int wave_inclusive_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true);  // row_bcast15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true);  // row_bcast31 -> rows 2, 3
    return v;
}

// bounding box of n points (xyz interleaved) by all threads of the block; s_box is 6 x 16 floats of LDS.
// On return lo[] / ext[] hold the minimum corner and the extents.
__device__ __forceinline__ void block_bbox3(const float *__restrict__ xyz, int n, float (*s_box)[16], float (&lo)[3],
                                            float (&ext)[3]) {
    const int q = threadIdx.x, T = blockDim.x, lane = q & 63, wave = q >> 6, nw = T >> 6;
    float hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    lo[0] = lo[1] = lo[2] = 3.4e38f;
    for (int k = q; k < n; k += T)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = xyz[k * 3 + a];
            lo[a] = fminf(lo[a], v);
            hi[a] = fmaxf(hi[a], v);
        }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], off, 64));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off, 64));
        }
        if (lane == 0) {
            s_box[a][wave] = lo[a];
            s_box[3 + a][wave] = hi[a];
        }
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = s_box[a][0], h = s_box[3 + a][0];
        for (int w = 1; w < nw; ++w) {
            l = fminf(l, s_box[a][w]);
            h = fmaxf(h, s_box[3 + a][w]);
        }
        lo[a] = l;
        ext[a] = h - l;
    }
}

// ---- cell sort: counting sort of the points by a <= 14-bit interleaved cell code ---------------------
// A uniform grid (one cell size for all axes, power-of-two cell counts per axis, at most 2^14 cells in
// all) is laid over the scene; a point's code interleaves the bits of its cell coordinates, so that
// consecutive codes are neighbouring cells. One histogram pass, one scan, one scatter -- ~20x cheaper
// than a full Morton sort, and buckets of 64 consecutive points are just as compact at the scale of a
// cell. The order INSIDE a cell is whatever the LDS atomics produce: the layout only decides which work
// can be skipped, never a result.
constexpr int kCellBits = 14;
constexpr int kCells = 1 << kCellBits;

struct CellGrid {
    float lo[3];
    float inv;    // 1 / cell size
    int bits[3];  // log2(cells) per axis
};

__device__ __forceinline__ CellGrid make_cell_grid(const float (&lo)[3], const float (&ext)[3], int max_bits = kCellBits) {
    CellGrid g;
    const float big = fmaxf(ext[0], fmaxf(ext[1], ext[2]));
    float cell = big > 0.f ? big / 64.f : 1.f;
    for (int iter = 0; iter < 8; ++iter) {
        int total = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int cells = (int)(ext[a] / cell) + 1;
            int b = 0;
            while ((1 << b) < cells && b < 6) ++b;
            g.bits[a] = b;
            total += b;
        }
        if (total <= max_bits) break;
        cell *= 1.26f;  // ~ one bit less in total every 3 iterations
    }
    g.lo[0] = lo[0]; g.lo[1] = lo[1]; g.lo[2] = lo[2];
    g.inv = 1.f / cell;
    return g;
}

__device__ __forceinline__ unsigned cell_code(const CellGrid &g, float px, float py, float pz) {
    const float p[3] = {px, py, pz};
    unsigned c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int hi = (1 << g.bits[a]) - 1;
        const int v = (int)((p[a] - g.lo[a]) * g.inv);
        c[a] = (unsigned)min(max(v, 0), hi);
    }
    unsigned code = 0;
    int pos = 0;
#pragma unroll
    for (int b = 0; b < 6; ++b)
#pragma unroll
        for (int a = 0; a < 3; ++a)
            if (b < g.bits[a]) {
                code |= ((c[a] >> b) & 1u) << pos;
                ++pos;
            }
    return code;
}

// padded histogram index: thread-contiguous runs of `per` bins hit distinct LDS banks
__device__ __forceinline__ int hist_at(int b, int per_shift) { return b + (b >> per_shift); }
__device__ __forceinline__ int cell_hist_words(int threads) { return kCells + kCells / (kCells / threads) + 64; }

// perm[p] = index of the p-th point in cell order, p < n (perm has >= n entries; hist has
// cell_hist_words(blockDim.x) ints; s_part has 16 ints). blockDim.x must be a power of two <= 1024.
// All threads of the block call this; ends with a barrier.
__device__ __forceinline__ void cell_sort_lds(const float *__restrict__ xyz, int n, const CellGrid &g, int *hist,
                                              int *s_part, unsigned short *perm) {
    const int q = threadIdx.x, T = blockDim.x, lane = q & 63, wave = q >> 6, nw = T >> 6;
    const int per = kCells / T;  // bins per thread in the scan
    int per_shift = 0;
    while ((1 << per_shift) < per) ++per_shift;
    for (int i = q; i < cell_hist_words(T); i += T) hist[i] = 0;
    __syncthreads();
    for (int k = q; k < n; k += T)
        atomicAdd(&hist[hist_at((int)cell_code(g, xyz[k * 3 + 0], xyz[k * 3 + 1], xyz[k * 3 + 2]), per_shift)], 1);
    __syncthreads();
    // exclusive scan: thread q owns bins [q*per, (q+1)*per)
    int sum = 0;
    for (int i = 0; i < per; ++i) sum += hist[hist_at(q * per + i, per_shift)];
    const int incl = wave_inclusive_scan(sum);
    if (lane == 63) s_part[wave] = incl;
    __syncthreads();
    int base = incl - sum;
    for (int w = 0; w < wave; ++w) base += s_part[w];
    (void)nw;
    for (int i = 0; i < per; ++i) {
        const int at = hist_at(q * per + i, per_shift);
        const int c = hist[at];
        hist[at] = base;
        base += c;
    }
    __syncthreads();
    for (int k = q; k < n; k += T) {
        const int pos = atomicAdd(&hist[hist_at((int)cell_code(g, xyz[k * 3 + 0], xyz[k * 3 + 1], xyz[k * 3 + 2]), per_shift)], 1);
        perm[pos] = (unsigned short)k;
    }
    __syncthreads();
}

// ---- scene index: caller scratch shared by FPS, ball query and three_nn of one level ---------------------
// per scene: the points counting-sorted by cell as float4 (x, y, z, original index; padding = 3e38 / -1),
// padded to np = a power of two >= 2048, followed (after all scenes) by one box (6 floats) per 64 sorted points
// ("bucket") and then one box per 256 sorted points ("quad")
inline int scene_index_np(int n) {
    int np = 2048;
    while (np < n) np <<= 1;
    return np;
}
// scenes beyond 16384 points: np floats more per scene at the end -- scratch of the sampling kernel (its running distances
// in sorted order); nothing else reads or writes that tail
inline size_t scene_index_bytes(int b, int n) {
    if (b <= 0 || n < 1024 || n > 65536) return 0;
    const size_t np = (size_t)scene_index_np(n);
    return (size_t)b * (np * sizeof(float4) + (np / 64 + np / 256) * 6 * sizeof(float) + (n > 16384 ? np * sizeof(float) : 0));
}
inline float *scene_index_sampling_scratch(int b, int n, void *index) {
    const size_t np = (size_t)scene_index_np(n);
    return (float *)((char *)index + (size_t)b * (np * sizeof(float4) + (np / 64 + np / 256) * 6 * sizeof(float)));
}

// defined in ball_query.hip
// qboxes (one box per 4 buckets) may be NULL; src_idx / gathered (n <= 16384): see bq_index_kernel
int spatial_index_launch(int b, int n, int np, const float *xyz, float4 *sorted, float *boxes, float *qboxes, hipStream_t s,
                         int n_src = 0, const int *src_idx = nullptr, float *gathered = nullptr);

}  // namespace epnet
