// roipool3d.hip -- point-in-box ROI pooling for gfx950.
//
// Replaces roipool3dLauncher and its three kernels assign_pts_to_box3d / get_pooled_idx /
// roipool3d_forward (lib/utils/roipool3d/src/roipool3d_kernel.cu:97-237; pt_in_box3d :14-28) and,
// with the same result, forward_slow (:31-94, :197-206).
//
// The reference materialises a (B,N,M) int assignment tensor (cudaMalloc'ed per call), then one
// THREAD per box walks it with stride-M reads to pick the first S in-box points. Here one
// workgroup owns one (scene, box): its 4 waves scan 4 contiguous quarters of the cloud 64 points
// at a time, __ballot + mbcnt compact the hits in index order into LDS, the four lists are
// concatenated in quarter order (= global index order), truncated to S and padded cyclically
// (k % cnt, :152-158); the same workgroup then copies the S rows (xyz + C features) with
// lane-contiguous loads/stores. No scratch tensor, no device allocation, no extra launches.
//
// Predicate arithmetic: as pt_in_box3d, fp32 in source order without contraction; the double
// sub-expressions of the reference (h / 2.0, -l / 2.0, ...) are exact halvings, so their float
// forms compare identically; cos/sin per the parity definition (correctly rounded, via double).
#include <math.h>

#include "common.h"

namespace epnet {

constexpr int kRpThreads = 256;
constexpr int kRpWaves = kRpThreads / 64;

__global__ __launch_bounds__(kRpThreads) void roipool3d_kernel(int pts_num, int boxes_num, int feature_in_len,
                                                               int sampled_pts_num, const float *__restrict__ xyz,
                                                               const float *__restrict__ boxes3d,
                                                               const float *__restrict__ pts_feature,
                                                               float *__restrict__ pooled_features,
                                                               int *__restrict__ pooled_empty_flag) {
    extern __shared__ int lds[];  // [kRpWaves][S] per-wave hit lists, then [S] final list
    __shared__ int wave_cnt[kRpWaves];
    const int S = sampled_pts_num;
    int *lists = lds;
    int *final_idx = lds + kRpWaves * S;

    const int box = blockIdx.x, bs = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    xyz += (size_t)bs * pts_num * 3;
    pts_feature += (size_t)bs * pts_num * feature_in_len;
    const float *bx = boxes3d + ((size_t)bs * boxes_num + box) * 7;

    const float cx = bx[0], bottom_y = bx[1], cz = bx[2], h = bx[3], w = bx[4], l = bx[5], angle = bx[6];
    const float max_dis = 10.0f;
    const float cy = (float)((double)bottom_y - (double)h / 2.0);
    const float hh = h * 0.5f, hw = w * 0.5f, hl = l * 0.5f;
    const float cosa = (float)cos((double)angle), sina = (float)sin((double)angle);
    const float nsina = -sina;

    // phase 1: each wave compacts the hits of its quarter of the cloud
    const int chunks = (pts_num + 63) / 64;
    const int per_wave = (chunks + kRpWaves - 1) / kRpWaves;
    const int k_begin = wave * per_wave * 64;
    const int k_end = min(pts_num, (wave + 1) * per_wave * 64);
    int cnt = 0;
    int *mylist = lists + wave * S;
    // the scan is a chain of dependent loads, not arithmetic: 8 chunks of 64 points are requested together and then
    // compacted in index order (positions beyond S are dropped, so testing up to 448 points past the S-th hit
    // changes nothing)
    constexpr int kRpChunks = 8;
    for (int k0 = k_begin; k0 < k_end && cnt < S; k0 += 64 * kRpChunks) {
        float px[kRpChunks], py[kRpChunks], pz[kRpChunks];
        bool valid[kRpChunks];
#pragma unroll
        for (int u = 0; u < kRpChunks; ++u) {
            const int k = k0 + u * 64 + lane;
            valid[u] = k < k_end;
            const int kk = valid[u] ? k : k_begin;
            px[u] = xyz[kk * 3 + 0];
            py[u] = xyz[kk * 3 + 1];
            pz[u] = xyz[kk * 3 + 2];
        }
#pragma unroll
        for (int u = 0; u < kRpChunks; ++u) {
            const float x = px[u], y = py[u], z = pz[u];
            bool in = false;
            if (valid[u] && !((fabsf(x - cx) > max_dis) || (fabsf(y - cy) > hh) || (fabsf(z - cz) > max_dis))) {
                const float x_rot = (x - cx) * cosa + (z - cz) * nsina;
                const float z_rot = (x - cx) * sina + (z - cz) * cosa;
                in = (x_rot >= -hl) & (x_rot <= hl) & (z_rot >= -hw) & (z_rot <= hw);
            }
            const unsigned long long mask = __ballot(in);
            if (mask) {
                const int pos = cnt + popc_below(mask);
                if (in && pos < S) mylist[pos] = k0 + u * 64 + lane;
                cnt += (int)__popcll(mask);
            }
        }
    }
    if (cnt > S) cnt = S;
    if (lane == 0) wave_cnt[wave] = cnt;
    __syncthreads();

    // phase 2: concatenate in quarter order, truncate to S, pad cyclically
    int offs[kRpWaves + 1];
    offs[0] = 0;
#pragma unroll
    for (int i = 0; i < kRpWaves; ++i) offs[i + 1] = offs[i] + wave_cnt[i];
    const int total = min(offs[kRpWaves], S);
    if (total == 0) {
        if (threadIdx.x == 0) pooled_empty_flag[(size_t)bs * boxes_num + box] = 1;  // :146-148
        return;  // rows of an empty box are left as the caller initialised them (:177-179)
    }
    for (int s = threadIdx.x; s < S; s += kRpThreads) {
        const int r = s % total;
        int src = 0;
#pragma unroll
        for (int i = 0; i < kRpWaves; ++i)
            if (r >= offs[i] && r < offs[i + 1]) src = lists[i * S + (r - offs[i])];
        final_idx[s] = src;
    }
    __syncthreads();

    // phase 3: copy xyz + features of the S sampled points, one row per wave at a time
    const int row = 3 + feature_in_len;
    float *dst_base = pooled_features + ((size_t)bs * boxes_num + box) * S * row;
    // kRpRows rows per wave and step: their gathers are in flight together (the copy is a chain of dependent
    // loads -- list entry, then the row -- not a bandwidth problem at these sizes)
    constexpr int kRpRows = 16;
    for (int s0 = wave * kRpRows; s0 < S; s0 += kRpWaves * kRpRows) {
        for (int e = lane; e < row; e += 64) {
            float v[kRpRows];
#pragma unroll
            for (int u = 0; u < kRpRows; ++u) {
                const int src = final_idx[min(s0 + u, S - 1)];
                v[u] = e < 3 ? xyz[(size_t)src * 3 + e] : pts_feature[(size_t)src * feature_in_len + (e - 3)];
            }
#pragma unroll
            for (int u = 0; u < kRpRows; ++u)
                if (s0 + u < S) dst_base[(size_t)(s0 + u) * row + e] = v[u];
        }
    }
}

}  // namespace epnet

using namespace epnet;

extern "C" size_t epnet_roipool3d_workspace_bytes(int, int, int) { return 0; }  // fused kernel: no scratch

extern "C" int epnet_roipool3d(int batch_size, int pts_num, int boxes_num, int feature_in_len, int sampled_pts_num,
                               const float *xyz, const float *boxes3d, const float *pts_feature, float *pooled_features,
                               int *pooled_empty_flag, void *, size_t, epnet_stream_t stream) {
    EPNET_REQUIRE(batch_size >= 0 && pts_num >= 0 && boxes_num >= 0 && feature_in_len >= 0 && sampled_pts_num >= 0);
    if (batch_size == 0 || boxes_num == 0) return EPNET_OK;
    EPNET_REQUIRE(boxes3d && pooled_empty_flag && (sampled_pts_num == 0 || pooled_features));
    EPNET_REQUIRE(pts_num == 0 || (xyz && (pts_feature || feature_in_len == 0)));
    if (batch_size > 65535) return EPNET_ELIMIT;
    const size_t lds = (size_t)(kRpWaves + 1) * sampled_pts_num * sizeof(int);
    if (lds > 150 * 1024) return EPNET_ELIMIT;
    hipLaunchKernelGGL(roipool3d_kernel, dim3(boxes_num, batch_size), dim3(kRpThreads), lds, (hipStream_t)stream, pts_num,
                       boxes_num, feature_in_len, sampled_pts_num, xyz, boxes3d, pts_feature, pooled_features,
                       pooled_empty_flag);
    return check_launch("roipool3d");
}
