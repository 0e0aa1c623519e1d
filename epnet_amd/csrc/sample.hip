// sample.hip -- LI-Fusion's point-to-pixel sampler for gfx950.
//
// The reference samples an image feature map at the projected pixel of every point with
// torch.nn.functional.grid_sample(feature_map (B,C,H,W), xy (B,1,N,2)) -> (B,C,1,N)  (Feature_Gather,
// lib/net/pointnet2_msg.py:107-120; bilinear, zero padding), after picking the xy of the sampled points with
// torch.gather over the FPS indices (:214-217). The stock kernel gives one thread a POINT and lets it walk the C
// channels: with 64 ... 4096 points per scene and 64 ... 512 channels most of the chip idles (0.40 ms for the
// (2, 512, 24, 80) map and 64 points). Here one thread owns a (point, channel) pair -- consecutive lanes are
// consecutive points of one channel, so the output is written coalesced and the four taps of neighbouring points
// share cache lines -- and the xy gather is folded in (idx != NULL).
//
// Arithmetic = grid_sampler_2d of PyTorch (aten/src/ATen/native/GridSampler.h, cuda/GridSampler.cu), bilinear / zeros:
//   ix = align_corners ? (x + 1) / 2 * (W - 1) : ((x + 1) * W - 1) / 2          (same for iy, H)
//   nw = (floor(ix), floor(iy)), ne = nw + (1, 0), sw = nw + (0, 1), se = nw + (1, 1)
//   weights  nw: (ix_se - ix) * (iy_se - iy)   ne: (ix - ix_sw) * (iy_sw - iy)
//            sw: (ix_ne - ix) * (iy - iy_ne)   se: (ix - ix_nw) * (iy - iy_nw)
//   out = sum over the in-bounds corners, in the order nw, ne, sw, se, of value * weight
// The backward w.r.t. the feature map adds grad * weight to the same four pixels (float atomics).
#include "common.h"

namespace epnet {

constexpr int kSampleThreads = 256;

struct Taps {
    int x0, y0;            // north-west corner
    float nw, ne, sw, se;  // bilinear weights
};

__device__ __forceinline__ Taps taps_of(float x, float y, int h, int w, int align_corners) {
    float ix, iy;
    if (align_corners) {
        ix = ((x + 1.f) / 2.f) * (float)(w - 1);
        iy = ((y + 1.f) / 2.f) * (float)(h - 1);
    } else {
        ix = ((x + 1.f) * (float)w - 1.f) / 2.f;
        iy = ((y + 1.f) * (float)h - 1.f) / 2.f;
    }
    float fx = floorf(ix), fy = floorf(iy);
    // a tap can only be inside the map for -1 <= floor <= size - 1; anything else (far outside, infinite, NaN) is moved to
    // a corner that has no in-bounds tap, so that the int conversion below is always defined
    if (!(fx >= -1.f && fx <= (float)(w - 1) && fy >= -1.f && fy <= (float)(h - 1))) {
        fx = -2.f;
        fy = -2.f;
        ix = -2.f;
        iy = -2.f;
    }
    Taps t;
    t.x0 = (int)fx;
    t.y0 = (int)fy;
    const float x_e = fx + 1.f, y_s = fy + 1.f;
    t.nw = (x_e - ix) * (y_s - iy);
    t.ne = (ix - fx) * (y_s - iy);
    t.sw = (x_e - ix) * (iy - fy);
    t.se = (ix - fx) * (iy - fy);
    return t;
}

__device__ __forceinline__ bool inside(int x, int y, int h, int w) { return x >= 0 && y >= 0 && x < w && y < h; }

// grid: (ceil(n / 256), channel chunks, b); thread = one point, loops over its chunk's channels
__global__ __launch_bounds__(kSampleThreads) void feature_gather_kernel(int c, int h, int w, int n_src, int n, int chunk,
                                                                        int align_corners, const float *__restrict__ fmap,
                                                                        const float *__restrict__ xy,
                                                                        const int *__restrict__ idx, float *__restrict__ out,
                                                                        float *__restrict__ xy_out) {
    const int bs = blockIdx.z;
    const int q = blockIdx.x * kSampleThreads + threadIdx.x;
    if (q >= n) return;
    const int src = idx ? idx[(size_t)bs * n + q] : q;
    const float x = xy[((size_t)bs * n_src + src) * 2], y = xy[((size_t)bs * n_src + src) * 2 + 1];
    if (xy_out && blockIdx.y == 0) {
        xy_out[((size_t)bs * n + q) * 2] = x;
        xy_out[((size_t)bs * n + q) * 2 + 1] = y;
    }
    const Taps t = taps_of(x, y, h, w, align_corners);
    const bool i_nw = inside(t.x0, t.y0, h, w), i_ne = inside(t.x0 + 1, t.y0, h, w), i_sw = inside(t.x0, t.y0 + 1, h, w),
               i_se = inside(t.x0 + 1, t.y0 + 1, h, w);
    const long long o_nw = (long long)t.y0 * w + t.x0;
    const int c0 = blockIdx.y * chunk, c1 = min(c, c0 + chunk);
    const float *plane = fmap + ((size_t)bs * c + c0) * h * w;
    float *dst = out + ((size_t)bs * c + c0) * n + q;
    // the four taps of a channel are requested together: a tap outside the map reads the map's first pixel instead and its
    // product is dropped (a load under `if (inside)` is waited for before the next one is issued: four serial round trips)
    const long long a_nw = i_nw ? o_nw : 0, a_ne = i_ne ? o_nw + 1 : 0, a_sw = i_sw ? o_nw + w : 0, a_se = i_se ? o_nw + w + 1 : 0;
    for (int ci = c0; ci < c1; ++ci) {
        const float p_nw = plane[a_nw], p_ne = plane[a_ne], p_sw = plane[a_sw], p_se = plane[a_se];
        float v = 0.f;
        v += i_nw ? p_nw * t.nw : 0.f;
        v += i_ne ? p_ne * t.ne : 0.f;
        v += i_sw ? p_sw * t.sw : 0.f;
        v += i_se ? p_se * t.se : 0.f;
        *dst = v;
        plane += (size_t)h * w;
        dst += n;
    }
}

__global__ __launch_bounds__(kSampleThreads) void feature_gather_grad_kernel(int c, int h, int w, int n, int chunk,
                                                                             int align_corners, const float *__restrict__ grad_out,
                                                                             const float *__restrict__ xy,
                                                                             float *__restrict__ grad_fmap) {
    const int bs = blockIdx.z;
    const int q = blockIdx.x * kSampleThreads + threadIdx.x;
    if (q >= n) return;
    const float x = xy[((size_t)bs * n + q) * 2], y = xy[((size_t)bs * n + q) * 2 + 1];
    const Taps t = taps_of(x, y, h, w, align_corners);
    const bool i_nw = inside(t.x0, t.y0, h, w), i_ne = inside(t.x0 + 1, t.y0, h, w), i_sw = inside(t.x0, t.y0 + 1, h, w),
               i_se = inside(t.x0 + 1, t.y0 + 1, h, w);
    const long long o_nw = (long long)t.y0 * w + t.x0;
    const int c0 = blockIdx.y * chunk, c1 = min(c, c0 + chunk);
    float *plane = grad_fmap + ((size_t)bs * c + c0) * h * w;
    const float *src = grad_out + ((size_t)bs * c + c0) * n + q;
    for (int ci = c0; ci < c1; ++ci) {
        const float g = *src;
        if (i_nw) atomicAdd(plane + o_nw, g * t.nw);
        if (i_ne) atomicAdd(plane + o_nw + 1, g * t.ne);
        if (i_sw) atomicAdd(plane + o_nw + w, g * t.sw);
        if (i_se) atomicAdd(plane + o_nw + w + 1, g * t.se);
        plane += (size_t)h * w;
        src += n;
    }
}

// channels per block row: enough blocks to fill the chip (>= ~2048 workgroups) without giving a thread fewer than 4 channels
static int channel_chunk(int b, int c, int n) {
    const int point_blocks = div_up(n, kSampleThreads) * b;
    int chunks = div_up(2048, point_blocks);
    if (chunks > div_up(c, 4)) chunks = div_up(c, 4);
    if (chunks < 1) chunks = 1;
    return div_up(c, chunks);
}

}  // namespace epnet

using namespace epnet;

extern "C" int epnet_feature_gather(int b, int c, int h, int w, int n_src, int n, int align_corners, const float *feature_map,
                                    const float *xy, const int *idx, float *out, float *xy_out, epnet_stream_t stream) {
    EPNET_REQUIRE(b >= 0 && c >= 0 && h >= 0 && w >= 0 && n >= 0 && n_src >= 0);
    if (b == 0 || n == 0) return EPNET_OK;
    EPNET_REQUIRE(xy && (c == 0 || (feature_map && out)) && h > 0 && w > 0 && (idx || n_src == n));
    if (b > 65535 || (long long)h * w > 0x7fffffffll) return EPNET_ELIMIT;
    const int cc = c > 0 ? c : 1;
    const int chunk = channel_chunk(b, cc, n);
    hipLaunchKernelGGL(feature_gather_kernel, dim3(div_up(n, kSampleThreads), div_up(cc, chunk), b), dim3(kSampleThreads), 0,
                       (hipStream_t)stream, c, h, w, n_src, n, chunk, align_corners, feature_map, xy, idx, out, xy_out);
    return check_launch("feature_gather");
}

extern "C" int epnet_feature_gather_grad(int b, int c, int h, int w, int n, int align_corners, const float *grad_out,
                                         const float *xy, float *grad_feature_map, epnet_stream_t stream) {
    EPNET_REQUIRE(b >= 0 && c >= 0 && h >= 0 && w >= 0 && n >= 0);
    if (b == 0 || n == 0 || c == 0) return EPNET_OK;
    EPNET_REQUIRE(grad_out && xy && grad_feature_map && h > 0 && w > 0);
    if (b > 65535 || (long long)h * w > 0x7fffffffll) return EPNET_ELIMIT;
    const int chunk = channel_chunk(b, c, n);
    hipLaunchKernelGGL(feature_gather_grad_kernel, dim3(div_up(n, kSampleThreads), div_up(c, chunk), b), dim3(kSampleThreads), 0,
                       (hipStream_t)stream, c, h, w, n, chunk, align_corners, grad_out, xy, grad_feature_map);
    return check_launch("feature_gather_grad");
}
