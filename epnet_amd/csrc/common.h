// common.h -- shared host/device helpers of libepnet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "epnet_ops.h"

namespace epnet {

constexpr int kWave = 64;  // CDNA4 wavefront

// Records the HIP error text for epnet_last_hip_error() and maps it to EPNET_ELAUNCH.
int record_hip_error(hipError_t e, const char *where);

inline int check_launch(const char *where) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return record_hip_error(e, where);
    return EPNET_OK;
}

inline int div_up(int a, int b) { return (a + b - 1) / b; }
inline long long div_up64(long long a, long long b) { return (a + b - 1) / b; }

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ int popc_below(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

// streaming (non-temporal) 16-byte store for outputs that are written once and not re-read by the kernel:
// keeps the gathered rows resident in L2 (+8 % on the grouping kernel at 256 scenes)
__device__ __forceinline__ void store_stream(float *dst, float x, float y, float z, float w) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = {x, y, z, w};
    __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(dst));
}

// XCD-aware relabelling of a (gridDim.x workgroups per scene, gridDim.y scenes) launch. Workgroups are dealt round-robin over the 8
// XCDs by their linear id, and every XCD has an L2 of its own: with the plain (blockIdx.x, blockIdx.y) mapping the eight
// neighbours in launch order -- which all work on the SAME scene -- sit on eight different XCDs, and each of them pulls its own
// copy of that scene's tables (scene index, coordinates, feature rows) through the fabric: 8x the compulsory fetch traffic on
// whatever the workgroups of a scene share (level-1 ball query: 2.9 MB moved per scene for 1.1 MB of index + output).
// Here XCD k takes the k-th eighth of the linear ids in order, i.e. whole scenes one after the other: a scene's tables pass
// through ONE L2. A bijection of the ids; placement is a matter of speed only (nothing assumes where a workgroup runs).
// EPNET_XCD_MAP=0 in the environment of the build (-DEPNET_NO_XCD_MAP) keeps the plain mapping for measurements.
__device__ __forceinline__ void xcd_scene_map(int &wg_x, int &bs) {
#ifdef EPNET_NO_XCD_MAP
    wg_x = (int)blockIdx.x;
    bs = (int)blockIdx.y;
#else
    const int gx = (int)gridDim.x, nwg = gx * (int)gridDim.y, orig = (int)blockIdx.y * gx + (int)blockIdx.x;
    const int xcd = orig & 7, qq = nwg >> 3, rr = nwg & 7;
    const int wgid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (orig >> 3);
    bs = wgid / gx;
    wg_x = wgid - bs * gx;
#endif
}

// the same for a (gridDim.x, gridDim.y) grid of workgroups per scene and gridDim.z scenes
__device__ __forceinline__ void xcd_scene_map3(int &wg_x, int &wg_y, int &bs) {
#ifdef EPNET_NO_XCD_MAP
    wg_x = (int)blockIdx.x;
    wg_y = (int)blockIdx.y;
    bs = (int)blockIdx.z;
#else
    const int gx = (int)gridDim.x, per_scene = gx * (int)gridDim.y, nwg = per_scene * (int)gridDim.z;
    const int orig = ((int)blockIdx.z * (int)gridDim.y + (int)blockIdx.y) * gx + (int)blockIdx.x;
    const int xcd = orig & 7, qq = nwg >> 3, rr = nwg & 7;
    const int wgid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (orig >> 3);
    bs = wgid / per_scene;
    const int in_scene = wgid - bs * per_scene;
    wg_y = in_scene / gx;
    wg_x = in_scene - wg_y * gx;
#endif
}

// sum over the 64 lanes of a wave, result in every lane
__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

}  // namespace epnet

#define EPNET_REQUIRE(cond) \
    do {                    \
        if (!(cond)) return EPNET_EINVAL; \
    } while (0)
