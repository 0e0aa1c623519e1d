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
