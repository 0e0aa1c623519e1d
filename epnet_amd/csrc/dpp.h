// dpp.h -- wave64 reductions on the vector ALU (DPP row steps + gfx950 row / half swaps): results land in
// every lane, no LDS (ds_bpermute) and no scalar round trip.
#pragma once
#include "common.h"

namespace epnet {

template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}

// max over each row of 16 lanes, result in every lane of the row
__device__ __forceinline__ int row16_max(int v) {
    v = max(v, dpp_i32<0xB1>(v));   // quad_perm [1,0,3,2]
    v = max(v, dpp_i32<0x4E>(v));   // quad_perm [2,3,0,1]
    v = max(v, dpp_i32<0x124>(v));  // row_ror:4
    v = max(v, dpp_i32<0x128>(v));  // row_ror:8
    return v;
}
__device__ __forceinline__ unsigned row16_min(unsigned v) {
    v = min(v, dpp_u32<0xB1>(v));
    v = min(v, dpp_u32<0x4E>(v));
    v = min(v, dpp_u32<0x124>(v));
    v = min(v, dpp_u32<0x128>(v));
    return v;
}

__device__ __forceinline__ int wave_max_all(int v) {
    v = row16_max(v);
    const auto a = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    v = max((int)a[0], (int)a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return max((int)b[0], (int)b[1]);
}

__device__ __forceinline__ unsigned wave_min_all(unsigned v) {
    v = row16_min(v);
    const auto a = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    v = min(a[0], a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return min(b[0], b[1]);
}

// float min / max over the 64 lanes, result in every lane (finite inputs; bit moves on the int pattern)
__device__ __forceinline__ float wave_minf_all(float v) {
    v = fminf(v, __int_as_float(dpp_i32<0xB1>(__float_as_int(v))));
    v = fminf(v, __int_as_float(dpp_i32<0x4E>(__float_as_int(v))));
    v = fminf(v, __int_as_float(dpp_i32<0x124>(__float_as_int(v))));
    v = fminf(v, __int_as_float(dpp_i32<0x128>(__float_as_int(v))));
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fminf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fminf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float wave_maxf_all(float v) { return -wave_minf_all(-v); }

// float min / max over each row of 16 lanes, result in every lane of the row
__device__ __forceinline__ float row16_minf(float v) {
    v = fminf(v, __int_as_float(dpp_i32<0xB1>(__float_as_int(v))));
    v = fminf(v, __int_as_float(dpp_i32<0x4E>(__float_as_int(v))));
    v = fminf(v, __int_as_float(dpp_i32<0x124>(__float_as_int(v))));
    v = fminf(v, __int_as_float(dpp_i32<0x128>(__float_as_int(v))));
    return v;
}
__device__ __forceinline__ float row16_maxf(float v) {
    v = fmaxf(v, __int_as_float(dpp_i32<0xB1>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(dpp_i32<0x4E>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(dpp_i32<0x124>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(dpp_i32<0x128>(__float_as_int(v))));
    return v;
}

// The boxes of the four 16-lane rows of a wave's points (x, y, z per lane): per-row minimum and maximum of each coordinate in every
// lane of the row. Written as 24 v_min / v_max with the DPP lane move folded into the instruction: through fminf / fmaxf the compiler
// emits three instructions per step (DPP move, a canonicalising v_max x,x the IEEE mode asks of minnum, the min itself). The six
// chains are interleaved, so every instruction reads registers written at least five instructions earlier (a DPP read needs two
// wait states after the write; the s_nops cover the inputs and whatever reads the results next). All lanes of the wave must be active. A NaN coordinate drops
// out of its row's box (v_min / v_max return the other operand), exactly like fminf / fmaxf.
__device__ __forceinline__ void row16_boxes(float x, float y, float z, float &lx, float &hx, float &ly, float &hy, float &lz,
                                            float &hz) {
#define EPNET_ROWBOX_STEP(ctl)                                         \
    "v_min_f32_dpp %0, %0, %0 " ctl " row_mask:0xf bank_mask:0xf\n"     \
    "v_max_f32_dpp %1, %1, %1 " ctl " row_mask:0xf bank_mask:0xf\n"     \
    "v_min_f32_dpp %2, %2, %2 " ctl " row_mask:0xf bank_mask:0xf\n"     \
    "v_max_f32_dpp %3, %3, %3 " ctl " row_mask:0xf bank_mask:0xf\n"     \
    "v_min_f32_dpp %4, %4, %4 " ctl " row_mask:0xf bank_mask:0xf\n"     \
    "v_max_f32_dpp %5, %5, %5 " ctl " row_mask:0xf bank_mask:0xf\n"
    lx = x; hx = x; ly = y; hy = y; lz = z; hz = z;
    asm volatile("s_nop 1\n" EPNET_ROWBOX_STEP("quad_perm:[1,0,3,2]") EPNET_ROWBOX_STEP("quad_perm:[2,3,0,1]")
                 EPNET_ROWBOX_STEP("row_ror:4") EPNET_ROWBOX_STEP("row_ror:8") "s_nop 1"
                 : "+v"(lx), "+v"(hx), "+v"(ly), "+v"(hy), "+v"(lz), "+v"(hz));
#undef EPNET_ROWBOX_STEP
}

// v_min_f32 / v_max_f32 as they are (through fminf / fmaxf every operand is canonicalised first: three instructions per step).
// Like fminf / fmaxf they return the other operand for a quiet NaN; a signalling NaN would come out quieted instead, so
// canonicalise inputs of unknown origin once (canonical()) before a chain of these.
__device__ __forceinline__ float min_raw(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float max_raw(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float canonical(float v) { return __builtin_canonicalizef(v); }

// Box of the wave's 64 points, result in every lane: lo* enter as the lanes' values for the minima, hi* for the maxima (a lane
// that holds no point passes +-3.4e38; NaNs must be quiet: canonical()). 24 DPP steps inside the rows of 16 (as row16_boxes),
// then the two row / half swaps: 48 vector instructions for the six numbers (wave_minf_all / wave_maxf_all: ~130).
__device__ __forceinline__ void wave_box(float &lx, float &hx, float &ly, float &hy, float &lz, float &hz) {
#define EPNET_WAVEBOX_STEP(ctl)                                        \
    "v_min_f32_dpp %0, %0, %0 " ctl " row_mask:0xf bank_mask:0xf\n"     \
    "v_max_f32_dpp %1, %1, %1 " ctl " row_mask:0xf bank_mask:0xf\n"     \
    "v_min_f32_dpp %2, %2, %2 " ctl " row_mask:0xf bank_mask:0xf\n"     \
    "v_max_f32_dpp %3, %3, %3 " ctl " row_mask:0xf bank_mask:0xf\n"     \
    "v_min_f32_dpp %4, %4, %4 " ctl " row_mask:0xf bank_mask:0xf\n"     \
    "v_max_f32_dpp %5, %5, %5 " ctl " row_mask:0xf bank_mask:0xf\n"
    asm volatile("s_nop 1\n" EPNET_WAVEBOX_STEP("quad_perm:[1,0,3,2]") EPNET_WAVEBOX_STEP("quad_perm:[2,3,0,1]")
                 EPNET_WAVEBOX_STEP("row_ror:4") EPNET_WAVEBOX_STEP("row_ror:8") "s_nop 1"
                 : "+v"(lx), "+v"(hx), "+v"(ly), "+v"(hy), "+v"(lz), "+v"(hz));
#undef EPNET_WAVEBOX_STEP
    auto across = [](float v, bool is_min) {
        const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = is_min ? min_raw(__uint_as_float(a[0]), __uint_as_float(a[1])) : max_raw(__uint_as_float(a[0]), __uint_as_float(a[1]));
        const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return is_min ? min_raw(__uint_as_float(b[0]), __uint_as_float(b[1])) : max_raw(__uint_as_float(b[0]), __uint_as_float(b[1]));
    };
    lx = across(lx, true);  hx = across(hx, false);
    ly = across(ly, true);  hy = across(hy, false);
    lz = across(lz, true);  hz = across(hz, false);
}

}  // namespace epnet
