// fps.hip -- furthest point sampling for gfx950.
//
// Replaces furthest_point_sampling_kernel / _launcher of the reference
// (pointnet2_lib/pointnet2/src/sampling_gpu.cu:94-253). One workgroup per scene, as there, but the
// M-1 dependent arg-max iterations are rebuilt around the CDNA4 execution model. Four kernels:
//
//   fps_wave_kernel      N <= 1024 (and the brute-force fallback): the scene lives in VGPRs; lane / slot order
//                        equals the reference's tie-break order, so the wave arg-max is a DPP max + ballot + ff1;
//   pruned::fps_*_kernel 1024 < N <= 16384: registers again, points in spatial order, exact bucket pruning
//                        (see "pruned kernel" below) -- over a caller-built scene index or self-sorting;
//   fps_bigscene_kernel  16384 < N <= 65536: bucket summaries in registers, points re-read from the index;
//   fps_stream_kernel    anything else: xyz / temp re-read through L2.
// Common to all but the last: NO global memory access inside the round loop (the selected indices are buffered in
// LDS and flushed with coalesced stores), one barrier per round (the reference has 11 and a dependent global load).
//
// Tie-breaking. The reference's result depends on its block size bs = opt_n_threads(N)
// (cuda_utils.h:10-14): thread tid scans k = tid, tid+bs, ... keeping the FIRST maximum (strict
// '>', :136-137), and at every level of the shared-memory tree the LOWER slot wins a tie (:86-91).
// Among equal distances the winner is therefore the point with the smallest
//         rank(k) = (bitreverse_{log2 bs}(k mod bs), k div bs)      (lexicographic).
// Physical thread q of this kernel holds the reference thread tid = bitreverse(q), slot j holds
// k = tid + j*bs, so rank order == (q, j) order == (wave, lane, slot) order: the winner among tied
// lanes is the lowest set bit of a ballot, among tied waves the lowest wave, among tied slots the
// lowest slot. Squared distances are >= +0 and never NaN for finite inputs, so comparing their
// bit patterns for equality is comparing the floats.
//
// Arithmetic: d = dx*dx + dy*dy + dz*dz evaluated left to right in fp32 without contraction
// (-ffp-contract=off), then min with the running distance, as sampling_gpu.cu:133-135.
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "dpp.h"
#include "spatial.h"

// clang exposes no builtin for v_writelane_b32; bind the LLVM intrinsic by name (value, lane, old)
extern "C" __device__ int epnet_llvm_writelane_i32(int, int, int) __asm("llvm.amdgcn.writelane.i32");

namespace epnet {

__device__ __forceinline__ int writelane_i32(int value, int lane, int old) {
    return epnet_llvm_writelane_i32(value, lane, old);
}

__device__ __forceinline__ unsigned bitrev_lg(unsigned v, int lg) {
    return lg == 0 ? 0u : (__brev(v) >> (32 - lg));
}

// ---- DPP helpers -------------------------------------------------------------------------------
// Distances are handled as their int32 bit patterns: for non-negative floats (and the -1.0f padding
// value, which is a negative int) signed integer order == float order and bit equality == float
// equality, and v_min_i32 / v_max3_i32 need none of the NaN-canonicalising v_max the compiler has to
// put in front of every fminf / fmaxf.
// wave-wide max, returned wave-uniform (scalar registers)
__device__ __forceinline__ int wave_max_i32(int v) {
    const int r = row16_max(v);
    const int a = __builtin_amdgcn_readlane(r, 0), b = __builtin_amdgcn_readlane(r, 16);
    const int c = __builtin_amdgcn_readlane(r, 32), d = __builtin_amdgcn_readlane(r, 48);
    return max(max(a, b), max(c, d));
}

constexpr int kIdxBuf = 4096;  // selected (thread, slot) codes buffered in LDS between flushes

// (thread q, slot s) -> point index. Thread q owns R = bs/T consecutive rank positions r = q*R + s/J
// (reference thread tid = bitreverse(r)) and, of each, the J = ceil(n/bs) points k = tid + j*bs.
__device__ __forceinline__ int fps_point_of(int q, int slot, int R, int J, int lg_bs) {
    const int r = q * R + slot / J;
    return (int)bitrev_lg((unsigned)r, lg_bs) + (slot % J) * (1 << lg_bs);
}

// first slot (lowest index) of lane `wl` whose distance equals `target`, and that point's coordinates
// -- all wave-uniform. Two-level search over groups of 4 slots for PPT >= 8.
template <int PPT>
__device__ __forceinline__ void find_slot(const int (&t)[PPT], const float (&x)[PPT], const float (&y)[PPT],
                                          const float (&z)[PPT], const int (&g)[(PPT + 3) / 4], int wl, int target,
                                          int &slot, float &sx, float &sy, float &sz) {
    slot = 0;
    sx = sy = sz = 0.f;
    if constexpr (PPT >= 8) {
        constexpr int NG = PPT / 4;
        int grp = 0;
#pragma unroll
        for (int gi = NG - 1; gi >= 0; --gi)
            if (__builtin_amdgcn_readlane(g[gi], wl) == target) grp = gi;  // wave-uniform
#pragma unroll
        for (int gi = 0; gi < NG; ++gi)
            if (grp == gi) {
#pragma unroll
                for (int j = 4 * gi + 3; j >= 4 * gi; --j)
                    if (__builtin_amdgcn_readlane(t[j], wl) == target) slot = j;
            }
#pragma unroll
        for (int j = 0; j < PPT; ++j)
            if (slot == j) {
                sx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x[j]), wl));
                sy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y[j]), wl));
                sz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z[j]), wl));
            }
    } else {
#pragma unroll
        for (int j = PPT - 1; j >= 0; --j)
            if (__builtin_amdgcn_readlane(t[j], wl) == target) {
                slot = j;
                sx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x[j]), wl));
                sy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y[j]), wl));
                sz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z[j]), wl));
            }
    }
}

// What fps_prefix_kernel does for a scene, as the first thing of a sampling kernel (one workgroup per scene) that can do it
// itself -- one dispatch less on the sampling chain of every level, where a dispatch beside the wide kernels of the pipelined
// stack waits for wave slots (a 6 us prefix kernel was seen to take 77 us there). init < 0: a separate launch has done it.
// Returns whether the scene's first m points are its samples already.
__device__ __forceinline__ bool fps_prologue(int m, const int *__restrict__ skip, int *__restrict__ idx_scene,
                                             int *__restrict__ prefix_out, int init) {
    const int have = skip ? skip[blockIdx.x] : 0;
    if (init >= 0) {
        if (threadIdx.x == 0 && prefix_out) prefix_out[blockIdx.x] = have >= m ? have : init;
        // (a scene whose rounds do run rewrites its indices itself; the known prefix is the same there)
        for (int i = threadIdx.x; i < min(have, m); i += blockDim.x) idx_scene[i] = i;
    }
    return have >= m;
}

// Register-resident kernel: W waves per scene, PPT point slots per thread. 64*W <= bs = 2^lg_bs,
// R = bs / (64*W), J = ceil(n / bs), R*J <= PPT.
template <int W, int PPT>
__global__ __launch_bounds__(64 * W) void fps_wave_kernel(int n, int m, int lg_bs, const float *__restrict__ xyz,
                                                          float *__restrict__ temp, int *__restrict__ idxs,
                                                          const int *__restrict__ skip, int *__restrict__ prefix_out = nullptr,
                                                          int prologue_init = -1) {
    // this scene's first m points are the samples already?
    if (fps_prologue(m, skip, idxs + (size_t)blockIdx.x * m, prefix_out, prologue_init)) return;
    __shared__ int s_val[2][16];     // per-wave maximum (bit pattern)
    __shared__ float4 s_rec[2][16];  // per-wave candidate: x, y, z, (q << 8 | slot) as int bits
    __shared__ int s_idx[kIdxBuf];
    constexpr int T = 64 * W;
    const int q = threadIdx.x;
    const int lane = q & 63, wave = q >> 6;
    const int bs = 1 << lg_bs;
    const int R = bs / T, J = (n + bs - 1) / bs;
    const int used = R * J;
    xyz += (size_t)blockIdx.x * n * 3;
    if (temp) temp += (size_t)blockIdx.x * n;
    idxs += (size_t)blockIdx.x * m;
    const int kNeg1 = __float_as_int(-1.f);

    float x[PPT], y[PPT], z[PPT];
    int t[PPT];  // running min squared distance, as bits
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
        const int k = s < used ? fps_point_of(q, s, R, J, lg_bs) : n;
        if (k < n) {
            x[s] = xyz[k * 3 + 0];
            y[s] = xyz[k * 3 + 1];
            z[s] = xyz[k * 3 + 2];
            t[s] = __float_as_int(temp ? temp[k] : 1e10f);
        } else {  // padding slot: distance pinned at -1, never a maximum (slot 0 of every thread is real)
            x[s] = y[s] = z[s] = 0.f;
            t[s] = kNeg1;
        }
    }
    if (W > 1 && q < 32) s_val[q >> 4][q & 15] = kNeg1;
    if (q == 0) s_idx[0] = 0;  // code of (thread 0, slot 0) == point 0
    float x1 = xyz[0], y1 = xyz[1], z1 = xyz[2];
    __syncthreads();

    for (int it = 1; it < m; ++it) {
        int g[(PPT + 3) / 4];
        int best = kNeg1;
        if constexpr (PPT >= 8) {
#pragma unroll
            for (int gi = 0; gi < PPT / 4; ++gi) {
                int gm = kNeg1;
#pragma unroll
                for (int j = 4 * gi; j < 4 * gi + 4; ++j) {
                    const float dx = x[j] - x1, dy = y[j] - y1, dz = z[j] - z1;
                    const float d = dx * dx + dy * dy + dz * dz;
                    t[j] = min(__float_as_int(d), t[j]);  // == fminf(d, temp[k]) for d >= 0 (padding: -1 stays)
                    gm = max(gm, t[j]);
                }
                g[gi] = gm;
                best = max(best, gm);
            }
        } else {
#pragma unroll
            for (int j = 0; j < PPT; ++j) {
                const float dx = x[j] - x1, dy = y[j] - y1, dz = z[j] - z1;
                const float d = dx * dx + dy * dy + dz * dz;
                t[j] = min(__float_as_int(d), t[j]);
                best = max(best, t[j]);
            }
            g[0] = best;
        }
        // wave arg-max: value by DPP, owner = lowest tied lane, slot = lowest tied slot of that lane
        const int wbest = wave_max_i32(best);
        const unsigned long long tied = __ballot(best == wbest);
        const int wl = (int)__builtin_ctzll(tied);
        int slot;
        float sx, sy, sz;
        find_slot<PPT>(t, x, y, z, g, wl, wbest, slot, sx, sy, sz);
        const int code = ((wave * 64 + wl) << 8) | slot;
        if constexpr (W == 1) {
            x1 = sx;
            y1 = sy;
            z1 = sz;
            s_idx[it & (kIdxBuf - 1)] = code;  // all 64 lanes, same word
        } else {
            const int buf = it & 1;
            if (lane == 0) {
                s_val[buf][wave] = wbest;
                s_rec[buf][wave] = make_float4(sx, sy, sz, __int_as_float(code));
            }
            __syncthreads();  // no global traffic is in flight here: this waits on LDS only
            // one LDS round trip: lane l < 16 fetches wave l's maximum AND record; the winner's record is
            // then picked out of the registers with readlane
            const int wv = s_val[buf][lane & 15];
            const float4 rec = s_rec[buf][lane & 15];
            const int bm = row16_max(wv);
            const unsigned long long wtied = __ballot(wv == bm) & 0xFFFFull;
            const int ww = (int)__builtin_ctzll(wtied);  // lowest tied wave
            x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rec.x), ww));
            y1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rec.y), ww));
            z1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rec.z), ww));
            // every lane of wave 0 stores the (uniform) winner code: a store guarded by `q == 0` would let
            // the compiler sink the LDS load feeding the readlane into a single-lane region, where the
            // lane being read has not loaded anything
            const int wcode = __builtin_amdgcn_readlane(__float_as_int(rec.w), ww);
            if (wave == 0) s_idx[it & (kIdxBuf - 1)] = wcode;
        }
        if ((it & (kIdxBuf - 1)) == kIdxBuf - 1 && wave == 0) {  // buffer full: wave 0 (the writer) flushes it
            const int base = it - (kIdxBuf - 1);
            for (int e = lane; e < kIdxBuf; e += 64) {
                const int c = s_idx[e];
                idxs[base + e] = fps_point_of(c >> 8, c & 0xFF, R, J, lg_bs);
            }
        }
    }
    if (wave == 0) {
        const int base = (m - 1) & ~(kIdxBuf - 1);
        for (int e = lane; base + e < m; e += 64) {
            const int c = s_idx[e];
            idxs[base + e] = fps_point_of(c >> 8, c & 0xFF, R, J, lg_bs);
        }
    }
    if (temp) {
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            const int k = s < used ? fps_point_of(q, s, R, J, lg_bs) : n;
            if (k < n) temp[k] = __int_as_float(t[s]);
        }
    }
}

// ---- pruned kernel ------------------------------------------------------------------------------
//
// Exact FPS with spatial pruning, for 1024 < N <= 16384. The points are sorted by grid cell (a scene index
// built beforehand, or an in-kernel counting sort on an interleaved cell code, spatial.h) and dealt in groups of
// 64 consecutive sorted points to (wave, slot) registers: group g is slot g/W of wave g%W, one point per lane.
// Every slot is split into 64/PPT buckets of PPT lanes; a bucket's summary (bounding box, maximum running
// distance bm) lives in one lane of the bucket itself. A new sample c can only lower distances in buckets whose
// box lies closer than their bm, so per round a wave
//   A. evaluates L = |clamp(c, box) - c|^2 for all its 64 buckets at once (one bucket per lane),
//   B. updates only the slots holding a bucket with L < bm (typically 0-3; the slot registers are addressed
//      with the gfx9 GPR-index mode, so there is one copy of the code and no branch tree),
//   C. reduces its bucket maxima; every thread holding the wave's maximum publishes its point (coordinates
//      into its record slot, (distance, rank, thread) into one 64-bit LDS atomic max),
//   D. after the single barrier reads the winning key and that thread's coordinates.
// Everything after the `active` mask stays on the vector ALU (DPP / permlane-swap reductions, lane
// masks instead of readlane -> SALU -> VALU round trips).
// Skipping is EXACT, not approximate: L_j is computed with the very expression used for point
// distances (dx*dx + dy*dy + dz*dz, fp32, no contraction) on the clamped point; fp32 subtraction,
// multiplication and addition are monotone, so every point of the bucket has d >= L_j >= bm_j >= its
// running distance and min(d, t) = t bit for bit. Tie-breaks cannot follow the lane order here (the
// layout is spatial): every maximum carries the reference rank of its point,
// rank(k) = (bitreverse10(k mod 1024) << 4) | (k div 1024)  (block size 1024 for every N > 1024),
// and equal distances are resolved by the smaller rank inside a bucket, between buckets and between
// waves -- exactly the winner of the reference's strided scan + shared-memory tree.
namespace pruned {

#ifndef EPNET_FPS_PRIO
#define EPNET_FPS_PRIO 3
#endif

#ifdef EPNET_FPS_STATS  // diagnostic build only (scratch/fps_stats.hip): phase cycle counters
__device__ unsigned long long g_stats[16];
#define EPNET_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define EPNET_ACC(slot, a, b) st_acc[slot] += (unsigned long long)((b) - (a))
#define EPNET_CNT(slot, v) st_acc[slot] += (unsigned long long)(v)
#define EPNET_STATS_BEGIN unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define EPNET_STATS_END if ((threadIdx.x & 63) == 0) for (int s_ = 0; s_ < 8; ++s_) atomicAdd(&g_stats[s_], st_acc[s_])
#else
#define EPNET_STAMP(var)
#define EPNET_ACC(slot, a, b)
#define EPNET_CNT(slot, v)
#define EPNET_STATS_BEGIN
#define EPNET_STATS_END
#endif

__device__ __forceinline__ unsigned rank14(int k) { return (bitrev_lg((unsigned)k & 1023u, 10) << 4) | ((unsigned)k >> 10); }
__device__ __forceinline__ int unrank14(unsigned r) { return (int)(bitrev_lg(r >> 4, 10) + ((r & 15u) << 10)); }

// max over aligned groups of G lanes (G = 8, 16 or 32), result in every lane of the group
template <int G>
__device__ __forceinline__ int group_max(int v) {
    v = max(v, dpp_i32<0xB1>(v));                  // quad_perm [1,0,3,2]
    v = max(v, dpp_i32<0x4E>(v));                  // quad_perm [2,3,0,1]
    v = max(v, dpp_i32<0x141>(v));                 // row_half_mirror: the other quad of the 8-lane half
    if (G >= 16) v = max(v, dpp_i32<0x140>(v));    // row_mirror: the other half of the row
    if (G >= 32) {
        const auto a = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
        v = max((int)a[0], (int)a[1]);
    }
    return v;
}
template <int G>
__device__ __forceinline__ float group_minf(float v) {
    v = fminf(v, __int_as_float(dpp_i32<0xB1>(__float_as_int(v))));
    v = fminf(v, __int_as_float(dpp_i32<0x4E>(__float_as_int(v))));
    v = fminf(v, __int_as_float(dpp_i32<0x141>(__float_as_int(v))));
    if (G >= 16) v = fminf(v, __int_as_float(dpp_i32<0x140>(__float_as_int(v))));
    if (G >= 32) {
        const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = fminf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    }
    return v;
}
template <int G>
__device__ __forceinline__ float group_maxf(float v) { return -group_minf<G>(-v); }

// data replicated with period P (a power of two <= 16) along the lanes: reduce over one period
template <int P>
__device__ __forceinline__ int period_max(int v) {
    if (P > 1) v = max(v, dpp_i32<0xB1>(v));
    if (P > 2) v = max(v, dpp_i32<0x4E>(v));
    if (P > 4) v = max(v, dpp_i32<0x124>(v));
    if (P > 8) v = max(v, dpp_i32<0x128>(v));
    return v;
}
template <int P>
__device__ __forceinline__ unsigned period_min(unsigned v) {
    if (P > 1) v = min(v, dpp_u32<0xB1>(v));
    if (P > 2) v = min(v, dpp_u32<0x4E>(v));
    if (P > 4) v = min(v, dpp_u32<0x124>(v));
    if (P > 8) v = min(v, dpp_u32<0x128>(v));
    return v;
}

// fold a 64-lane ballot over the 64/PPT parts of a slot: bit j set <=> some part of slot j is set
template <int PPT>
__device__ __forceinline__ unsigned fold_parts(unsigned long long a) {
    if (PPT <= 32) a |= a >> 32;
    if (PPT <= 16) a |= a >> 16;
    if (PPT <= 8) a |= a >> 8;
    return (unsigned)a & (PPT >= 32 ? 0xFFFFFFFFu : ((1u << PPT) - 1u));
}

constexpr int kIdxBufP = 1024;  // selected ranks buffered in LDS between flushes

// reference ranks of a thread's PPT points, two 16-bit ranks per register (0xFFFF = padding)
template <typename VH>
__device__ __forceinline__ unsigned rank_of(const VH &rk2, int j) {
    return ((unsigned)rk2[j >> 1] >> ((j & 1) << 4)) & 0xFFFFu;
}
template <typename VH>
__device__ __forceinline__ void set_rank(VH &rk2, int j, unsigned r) {  // j is a compile-time constant at the call sites
    rk2[j >> 1] = (j & 1) ? (int)(((unsigned)rk2[j >> 1] & 0xFFFFu) | (r << 16)) : (int)r;
}

// The M-1 rounds. W waves per scene (4: one per SIMD; 8 for N > 8192), PPT slots of 64 points per wave;
// thread (wave, lane) holds in slot j the point at sorted position ((j*kW + wave) << 6) | lane: coordinates,
// running distance t (bits; -1.0f = padding) and reference rank rk.
// A slot is split into 64/PPT parts of PPT lanes: these parts are the pruning buckets, and bucket
// (slot j, part p) is summarised in lane p*PPT + j -- a lane of the part itself, so a part's maximum
// reaches its summary lane with an in-row DPP reduction and a lane-id compare, no cross-lane move.
// Cross-wave arg-max: every thread holding its wave's maximum enters ONE 64-bit LDS atomic max on the key
// (distance bits << 32 | (0x3FFF - rank) << 10 | thread): the larger distance wins, equal distances go to the
// smaller reference rank -- and after the round's single barrier the winner is read back with two
// dependent LDS loads (key, then that thread's coordinates), no cross-lane reduction at all.
// kCtr: also emit the selected points' coordinates (ctr, m x 3 floats): the gather that follows the sampling
// in an SA module (pointnet2_modules.py:39-45) comes for free, the round's winner is in registers anyway.
template <int kW, int PPT, bool kCtr, typename VF, typename VI, typename VH>
__device__ __forceinline__ void fps_rounds(int m, const VF &x, const VF &y, const VF &z, VI &t, const VH &rk2,
                                           float cx, float cy, float cz, int *__restrict__ idxs,
                                           float *__restrict__ ctr, int *__restrict__ tie_free = nullptr, int known = 0,
                                           const float *__restrict__ xyz_in_order = nullptr, int detect_upto = 0x7fffffff) {
    __shared__ unsigned long long s_key[3];
    __shared__ float4 s_rec[2][64 * kW];  // one record slot per thread
    __shared__ int s_idx[kIdxBufP];
    __shared__ float s_ctr[kCtr ? kIdxBufP * 3 : 1];
    const int q = threadIdx.x;
    const int lane = q & 63, wave = q >> 6;
    const int sub = lane & (PPT - 1);  // the slot this lane summarises (for its own part)
    const int kNeg1 = __float_as_int(-1.f);

    // ---- bucket summaries
    int bm = kNeg1;
    float lox = 0.f, hix = 0.f, loy = 0.f, hiy = 0.f, loz = 0.f, hiz = 0.f;
    for (int j = 0; j < PPT; ++j) {  // runtime loop, indexed registers: one copy of the code
        const float xj = x[j], yj = y[j], zj = z[j];
        const int tj = t[j];
        const bool real = tj != kNeg1;
        const float mnx = group_minf<PPT>(real ? xj : 3.4e38f), mxx = group_maxf<PPT>(real ? xj : -3.4e38f);
        const float mny = group_minf<PPT>(real ? yj : 3.4e38f), mxy = group_maxf<PPT>(real ? yj : -3.4e38f);
        const float mnz = group_minf<PPT>(real ? zj : 3.4e38f), mxz = group_maxf<PPT>(real ? zj : -3.4e38f);
        const int gm = group_max<PPT>(tj);  // -1 for an all-padding bucket: never active, never best
        if (sub == j) {
            const bool any = gm != kNeg1;
            bm = gm;
            lox = any ? mnx : 0.f; hix = any ? mxx : 0.f;
            loy = any ? mny : 0.f; hiy = any ? mxy : 0.f;
            loz = any ? mnz : 0.f; hiz = any ? mxz : 0.f;
        }
    }
    if (q < 3) s_key[q] = 0ull;
    if (q == 0) s_idx[0] = 0;  // rank 0 == point 0
    if (kCtr && q < 3) s_ctr[q] = q == 0 ? cx : (q == 1 ? cy : cz);
    // `known` leading samples are given (points 0 .. known-1 of the cloud in its own order: a sampling pyramid's chain,
    // epnet_sample_centres_chain): their rounds need no selection, only the distance updates -- no cross-wave exchange, no barrier
    if (known > kIdxBufP || known > m || !xyz_in_order || kCtr) known = 0;
    for (int i = q; i < known; i += 64 * kW) s_idx[i] = (int)rank14(i);
    __syncthreads();
    for (int it = 1; it < known; ++it) {
        const float px = __builtin_amdgcn_fmed3f(cx, lox, hix), py = __builtin_amdgcn_fmed3f(cy, loy, hiy),
                    pz = __builtin_amdgcn_fmed3f(cz, loz, hiz);
        const float bdx = px - cx, bdy = py - cy, bdz = pz - cz;
        const float L = bdx * bdx + bdy * bdy + bdz * bdz;
        unsigned active = fold_parts<PPT>(__ballot(__float_as_int(L) < bm));
        const float nx = xyz_in_order[it * 3 + 0], ny = xyz_in_order[it * 3 + 1], nz = xyz_in_order[it * 3 + 2];  // the next sample
        while (active) {
            const int j = (int)__builtin_ctz(active);
            active &= active - 1u;
            const float xj = x[j], yj = y[j], zj = z[j];
            const int told = t[j];
            __builtin_amdgcn_sched_barrier(0);
            const float dx = xj - cx, dy = yj - cy, dz = zj - cz;
            const float d = dx * dx + dy * dy + dz * dz;
            const int tj = min(__float_as_int(d), told);
            t[j] = tj;
            const int gm = group_max<PPT>(tj);
            bm = (sub == j) ? gm : bm;
        }
        cx = nx;
        cy = ny;
        cz = nz;
    }

    // a strictly serial chain: when it shares a SIMD with a wide kernel (software-pipelined SA stack), every
    // instruction it has ready should issue first
    __builtin_amdgcn_s_setprio(EPNET_FPS_PRIO);
    EPNET_STATS_BEGIN;
    EPNET_STAMP(t_loop0);
    // this wave's best point(s), recomputed only in rounds that update a bucket holding one of them
    bool stale = true;
    int wbest = kNeg1;
    unsigned long long hbuckets = 0ull;  // buckets (by summary lane) in which this lane holds the wave's maximum
    unsigned racc = 0xFFFFFFFFu;  // != ~0 in the lanes that publish
    float xa = 0.f, ya = 0.f, za = 0.f;
    int kb = 1;  // key slot of the round = it % 3
    // the first round in which the reference's tie-break decides WHICH COORDINATES are sampled: several points at the maximum
    // running distance that are not all exact twins of one another (or a maximum of zero: every point coincides with a sample).
    // Up to that round the sequence of sampled coordinates is a property of the coordinates alone (epnet_sample_centres_chain).
    // Exact twins -- the reference's loader pads short scenes by re-drawing rows, kitti_rcnn_dataset.py:338-342 -- tie at the
    // round one of them is picked, harmlessly: the others sit at distance 0 from then on and cannot be sampled while the
    // maximum is positive.
    bool wave_multi = false;  // several points of this wave hold its maximum
    int tied_v = 0x7fffffff;  // wave-uniform, kept scalar
    // one round; kTies: also look for a second holder of the round's maximum (only the rounds a later level can ask about pay
    // for that: the two instantiations of the body are run one after the other)
    auto round = [&](auto ties_tag, const int it) __attribute__((always_inline)) {
        constexpr bool kTies = decltype(ties_tag)::value;
        EPNET_STAMP(t0);
        // A. which buckets can change?
        const float px = __builtin_amdgcn_fmed3f(cx, lox, hix), py = __builtin_amdgcn_fmed3f(cy, loy, hiy),
                    pz = __builtin_amdgcn_fmed3f(cz, loz, hiz);
        const float bdx = px - cx, bdy = py - cy, bdz = pz - cz;
        const float L = bdx * bdx + bdy * bdy + bdz * bdz;
        const unsigned long long act64 = __ballot(__float_as_int(L) < bm);  // bit = summary lane of an active bucket
        unsigned active = fold_parts<PPT>(act64);
        EPNET_CNT(0, __popc(active));
        // distances only fall: the wave's maximum and its holders stand unless one of THEIR buckets is updated
        stale = stale || __ballot((hbuckets & act64) != 0ull) != 0ull;
        EPNET_STAMP(t1);
        // B. update the slots that contain one (handling two slots per iteration was measured: 20 % slower)
        while (active) {
            const int j = (int)__builtin_ctz(active);
            active &= active - 1u;
            const float xj = x[j], yj = y[j], zj = z[j];
            const int told = t[j];
            __builtin_amdgcn_sched_barrier(0);  // keep the indexed reads together: one GPR-index window
            const float dx = xj - cx, dy = yj - cy, dz = zj - cz;
            const float d = dx * dx + dy * dy + dz * dz;
            const int tj = min(__float_as_int(d), told);  // == fminf(d, temp[k]); padding stays at -1
            t[j] = tj;
            const int gm = group_max<PPT>(tj);
            bm = (sub == j) ? gm : bm;
        }
        EPNET_STAMP(t2);
        // C. this wave's maximum. EVERY lane holding it (usually exactly one) publishes its point: its own record
        // slot and the atomic key, which orders equal distances by reference rank -- no holder count, no second
        // reduction, whatever the number of ties
        if (stale) {
            stale = false;
            wbest = wave_max_all(bm);
            unsigned cand = fold_parts<PPT>(__ballot(bm == wbest));
            racc = 0xFFFFFFFFu;
            hbuckets = 0ull;
            do {
                const int j = (int)__builtin_ctz(cand);
                cand &= cand - 1u;
                const int tj = t[j];
                const float xj = x[j], yj = y[j], zj = z[j];
                const unsigned rw = (unsigned)rk2[j >> 1];
                __builtin_amdgcn_sched_barrier(0);  // one GPR-index window for the four slot registers
                const unsigned r = tj == wbest ? ((rw >> ((j & 1) << 4)) & 0xFFFFu) : 0xFFFFFFFFu;
                if (r != 0xFFFFFFFFu) hbuckets |= 1ull << ((lane & ~(PPT - 1)) | j);  // summary lane of (slot j, my part)
                const bool take = r < racc;  // a lane holding the maximum in two of its slots keeps the smaller rank
                racc = take ? r : racc;
                xa = take ? xj : xa;
                ya = take ? yj : ya;
                za = take ? zj : za;
            } while (cand);
            if (wbest == kNeg1) {  // a wave of padding only
                racc = 0xFFFFFFFFu;
                hbuckets = 0ull;
            }
            if (kTies) {
                // (hbuckets has one bit per slot in which the lane holds the maximum)
                const unsigned long long holders = __ballot(hbuckets != 0ull);
                wave_multi = (holders & (holders - 1ull)) != 0ull || __ballot((hbuckets & (hbuckets - 1ull)) != 0ull) != 0ull;
            }
        }
        const int buf = it & 1;
        if (racc != 0xFFFFFFFFu) {
            s_rec[buf][q] = make_float4(xa, ya, za, 0.f);
            const unsigned long long key =
                ((unsigned long long)(unsigned)wbest << 32) | (unsigned long long)(((0x3FFFu - racc) << 10) | (unsigned)q);
            __hip_atomic_fetch_max(&s_key[kb], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        EPNET_STAMP(t3);
        __syncthreads();
        EPNET_STAMP(t4);
        // D. the winner: key, then its thread's coordinates (both loads are wave-uniform broadcasts)
        const unsigned long long kfull = s_key[kb];
        const unsigned klo = (unsigned)kfull;
        const float4 rec = s_rec[buf][klo & 1023u];
        cx = rec.x;
        cy = rec.y;
        cz = rec.z;
        if (kTies) {
            // this wave's (exact, possibly cached) maximum equals the winner's distance and the winner is not its only holder: a
            // tie -- unless every other holder is an exact twin of the winner. The first test is what every round pays (a handful
            // of instructions; the twin rule computed branch-free on every round cost 0.1 us on each of the 1024 watched rounds of
            // the level-1 sampling: 2.05 -> 2.15 ms); the coordinates are compared behind a wave-uniform branch that only a round
            // with several holders takes.
            const bool mine = (int)((klo & 1023u) >> 6) == wave;
            const int wd = (int)(unsigned)(kfull >> 32);
            const bool suspect = (wbest == wd && (!mine || wave_multi)) || wd == 0;
            if (__builtin_amdgcn_readfirstlane(suspect ? 1 : 0)) {
                // every slot in which a lane holds the maximum (bit (lane's part | j) of hbuckets) against the winner's coordinates
                // (the winner's own slot compares equal to itself)
                bool differs = false;
                const unsigned mine_slots = (unsigned)((hbuckets >> (lane & ~(PPT - 1))) & ((1ull << PPT) - 1ull));   // (PPT <= 32)
                unsigned any_slots = mine_slots;   // slots in which ANY lane of the wave holds the maximum
                for (int off = 32; off >= 1; off >>= 1) any_slots |= (unsigned)__shfl_xor((int)any_slots, off, 64);
                // a runtime loop with indexed registers: one copy of the code, no registers beyond the round's own
#pragma unroll 1
                while (any_slots) {
                    const int j = (int)__builtin_ctz(any_slots);
                    any_slots &= any_slots - 1u;
                    const float xj = x[j], yj = y[j], zj = z[j];
                    __builtin_amdgcn_sched_barrier(0);
                    differs = differs || (((mine_slots >> j) & 1u) && (xj != cx || yj != cy || zj != cz));
                }
                if (wd == 0 || __ballot(differs)) tied_v = min(tied_v, it);
            }
        }
        const int kb2 = kb == 0 ? 2 : kb - 1;  // == (it + 2) % 3: last read in round it-1, next used in round it+2
        kb = kb == 2 ? 0 : kb + 1;
        if (wave == 0) {
            s_idx[it & (kIdxBufP - 1)] = (int)(0x3FFFu - (klo >> 10));  // all lanes, same word; converted at the flush
            if (lane == 0) s_key[kb2] = 0ull;
            if (kCtr && lane < 3) s_ctr[(it & (kIdxBufP - 1)) * 3 + lane] = lane == 0 ? cx : (lane == 1 ? cy : cz);
            if ((it & (kIdxBufP - 1)) == kIdxBufP - 1) {
                const int base = it - (kIdxBufP - 1);
                for (int e = lane; e < kIdxBufP; e += 64) idxs[base + e] = unrank14((unsigned)s_idx[e]);
                if (kCtr) {
#pragma unroll 1
                    for (int e = lane; e < kIdxBufP * 3; e += 64) ctr[(size_t)base * 3 + e] = s_ctr[e];
                }
            }
        }
        EPNET_STAMP(t5);
        EPNET_ACC(1, t0, t1); EPNET_ACC(2, t1, t2); EPNET_ACC(3, t2, t3); EPNET_ACC(4, t3, t4); EPNET_ACC(5, t4, t5);
    };
    int it0 = max(1, known);
    if (tie_free) {
        const int upto = min(m, detect_upto);
        for (; it0 < upto; ++it0) round(std::true_type{}, it0);
    }
    for (; it0 < m; ++it0) round(std::false_type{}, it0);
    EPNET_STAMP(t_loop1);
    EPNET_ACC(6, t_loop0, t_loop1);
    EPNET_STATS_END;
    if (tie_free && lane == 0) atomicMin(tie_free, min(min(tied_v, m), detect_upto));  // (initialised to m by the caller's launch sequence)
    if (wave == 0) {
        const int base = (m - 1) & ~(kIdxBufP - 1);
        for (int e = lane; base + e < m; e += 64) idxs[base + e] = unrank14((unsigned)s_idx[e]);
        if (kCtr) {
#pragma unroll 1
            for (int e = lane; e < (m - base) * 3; e += 64) ctr[(size_t)base * 3 + e] = s_ctr[e];
        }
    }
}

// Self-contained kernel (no caller scratch): counting-sorts the scene by grid cell in LDS, then runs the rounds.
template <int kW, int PPT>
__global__ __launch_bounds__(64 * kW) void fps_pruned_kernel(int n, int m, const float *__restrict__ xyz,
                                                             float *__restrict__ temp, int *__restrict__ idxs,
                                                             const int *__restrict__ skip, int *__restrict__ prefix_out,
                                                             int prefix_cap) {
    if (skip && skip[blockIdx.x] >= m) return;
    typedef float vecf __attribute__((ext_vector_type(PPT)));
    typedef int veci __attribute__((ext_vector_type(PPT)));
    constexpr int kT = 64 * kW;
    extern __shared__ int s_dyn[];  // cell histogram, then the 16-bit point indices in cell order
    int *hist = s_dyn;
    unsigned short *perm = reinterpret_cast<unsigned short *>(s_dyn + kCells + kCells / (kCells / kT) + 64);
    __shared__ int s_part[16];
    __shared__ float s_box[6][16];
    const int q = threadIdx.x;
    const int lane = q & 63, wave = q >> 6;
    xyz += (size_t)blockIdx.x * n * 3;
    if (temp) temp += (size_t)blockIdx.x * n;
    idxs += (size_t)blockIdx.x * m;

    // ---- spatial order: counting sort by grid cell (spatial.h); positions >= n are padding
    float lo[3], ext[3];
    block_bbox3(xyz, n, s_box, lo, ext);
    const CellGrid grid = make_cell_grid(lo, ext);
    cell_sort_lds(xyz, n, grid, hist, s_part, perm);

    // ---- sorted position p = (group << 6 | lane); group g of 64 sorted points -> (wave g % kW, slot g / kW)
    veci kk;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int p = ((j * kW + wave) << 6) | lane;
        kk[j] = p < n ? (int)perm[p] : -1;
    }
    typedef int vech __attribute__((ext_vector_type(PPT / 2)));
    vecf x, y, z;
    veci t;
    vech rk2;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int k = kk[j];
        if (k >= 0) {
            x[j] = xyz[k * 3 + 0];
            y[j] = xyz[k * 3 + 1];
            z[j] = xyz[k * 3 + 2];
            t[j] = __float_as_int(temp ? temp[k] : 1e10f);
            set_rank(rk2, j, rank14(k));
        } else {  // padding: distance pinned at -1
            x[j] = y[j] = z[j] = 0.f;
            t[j] = __float_as_int(-1.f);
            set_rank(rk2, j, 0xFFFFu);
        }
    }
    fps_rounds<kW, PPT, false>(m, x, y, z, t, rk2, xyz[0], xyz[1], xyz[2], idxs, nullptr,
                               prefix_out ? prefix_out + blockIdx.x : nullptr, 0, nullptr, prefix_cap);
    if (temp) {
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const unsigned r = rank_of(rk2, j);
            if (r != 0xFFFFu) temp[unrank14(r)] = __int_as_float(t[j]);
        }
    }
}

// Same rounds over a scene index built beforehand (spatial.h: cell-sorted float4 copy x, y, z, original index;
// padding entries carry index -1). Needs no dynamic LDS, so it shares a CU with the bandwidth-bound kernels.
template <int kW, int PPT, bool kCtr>
__global__ __launch_bounds__(64 * kW) void fps_indexed_kernel(int n, int m, const float4 *__restrict__ sorted,
                                                              float *__restrict__ temp, int *__restrict__ idxs,
                                                              float *__restrict__ ctr, const int *__restrict__ prefix_in,
                                                              int *__restrict__ prefix_out, const float *__restrict__ xyz,
                                                              int prefix_cap, int prologue_init) {
    typedef float vecf __attribute__((ext_vector_type(PPT)));
    typedef int veci __attribute__((ext_vector_type(PPT)));
    constexpr int NP = 64 * kW * PPT;
    __shared__ float s_first[4];
    const int q = threadIdx.x;
    const int lane = q & 63, wave = q >> 6;
    const int known = prefix_in ? prefix_in[blockIdx.x] : 0;
    // the first m points ARE the samples?
    if (fps_prologue(m, prefix_in, idxs + (size_t)blockIdx.x * m, prefix_out, prologue_init)) return;
    sorted += (size_t)blockIdx.x * NP;
    if (temp) temp += (size_t)blockIdx.x * n;
    idxs += (size_t)blockIdx.x * m;
    if (kCtr) ctr += (size_t)blockIdx.x * m * 3;
    typedef int vech __attribute__((ext_vector_type(PPT / 2)));
    vecf x, y, z;
    veci t;
    vech rk2;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const float4 v = sorted[((j * kW + wave) << 6) | lane];
        const int k = __float_as_int(v.w);
        if (k >= 0) {
            x[j] = v.x;
            y[j] = v.y;
            z[j] = v.z;
            t[j] = __float_as_int(temp ? temp[k] : 1e10f);
            set_rank(rk2, j, rank14(k));
            if (k == 0) {  // the first sample is point 0 (sampling_gpu.cu:118)
                s_first[0] = v.x;
                s_first[1] = v.y;
                s_first[2] = v.z;
            }
        } else {
            x[j] = y[j] = z[j] = 0.f;
            t[j] = __float_as_int(-1.f);
            set_rank(rk2, j, 0xFFFFu);
        }
    }
    __syncthreads();
    fps_rounds<kW, PPT, kCtr>(m, x, y, z, t, rk2, s_first[0], s_first[1], s_first[2], idxs, ctr,
                              prefix_out ? prefix_out + blockIdx.x : nullptr, known,
                              xyz ? xyz + (size_t)blockIdx.x * n * 3 : nullptr, prefix_cap);
    if (temp) {
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const unsigned r = rank_of(rk2, j);
            if (r != 0xFFFFu) temp[unrank14(r)] = __int_as_float(t[j]);
        }
    }
}

// ---- scenes beyond the register file (16384 < N <= 65536), over a scene index ------------------------
// The points stay in the index (L2-resident, <= 1 MB) and the running distances in the caller's temp buffer;
// only the bucket summaries live in registers: lane j of wave w owns bucket j * 16 + w (64 consecutive sorted points) -- its
// box (from the index), its maximum running distance bm and the rank and coordinates of the point holding it.
// A round tests all buckets against the new sample with the same exact lower bound, re-reads and updates only
// the active buckets (one wave per bucket, a coalesced 1 KB row of the index), and finds the arg-max over the
// summaries exactly as fps_rounds does (wave maximum, 64-bit LDS atomic key, one barrier).
// rank(k) = (bitreverse10(k mod 1024) << 6) | (k div 1024) -- 16 bits for k < 65536.
__device__ __forceinline__ unsigned rank16(int k) { return (bitrev_lg((unsigned)k & 1023u, 10) << 6) | ((unsigned)k >> 10); }
__device__ __forceinline__ int unrank16(unsigned r) { return (int)(bitrev_lg(r >> 6, 10) + ((r & 63u) << 10)); }

// kW waves per scene, every lane owning 1024 / (64 kW) buckets. Measured (65536 -> 16384, one scene / 256 scenes):
// 16 waves 0.88 / 1.18 us per round, 8 waves 1.11 / 1.47, 4 waves 1.50 / 1.97 -- the round is a chain of latencies through ONE
// wave (winner read-back -> box tests -> row load -> update + bucket maximum -> the wave's arg-max -> LDS atomic -> barrier), and
// a wave with four buckets per lane runs four box tests and up to four times the row updates back to back; the instructions
// the sixteen waves issue in total are not what bounds it. 16 is the default (EPNET_FPS_BIG_WAVES = 4 | 8 keeps the others).
template <int kW>
__global__ __launch_bounds__(64 * kW) void fps_bigscene_kernel(int n, int np, int m, const float *__restrict__ xyz,
                                                               const float4 *__restrict__ sorted,
                                                               const float *__restrict__ boxes,
                                                               float *__restrict__ temp, float *__restrict__ tsort,
                                                               int *__restrict__ idxs, const int *__restrict__ skip) {
    if (skip && skip[blockIdx.x] >= m) return;
    constexpr int kT = 64 * kW;
    constexpr int kBPL = 1024 / kT;   // buckets per lane (np <= 65536: at most 1024 buckets)
    constexpr int kLgW = kW == 16 ? 4 : (kW == 8 ? 3 : 2);
    static_assert(kW == 4 || kW == 8 || kW == 16, "bucket dealing assumes 4, 8 or 16 waves");
    __shared__ unsigned long long s_key[3];
    __shared__ float4 s_rec[2][kW];
    __shared__ int s_idx[kIdxBufP];
    const int q = threadIdx.x;
    const int lane = q & 63, wave = q >> 6;
    const int nb = np >> 6;
    xyz += (size_t)blockIdx.x * n * 3;
    sorted += (size_t)blockIdx.x * np;
    boxes += (size_t)blockIdx.x * nb * 6;
    temp += (size_t)blockIdx.x * n;
    tsort += (size_t)blockIdx.x * np;
    idxs += (size_t)blockIdx.x * m;
    const int kNeg1 = __float_as_int(-1.f);
    // Bucket e (= 0 .. 64 kBPL - 1) of this wave is bucket (e * kW + wave) of the scene; lane e % 64 keeps its summary in slot
    // e / 64. Consecutive buckets of the sorted order are spatial neighbours, and a new sample touches a handful of neighbouring
    // buckets -- dealt round-robin they land on different waves, which re-read them side by side instead of one wave working
    // through them one L2 round trip after the other
    auto bid = [&](int e) { return (e << kLgW) | wave; };

    float lox[kBPL], hix[kBPL], loy[kBPL], hiy[kBPL], loz[kBPL], hiz[kBPL];
    bool own[kBPL];
    int bm[kBPL];               // maximum running distance of my bucket (bits); -1: nothing real in it
    unsigned brank[kBPL];       // reference rank of the point holding it
    float bxx[kBPL], byy[kBPL], bzz[kBPL];
#pragma unroll
    for (int s_ = 0; s_ < kBPL; ++s_) {
        const int b_ = bid(s_ * 64 + lane);
        own[s_] = b_ < nb;
        lox[s_] = hix[s_] = loy[s_] = hiy[s_] = loz[s_] = hiz[s_] = 0.f;
        if (own[s_]) {
            const float *bx = boxes + b_ * 6;
            lox[s_] = bx[0]; hix[s_] = bx[1]; loy[s_] = bx[2]; hiy[s_] = bx[3]; loz[s_] = bx[4]; hiz[s_] = bx[5];
        }
        bm[s_] = kNeg1;
        brank[s_] = 0xFFFFu;
        bxx[s_] = byy[s_] = bzz[s_] = 0.f;
    }
    const int nmine = 64 * kBPL;   // bucket slots of this wave

    // The caller's running distances are indexed by ORIGINAL point number: a bucket's 64 values would be 64 cache lines.
    // The rounds keep them in SORTED order instead, in the sampling scratch at the tail of the scene index (one 256-byte
    // row per bucket, fetched beside the bucket's 1 KB row of the index); gathered on the way in, scattered back on the
    // way out. (The original index of a point rides in the .w of its index row: nothing per point is kept in registers --
    // a workgroup that holds few registers leaves the rest of the CU to the bandwidth-bound kernels beside it.)
    for (int e = 0; e < nmine; ++e) {
        const int pos = (bid(e) << 6) + lane;
        if (pos < n) tsort[pos] = temp[__float_as_int(sorted[pos].w)];
    }

    struct Row {
        float4 p;
        int t;
    };
    auto load_row = [&](int e) {
        Row r;
        const int pos = (bid(e) << 6) + lane;
        r.p = sorted[pos];
        r.t = pos < n ? __float_as_int(tsort[pos]) : kNeg1;
        return r;
    };
    // (re)computes the summary of bucket e of this wave from its row; with `update`, first lowers its distances by the sample c
    auto finish = [&](int e, Row r, bool update, float cx, float cy, float cz) {
        const int pos = (bid(e) << 6) + lane;
        const int k = __float_as_int(r.p.w);   // original index (padding rows: -1, never `real`)
        const bool real = pos < n;
        int t = r.t;
        if (real && update) {
            const float dx = r.p.x - cx, dy = r.p.y - cy, dz = r.p.z - cz;
            const float d = dx * dx + dy * dy + dz * dz;
            const int tn = min(__float_as_int(d), t);  // == fminf(d, temp[k])
            if (tn != t) tsort[pos] = __int_as_float(tn);
            t = tn;
        }
        const int mx = wave_max_all(t);
        const unsigned rk = (t == mx && real) ? rank16(k) : 0xFFFFFFFFu;
        unsigned long long holders = __ballot(rk != 0xFFFFFFFFu);
        unsigned rwin = 0xFFFFu;
        int wl = 0;
        if (holders) {
            if (holders & (holders - 1ull)) holders = __ballot(rk == wave_min_all(rk));  // several: the smallest rank
            wl = (int)__builtin_ctzll(holders);
            rwin = (unsigned)__builtin_amdgcn_readlane((int)rk, wl);
        }
        const float wx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r.p.x), wl));
        const float wy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r.p.y), wl));
        const float wz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r.p.z), wl));
        const bool mine = lane == (e & 63);
        const int slot = e >> 6;   // wave-uniform
#pragma unroll
        for (int s_ = 0; s_ < kBPL; ++s_)
            if (mine && s_ == slot) {
                bm[s_] = mx;
                brank[s_] = rwin;
                bxx[s_] = wx; byy[s_] = wy; bzz[s_] = wz;
            }
    };

    for (int e = 0; e < nmine; ++e)
        if (bid(e) < nb) finish(e, load_row(e), false, 0.f, 0.f, 0.f);  // wave-uniform
    if (q < 3) s_key[q] = 0ull;
    if (q == 0) s_idx[0] = 0;  // rank 0 == point 0
    float cx = xyz[0], cy = xyz[1], cz = xyz[2];
    __syncthreads();

    __builtin_amdgcn_s_setprio(EPNET_FPS_PRIO);
    bool stale = true;
    int wbest = kNeg1;
    bool publisher = false;
    unsigned pub_rank = 0xFFFFu;
    float pub_x = 0.f, pub_y = 0.f, pub_z = 0.f;
    int kb = 1;
    for (int it = 1; it < m; ++it) {
        // A. which buckets can change?  (same fp32 expression as the point distance: exact)
        unsigned long long active[kBPL];
        bool any = false;
#pragma unroll
        for (int s_ = 0; s_ < kBPL; ++s_) {
            const float px = __builtin_amdgcn_fmed3f(cx, lox[s_], hix[s_]), py = __builtin_amdgcn_fmed3f(cy, loy[s_], hiy[s_]),
                        pz = __builtin_amdgcn_fmed3f(cz, loz[s_], hiz[s_]);
            const float bdx = px - cx, bdy = py - cy, bdz = pz - cz;
            const float L = bdx * bdx + bdy * bdy + bdz * bdz;
            active[s_] = __ballot(own[s_] && __float_as_int(L) < bm[s_]);
            any = any || active[s_] != 0ull;
        }
        stale = stale || any;
        // B. re-read and update them, two rows per trip to L2
#pragma unroll
        for (int s_ = 0; s_ < kBPL; ++s_) {
            unsigned long long act = active[s_];
            while (act) {
                const int e0 = s_ * 64 + (int)__builtin_ctzll(act);
                act &= act - 1ull;
                const bool two = act != 0ull;
                const int e1 = two ? s_ * 64 + (int)__builtin_ctzll(act) : e0;
                act &= act - 1ull;  // (no-op on 0)
                const Row r0 = load_row(e0);
                Row r1 = r0;
                if (two) r1 = load_row(e1);
                finish(e0, r0, true, cx, cy, cz);
                if (two) finish(e1, r1, true, cx, cy, cz);
            }
        }
        // C. this wave's best bucket (ties by rank)
        if (stale) {
            stale = false;
            int lmax = bm[0];
#pragma unroll
            for (int s_ = 1; s_ < kBPL; ++s_) lmax = max(lmax, bm[s_]);
            wbest = wave_max_all(lmax);
            // my best candidate among the slots that hold the wave's maximum: the smallest rank
            unsigned lrank = 0xFFFFFFFFu;
            float lx = 0.f, ly = 0.f, lz = 0.f;
#pragma unroll
            for (int s_ = 0; s_ < kBPL; ++s_) {
                const bool take = bm[s_] == wbest && bm[s_] != kNeg1 && brank[s_] < lrank;
                lrank = take ? brank[s_] : lrank;
                lx = take ? bxx[s_] : lx;
                ly = take ? byy[s_] : ly;
                lz = take ? bzz[s_] : lz;
            }
            unsigned long long cand = __ballot(lrank != 0xFFFFFFFFu);
            if (cand & (cand - 1ull)) {
                const unsigned rmin = wave_min_all(lrank);   // (every lane takes part: no short-circuit in front of it)
                cand = __ballot(lrank == rmin);
            }
            publisher = cand != 0ull && lane == (int)__builtin_ctzll(cand);
            pub_rank = lrank;
            pub_x = lx; pub_y = ly; pub_z = lz;
        }
        const int buf = it & 1;
        if (publisher) {
            s_rec[buf][wave] = make_float4(pub_x, pub_y, pub_z, 0.f);
            const unsigned long long key =
                ((unsigned long long)(unsigned)wbest << 32) | (unsigned long long)(((0xFFFFu - pub_rank) << 4) | (unsigned)wave);
            __hip_atomic_fetch_max(&s_key[kb], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();
        // D. the winner
        const unsigned klo = (unsigned)s_key[kb];
        const float4 rec = s_rec[buf][klo & 15u];
        cx = rec.x;
        cy = rec.y;
        cz = rec.z;
        const int kb2 = kb == 0 ? 2 : kb - 1;
        kb = kb == 2 ? 0 : kb + 1;
        if (wave == 0) {
            s_idx[it & (kIdxBufP - 1)] = (int)(0xFFFFu - (klo >> 4));
            if (lane == 0) s_key[kb2] = 0ull;
            if ((it & (kIdxBufP - 1)) == kIdxBufP - 1) {
                const int base = it - (kIdxBufP - 1);
                for (int e = lane; e < kIdxBufP; e += 64) idxs[base + e] = unrank16((unsigned)s_idx[e]);
            }
        }
    }
    if (wave == 0) {
        const int base = (m - 1) & ~(kIdxBufP - 1);
        for (int e = lane; base + e < m; e += 64) idxs[base + e] = unrank16((unsigned)s_idx[e]);
    }
    // the running distances back into the caller's order (a wave reads only what it wrote itself)
    __builtin_amdgcn_s_setprio(0);
    for (int e = 0; e < nmine; ++e) {
        const int pos = (bid(e) << 6) + lane;
        if (pos < n) temp[__float_as_int(sorted[pos].w)] = tsort[pos];
    }
}

}  // namespace pruned

// ---- generic paths (tiny clouds with a reference block < one wave; clouds beyond the register file)

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

// xyz and temp re-read through L2 each iteration; arg-max on the 64-bit key
// (bits(d2) << 32) | (0x7fffffff - rank(k)), whose unique maximum is the reference's winner.
// blockDim.x is a multiple of the reference block size 2^lg_bs, so a thread's points share k mod bs.
__global__ __launch_bounds__(1024) void fps_stream_kernel(int n, int m, int lg_bs, const float *__restrict__ xyz,
                                                          float *__restrict__ temp, int *__restrict__ idxs,
                                                          const int *__restrict__ skip) {
    if (skip && skip[blockIdx.x] >= m) return;
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
        // threads beyond n own no point: best = -1 sorts below every real distance (signed high word)
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

// skip: per scene, "the first skip[b] points of xyz are an unambiguous furthest-point sequence" (scenes with skip[b] >= m are
// left alone: fps_prefix_kernel has written their samples); prefix_out: per scene, receives the number of leading rounds of
// THIS sampling whose maximum was unique (initialised by fps_prefix_kernel) where the kernel can tell. Both may be NULL.
__global__ void fps_prefix_kernel(int m, const int *__restrict__ skip, int *__restrict__ idx, int *__restrict__ prefix_out, int init);

// prologue_init >= 0: skip / prefix_out still want their fps_prefix_kernel treatment with that initial value -- folded into the
// register-resident kernel, launched separately in front of the others
static int fps_plain(int b, int n, int m, const float *xyz, float *temp, int *idx, const int *skip, int *prefix_out,
                     epnet_stream_t stream, int prefix_cap = 0x7fffffff, int prologue_init = -1) {
    EPNET_REQUIRE(b >= 0 && n >= 1 && m >= 0);
    if (b == 0 || m == 0) return EPNET_OK;  // the reference kernel returns at once for m <= 0
    EPNET_REQUIRE(xyz && idx);
    if ((long long)n > (1ll << 20) * 1024) return EPNET_ELIMIT;  // rank field: 20 bits of k div bs
    hipStream_t s = (hipStream_t)stream;
    const int lg = ref_block_lg(n);
    const int bs_ref = 1 << lg;
    dim3 grid(b);
    const int J = div_up(n, bs_ref);
    // 1024 < n <= 16384: exact spatially-pruned kernel (EPNET_FPS_PRUNE=0 forces the brute-force path)
    // (the tuning variables are read on every call, so that the one-process GPU test run reaches every variant)
    const bool prune_enabled = !(getenv("EPNET_FPS_PRUNE") && atoi(getenv("EPNET_FPS_PRUNE")) == 0);
    const int prune_min = getenv("EPNET_FPS_PRUNE_MIN") ? atoi(getenv("EPNET_FPS_PRUNE_MIN")) : 1024;
    auto prefix_first = [&]() -> int {   // the kernels below do not do the prologue themselves
        if (prologue_init < 0) return EPNET_OK;
        hipLaunchKernelGGL(fps_prefix_kernel, dim3(b), dim3(256), 0, s, m, skip, idx, prefix_out, prologue_init);
        return check_launch("sampling prefix");
    };
    if (prune_enabled && n > 1024 && n > prune_min && n <= 16384 && m > 1) {
        if (int rc = prefix_first()) return rc;
        // 64-point slots: 4 waves x {8,16,32} slots, 8 waves above 8192 points (EPNET_FPS_PWAVES overrides)
        int waves = n > 8192 ? 8 : 4;
        if (const char *e = getenv("EPNET_FPS_PWAVES")) {
            const int w = atoi(e);
            if ((w == 2 || w == 4 || w == 8) && div_up(n, 64 * w) <= 32 && div_up(n, 64 * w) >= 1) waves = w;
        }
        const int ppt_need = div_up(n, 64 * waves);
        const int ppt = ppt_need <= 8 ? 8 : ppt_need <= 16 ? 16 : 32;
        const size_t lds = (size_t)(kCells + kCells / (kCells / (64 * waves)) + 64) * sizeof(int) + (size_t)64 * waves * ppt * 2;
#define EPNET_FPS_PRUNED(W_, P_) \
    hipLaunchKernelGGL((pruned::fps_pruned_kernel<W_, P_>), grid, dim3(64 * W_), lds, s, n, m, xyz, temp, idx, skip, prefix_out, prefix_cap)
        if (waves == 8) {
            if (ppt == 8) EPNET_FPS_PRUNED(8, 8);
            else if (ppt == 16) EPNET_FPS_PRUNED(8, 16);
            else EPNET_FPS_PRUNED(8, 32);
        } else if (waves == 2) {
            if (ppt == 8) EPNET_FPS_PRUNED(2, 8);
            else if (ppt == 16) EPNET_FPS_PRUNED(2, 16);
            else EPNET_FPS_PRUNED(2, 32);
        } else {
            if (ppt == 8) EPNET_FPS_PRUNED(4, 8);
            else if (ppt == 16) EPNET_FPS_PRUNED(4, 16);
            else EPNET_FPS_PRUNED(4, 32);
        }
#undef EPNET_FPS_PRUNED
        return check_launch("furthest_point_sampling");
    }
    if (bs_ref >= 64 && J <= 16) {
        // fewest waves whose threads can hold the scene in <= 16 slots each (fewer waves = cheaper
        // cross-wave exchange and more scenes per CU); EPNET_FPS_WAVES overrides for tuning
        int waves = bs_ref / 64;
        for (int w = 1; w <= bs_ref / 64; w *= 2)
            if ((bs_ref / (64 * w)) * J <= 16) {
                waves = w;
                break;
            }
        // few scenes of 512 < n <= 1024 points: the chip is idle anyway and a round is shorter with 2 slots per
        // thread on 8 waves (0.48 us) than with 16 slots on one wave (0.66 us); many small scenes pack best at 1 wave
        if (n > 512 && b <= 512 && bs_ref / 64 >= 8 && waves < 8) waves = 8;
        if (const char *e = getenv("EPNET_FPS_WAVES")) {
            const int w = atoi(e);
            if (w >= 1 && w <= bs_ref / 64 && (w & (w - 1)) == 0 && (bs_ref / (64 * w)) * J <= 16) waves = w;
        }
        const int ppt = (bs_ref / (64 * waves)) * J;
#define EPNET_FPS_LAUNCH(W_, P_) \
    hipLaunchKernelGGL((fps_wave_kernel<W_, P_>), grid, dim3(64 * W_), 0, s, n, m, lg, xyz, temp, idx, skip, prefix_out, prologue_init)
#define EPNET_FPS_PPT(W_)                      \
    do {                                       \
        if (ppt <= 1) EPNET_FPS_LAUNCH(W_, 1); \
        else if (ppt <= 2) EPNET_FPS_LAUNCH(W_, 2); \
        else if (ppt <= 4) EPNET_FPS_LAUNCH(W_, 4); \
        else if (ppt <= 8) EPNET_FPS_LAUNCH(W_, 8); \
        else EPNET_FPS_LAUNCH(W_, 16);         \
    } while (0)
        switch (waves) {
            case 1: EPNET_FPS_PPT(1); break;
            case 2: EPNET_FPS_PPT(2); break;
            case 4: EPNET_FPS_PPT(4); break;
            case 8: EPNET_FPS_PPT(8); break;
            default: EPNET_FPS_PPT(16); break;
        }
#undef EPNET_FPS_PPT
#undef EPNET_FPS_LAUNCH
        return check_launch("furthest_point_sampling");
    }
    // generic path: needs the running distances in memory
    float *tbuf = temp;
    if (!tbuf) return EPNET_EINVAL;  // temp may only be NULL on the register-resident path
    if (int rc = prefix_first()) return rc;
    const int bs = bs_ref < 64 ? 64 : bs_ref;
    hipLaunchKernelGGL(fps_stream_kernel, grid, dim3(bs), 0, s, n, m, lg, xyz, tbuf, idx, skip);
    return check_launch("furthest_point_sampling");
}

extern "C" int epnet_furthest_point_sampling(int b, int n, int m, const float *xyz, float *temp, int *idx,
                                             epnet_stream_t stream) {
    return fps_plain(b, n, m, xyz, temp, idx, nullptr, nullptr, stream, 0x7fffffff);
}

// does the kernel the dispatch above / below picks report its tie-free rounds? (the pruned kernels do)
static bool fps_detects_ties(int n, int m, bool indexed) {
    if (m <= 1 || n <= 1024 || n > 16384) return false;
    if (indexed) return true;
    const bool prune_enabled = !(getenv("EPNET_FPS_PRUNE") && atoi(getenv("EPNET_FPS_PRUNE")) == 0);
    const int prune_min = getenv("EPNET_FPS_PRUNE_MIN") ? atoi(getenv("EPNET_FPS_PRUNE_MIN")) : 1024;
    return prune_enabled && n > prune_min;
}

// scenes whose first m points are known to be the samples (skip[b] >= m): idx = 0 .. m-1, and the knowledge is passed on;
// the others: prefix_out starts at `init` (m where the sampling kernel reports ties by lowering it, 0 where it cannot tell)
__global__ __launch_bounds__(256) void fps_prefix_kernel(int m, const int *__restrict__ skip, int *__restrict__ idx,
                                                         int *__restrict__ prefix_out, int init) {
    const int bs = blockIdx.x;
    const int have = skip ? skip[bs] : 0;
    const bool known = have >= m;
    if (threadIdx.x == 0 && prefix_out) prefix_out[bs] = known ? have : init;
    // (a scene whose rounds do run rewrites its indices itself; the known prefix is the same there)
    for (int i = threadIdx.x; i < min(have, m); i += 256) idx[(size_t)bs * m + i] = i;
}

// rows of the (B,N,3) cloud picked by idx (B,M): the centres of an SA level
__global__ __launch_bounds__(256) void gather_centres_kernel(int n, int m, const float *__restrict__ xyz,
                                                             const int *__restrict__ idx, float *__restrict__ out) {
    const int bs = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;  // element of the (m, 3) output
    if (e >= m * 3) return;
    const int i = e / 3, a = e - i * 3;
    out[(size_t)bs * m * 3 + e] = xyz[((size_t)bs * n + idx[(size_t)bs * m + i]) * 3 + a];
}

// four centres per thread (one 16-byte load of indices, three 16-byte stores): a twelfth of the waves of the kernel above. The
// gather sits on the sampling chain behind every level's FPS, and beside the wide kernels of the pipelined stack what a small
// kernel waits for is wave slots (77 - 118 us for a copy that takes 8 alone)
__global__ __launch_bounds__(256) void gather_centres4_kernel(int n, int m, const float *__restrict__ xyz,
                                                              const int *__restrict__ idx, float *__restrict__ out) {
    const int bs = blockIdx.y;
    const int i4 = blockIdx.x * 256 + threadIdx.x;  // group of four centres
    if (i4 * 4 >= m) return;
    const int4 ids = reinterpret_cast<const int4 *>(idx + (size_t)bs * m)[i4];
    const float *src = xyz + (size_t)bs * n * 3;
    const float *p0 = src + (size_t)ids.x * 3, *p1 = src + (size_t)ids.y * 3, *p2 = src + (size_t)ids.z * 3, *p3 = src + (size_t)ids.w * 3;
    const float a0 = p0[0], a1 = p0[1], a2 = p0[2], b0 = p1[0], b1 = p1[1], b2 = p1[2];
    const float c0 = p2[0], c1 = p2[1], c2 = p2[2], d0 = p3[0], d1 = p3[1], d2 = p3[2];
    float4 *dst = reinterpret_cast<float4 *>(out + ((size_t)bs * m + (size_t)i4 * 4) * 3);
    dst[0] = make_float4(a0, a1, a2, b0);
    dst[1] = make_float4(b1, b2, c0, c1);
    dst[2] = make_float4(c2, d0, d1, d2);
}

static void launch_gather_centres(int b, int n, int m, const float *xyz, const int *idx, float *new_xyz, hipStream_t s) {
    if ((m & 3) == 0 && (((uintptr_t)idx | (uintptr_t)new_xyz) & 15) == 0)
        hipLaunchKernelGGL(gather_centres4_kernel, dim3(div_up(m / 4, 256), b), dim3(256), 0, s, n, m, xyz, idx, new_xyz);
    else
        hipLaunchKernelGGL(gather_centres_kernel, dim3(div_up(m * 3, 256), b), dim3(256), 0, s, n, m, xyz, idx, new_xyz);
}

// shared by the two entry points below; new_xyz may be NULL
static int fps_over_index(int b, int n, int m, const float *xyz, void *index, size_t index_bytes, float *temp, int *idx,
                          float *new_xyz, hipStream_t s, const int *skip = nullptr, int *prefix_out = nullptr,
                          int prefix_cap = 0x7fffffff) {
    const size_t need = scene_index_bytes(b, n);
    bool centres_done = false;
    int rc;
    const bool plain = need == 0 || !index || n <= 1024 || m <= 1 || (n > 16384 && !temp);
    int fold_init = -1;   // >= 0: the known prefixes / the tie-free counts still need their initial treatment (fps_prologue)
    if (skip || prefix_out) {
        EPNET_REQUIRE(idx && b <= 65535);
        if (prefix_cap < 1) prefix_cap = 0x7fffffff;
        fold_init = fps_detects_ties(n, m, !plain) ? m : 0;
    }
    if (fold_init >= 0 && m == 0) {   // (no kernel below runs for m == 0)
        hipLaunchKernelGGL(fps_prefix_kernel, dim3(b), dim3(256), 0, s, m, skip, idx, prefix_out, fold_init);
        return check_launch("sampling prefix");
    }
    // n <= 1024: the reference block size (hence the tie-break rank) depends on n; the one-wave kernel handles it
    if (plain) {
        rc = fps_plain(b, n, m, xyz, temp, idx, skip, prefix_out, (epnet_stream_t)s, prefix_cap, fold_init);
    } else {
        EPNET_REQUIRE(idx);
        if (index_bytes < need) return EPNET_ENOMEM;
        const float4 *sorted = (const float4 *)index;
        dim3 grid(b);
        if (n > 16384) {  // beyond the register file: bucket summaries in registers, points re-read from the index
            EPNET_REQUIRE(xyz);
            if (fold_init >= 0) {
                hipLaunchKernelGGL(fps_prefix_kernel, dim3(b), dim3(256), 0, s, m, skip, idx, prefix_out, fold_init);
                rc = check_launch("sampling prefix");
                if (rc) return rc;
            }
            const int np = scene_index_np(n);
            const float *bucket_boxes = (const float *)(sorted + (size_t)b * np);
            float *tsort = scene_index_sampling_scratch(b, n, index);
            const int big_waves = getenv("EPNET_FPS_BIG_WAVES") ? atoi(getenv("EPNET_FPS_BIG_WAVES")) : 16;
            if (big_waves == 4)
                hipLaunchKernelGGL(pruned::fps_bigscene_kernel<4>, grid, dim3(256), 0, s, n, np, m, xyz, sorted, bucket_boxes, temp, tsort, idx, skip);
            else if (big_waves == 8)
                hipLaunchKernelGGL(pruned::fps_bigscene_kernel<8>, grid, dim3(512), 0, s, n, np, m, xyz, sorted, bucket_boxes, temp, tsort, idx, skip);
            else
                hipLaunchKernelGGL(pruned::fps_bigscene_kernel<16>, grid, dim3(1024), 0, s, n, np, m, xyz, sorted, bucket_boxes, temp, tsort, idx, skip);
        } else {
            const int wide = getenv("EPNET_FPS_WIDE") ? atoi(getenv("EPNET_FPS_WIDE")) : 0;
            // the centres can come out of the sampling kernel itself (kCtr: the round's winner is in registers anyway) or from a
            // small gather afterwards. In-kernel costs wave 0 an LDS write per round and 12 KB more LDS: 1.7 % on one scene,
            // 5 % in the software-pipelined stack -- more than the extra launch, so the separate gather is the default
            const bool ctr_in_kernel = getenv("EPNET_FPS_CTR") && atoi(getenv("EPNET_FPS_CTR")) != 0 && !skip;
#define EPNET_FPS_INDEXED(W_, P_)                                                                                          \
    do {                                                                                                                   \
        if (new_xyz && ctr_in_kernel)                                                                                      \
            hipLaunchKernelGGL((pruned::fps_indexed_kernel<W_, P_, true>), grid, dim3(64 * W_), 0, s, n, m, sorted, temp, \
                               idx, new_xyz, skip, prefix_out, xyz, prefix_cap, fold_init);                                \
        else                                                                                                               \
            hipLaunchKernelGGL((pruned::fps_indexed_kernel<W_, P_, false>), grid, dim3(64 * W_), 0, s, n, m, sorted, temp, \
                               idx, (float *)nullptr, skip, prefix_out, xyz, prefix_cap, fold_init);                       \
    } while (0)
            switch (scene_index_np(n)) {
                case 2048: EPNET_FPS_INDEXED(4, 8); break;
                case 4096: EPNET_FPS_INDEXED(4, 16); break;
                case 8192: EPNET_FPS_INDEXED(4, 32); break;
                default:
                    if (wide) EPNET_FPS_INDEXED(16, 16);
                    else EPNET_FPS_INDEXED(8, 32);
                    break;
            }
#undef EPNET_FPS_INDEXED
            centres_done = new_xyz != nullptr && ctr_in_kernel;
        }
        rc = check_launch("furthest_point_sampling");
    }
    if (rc || !new_xyz || centres_done) return rc;
    if (m == 0) return EPNET_OK;
    EPNET_REQUIRE(xyz && b <= 65535);
    launch_gather_centres(b, n, m, xyz, idx, new_xyz, s);
    return check_launch("sample_centres gather");
}

extern "C" int epnet_furthest_point_sampling_indexed(int b, int n, int m, const float *xyz, void *index,
                                                     size_t index_bytes, float *temp, int *idx,
                                                     epnet_stream_t stream) {
    return fps_over_index(b, n, m, xyz, index, index_bytes, temp, idx, nullptr, (hipStream_t)stream);
}

// furthest point sampling from a fresh state AND the gather of the selected coordinates, i.e. the head of an SA module
// (pointnet2_modules.py:39-45: furthest_point_sample, then gather_operation on the flipped cloud, flipped back):
// idx (B,M) and new_xyz (B,M,3) = xyz[b, idx[b,i], :]. temp: running-distance scratch (B,N) or NULL (then every
// distance starts at 1e10 as pointnet2_utils.py:26 sets it; allowed for 64 <= n <= 16384); index may be NULL.
extern "C" int epnet_sample_centres(int b, int n, int m, const float *xyz, void *index, size_t index_bytes,
                                    float *temp, int *idx, float *new_xyz, epnet_stream_t stream) {
    EPNET_REQUIRE(b >= 0 && n >= 1 && m >= 0);
    if (b == 0 || m == 0) return EPNET_OK;
    EPNET_REQUIRE(xyz && idx && new_xyz);
    return fps_over_index(b, n, m, xyz, index, index_bytes, temp, idx, new_xyz, (hipStream_t)stream);
}

// epnet_sample_centres for the levels of a sampling PYRAMID (SA level l + 1 samples the centres of level l).
// Furthest point sampling is nested: as long as every round's maximum is unique, the first m' samples of a sequence are the
// furthest-point samples OF that sequence (the global maximiser over all points is itself one of the candidates; the running
// distances are the same fp32 values), so idx = 0 .. m'-1 -- bit for bit what the reference's kernel computes on the centres,
// whose tie-break only matters among equal maxima. prefix_in[b] (or NULL) = the number of leading rounds of the sampling that
// produced xyz in which the maximum was unique up to exact twins; scenes with prefix_in[b] >= m take the identity, the others run
// the rounds.
// prefix_out[b] (or NULL) receives the same knowledge about THIS sampling's output (0 where the kernel cannot tell), looked
// for during the first prefix_cap rounds only (<= 0: all rounds) -- the next level's sample count is all anybody will ask for.
extern "C" int epnet_sample_centres_chain(int b, int n, int m, const float *xyz, void *index, size_t index_bytes,
                                          float *temp, int *idx, float *new_xyz, const int *prefix_in, int *prefix_out,
                                          int prefix_cap, epnet_stream_t stream) {
    EPNET_REQUIRE(b >= 0 && n >= 1 && m >= 0);
    if (b == 0 || m == 0) return EPNET_OK;
    EPNET_REQUIRE(xyz && idx);   // new_xyz may be NULL: the indices alone (the centres come from epnet_scene_index_build_gathered)
    if (m > n) prefix_in = nullptr;  // more samples than points: the sequence repeats points, nothing is known
    return fps_over_index(b, n, m, xyz, index, index_bytes, temp, idx, new_xyz, (hipStream_t)stream, prefix_in, prefix_out,
                          prefix_cap);
}
