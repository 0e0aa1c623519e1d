// runsum.h -- scatter-add without atomics for the three gradient ops (gather / group_points / three_interpolate grads;
// reference: sampling_gpu.cu:46-63, group_points_gpu.cu:8-25, interpolate_gpu.cu:120-142, all atomicAdd scatters).
//
// grad_points[c][j] = sum over the entries t with key(t) = j of  grad_out[c][pos(t)] (* weight[t]).
//
// Two kernels, both over caller scratch:
//   pack_kernel     groups the entries of a scene by their target once (counting sort in LDS, `parts` workgroups per
//                   scene, each owning a range of targets) into 32-bit words  end << 31 | key << 16 | pos, `end` marking
//                   the last entry of a target's run; the array is padded to a multiple of 4096 with one-entry runs of a dump
//                   target and stored blocked-transposed, so that thread q of the sum kernel owns the sorted slots
//                   [q * ept, (q + 1) * ept) and still reads them with coalesced 16-byte loads.
//   scatter_kernel  one 1024-thread workgroup walks a sequence of channel-row groups of one scene. Per group: the R rows of
//                   grad_out are copied into LDS (16-byte loads, each element read from HBM once; the NEXT group's rows are
//                   requested as soon as this group's entry loads are out), every thread sums its ept entries
//                   in sorted order -- closed runs go to the LDS copy of the output row with plain stores, a run that
//                   crosses thread boundaries is finished by the thread it starts in (partial sums handed over through
//                   LDS) -- and the output rows are added to grad_points with coalesced 16-byte accesses. The entry words
//                   travel a few groups ahead of the sums in a register ring (unconditional loads: countable).
// Every thread does the same amount of work whatever the list lengths are (ball-query padding repeats an index up to
// nsample times, FPS-subset points near the sensor are nearest to hundreds of unknowns), no list is walked through a
// chain of dependent loads, and nothing is atomic. The order of the terms inside a run is whatever the counting sort
// produced (the reference's atomicAdd order is unspecified as well).
#pragma once
#include <stdlib.h>

#include "common.h"
#include "spatial.h"

namespace epnet {
namespace runsum {

#ifdef EPNET_RUNSUM_STATS  // diagnostic build only (profiles/micro/runsum_stats.py): phase cycle counters of the scatter kernel
__device__ unsigned long long g_runsum_stats[16];   // [0..7]: wave 0 of every workgroup, [8..15]: its last wave
__device__ unsigned long long g_pack_stats[8];      // pack_kernel, wave 0: count pass, scans, placing pass, padding, workgroups
#define EPNET_RS_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define EPNET_RS_ACC(slot, a, b) rs_acc[slot] += (unsigned long long)((b) - (a))
#define EPNET_RS_BEGIN unsigned long long rs_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define EPNET_RS_END                                                                                     \
    if ((threadIdx.x & 63) == 0 && (threadIdx.x == 0 || threadIdx.x == kThreads - 64))                    \
        for (int s_ = 0; s_ < 8; ++s_) atomicAdd(&g_runsum_stats[(threadIdx.x ? 8 : 0) + s_], rs_acc[s_])
#else
#define EPNET_RS_STAMP(var)
#define EPNET_RS_ACC(slot, a, b)
#define EPNET_RS_BEGIN
#define EPNET_RS_END
#endif

constexpr int kThreads = 1024;
constexpr int kChunk = kThreads * 4;      // the entry array is padded to whole 16-byte loads of the workgroup
constexpr int kMaxEntries = 65536;        // pos field: 16 bits
constexpr int kMaxRowFloats = 32768;      // 128 KB of LDS for the staged rows
constexpr int kMaxTargets = 16384;        // key field: 15 bits, one value kept for the dump target
constexpr int kLdsLimit = 158 * 1024;

__host__ __device__ inline int padded_entries(int p) { return (p + kChunk - 1) / kChunk * kChunk; }

// memory index of sorted slot s (thread s / ept holds it as element s % ept): 16-byte groups interleaved over the threads
__host__ __device__ inline int slot_to_mem(int s, int ept) {
    const int q = s / ept, i = s - q * ept;
    return (((i >> 2) * kThreads) + q) * 4 + (i & 3);
}

// Rows longer than the LDS budget are worked off in TILES of positions (a launch pair per tile, each adding to grad_points):
// the longest tile whose single row fits beside the output row of n targets and the hand-over slots.
inline int tile_floats(int n, int div, long long row_floats) {
    const long long n_pad = (n + 1 + 3) / 4 * 4;
    long long t = ((long long)kLdsLimit / 4 - n_pad - 2 * kThreads);
    if (t > kMaxRowFloats) t = kMaxRowFloats;
    if (t * div > kMaxEntries) t = kMaxEntries / div;
    if (const char *e = getenv("EPNET_RUNSUM_TILE")) {  // tuning: shorter tiles, more workgroups per CU
        const long long cap = atoll(e);
        if (cap >= kChunk && cap < t) t = cap;
    }
    if (t >= row_floats) return (int)row_floats;   // one tile
    return (int)(t / kChunk * kChunk);             // whole chunks of 4096 positions (0: does not fit)
}

inline bool usable(int n, int div, long long entries, long long row_floats) {
    return n >= 1 && n <= kMaxTargets && entries >= 1 && entries == row_floats * div && row_floats >= 4 && row_floats % 4 == 0 &&
           row_floats <= (1 << 22) && tile_floats(n, div, row_floats) >= 4;
}

inline size_t lds_bytes(int rows, int n, int row_floats) {
    const int n_pad = (n + 1 + 3) / 4 * 4;
    return ((size_t)rows * row_floats + (size_t)rows * n_pad + (size_t)2 * rows * kThreads) * sizeof(float);
}

// rows per pass: as many as fit (the entry words are read once per pass), but not so many that the chip runs short of
// workgroups
inline int pick_rows(int b, int c, int n, int row_floats) {
    int rows = 8;
    while (rows > 1 && ((long long)rows * row_floats > kMaxRowFloats || lds_bytes(rows, n, row_floats) > (size_t)kLdsLimit)) rows >>= 1;
    while (rows > 1 && (long long)b * div_up(c, rows) < 256) rows >>= 1;
    return rows;
}

// (n, div, row_floats as for usable(): the scratch holds the sorted entries of ONE tile of every scene)
inline size_t workspace_bytes(int b, int n, int div, long long row_floats, bool weighted) {
    const long long tile = tile_floats(n, div, row_floats);
    return (size_t)b * (size_t)padded_entries((int)(tile * div)) * (weighted ? 8 : 4);
}

// ---- the inverse index -------------------------------------------------------------------------------------------
// grid (parts, b). idx: p targets per scene, clamped into [0, n) (an index outside is undefined behaviour in the reference;
// here it cannot leave the scene). pos of entry t = t / div (div = 3: the three neighbours of an unknown point).
template <bool W>
__global__ __launch_bounds__(kThreads) void pack_kernel(int n, int p, int div, size_t idx_stride, const int *__restrict__ idx,
                                                        const float *__restrict__ weight, unsigned *__restrict__ ent,
                                                        float *__restrict__ wsorted) {
    extern __shared__ int s_bins[];  // counts (nb), then run starts (nb + 1), then a dummy counter per thread
    __shared__ int s_part[kThreads / 64];
    __shared__ int s_below[kThreads / 64];
    const int q = threadIdx.x, lane = q & 63, wave = q >> 6;
    const int parts = gridDim.x, part = blockIdx.x, bs = blockIdx.y;
    const int P = padded_entries(p), ept = P / kThreads;
    idx += (size_t)bs * idx_stride;   // p entries of this tile; the scenes are idx_stride entries apart
    ent += (size_t)bs * P;
    if (W) {
        weight += (size_t)bs * idx_stride;
        wsorted += (size_t)bs * P;
    }
    const int nb = (n + parts - 1) / parts;
    const int j0 = part * nb, j1 = min(n, j0 + nb);
    int *s_cnt = s_bins, *s_start = s_bins + nb;
    for (int i = q; i < nb; i += kThreads) s_cnt[i] = 0;
    __syncthreads();
    EPNET_RS_STAMP(p_a);
    // (both passes over the entries: kUnroll targets are requested together and their LDS atomics issued back to back -- one at a
    // time every iteration is a chain of an L2 load, a returning LDS atomic and a dependent LDS read. The unrolled body has no
    // bounds checks at all -- the compiler sinks a load whose value is only used under `t < p` into that branch and waits for
    // it on the spot -- and the last partial block goes one entry at a time)
    constexpr int kUnroll = 8;
    int below = 0;
    auto count = [&](int j) {
        if (j < j0) ++below;
        else if (j < j1) atomicAdd(&s_cnt[j - j0], 1);
    };
    int t0 = q;
    for (; t0 + (kUnroll - 1) * kThreads < p; t0 += kThreads * kUnroll) {
        int js[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) js[u] = min(max(idx[t0 + u * kThreads], 0), n - 1);
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) count(js[u]);
    }
    for (int t = t0; t < p; t += kThreads) count(min(max(idx[t], 0), n - 1));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) below += __shfl_xor(below, off, 64);
    if (lane == 0) s_below[wave] = below;
    __syncthreads();
    EPNET_RS_STAMP(p_b);
    const int per = (nb + kThreads - 1) / kThreads;
    int sum = 0;
    for (int i = 0; i < per; ++i) {
        const int k = q * per + i;
        if (k < nb) sum += s_cnt[k];
    }
    const int incl = wave_inclusive_scan(sum);
    if (lane == 63) s_part[wave] = incl;
    __syncthreads();
    int base = incl - sum;
    for (int w = 0; w < wave; ++w) base += s_part[w];
    int before = 0;
    for (int w = 0; w < kThreads / 64; ++w) before += s_below[w];
    for (int i = 0; i < per; ++i) {
        const int k = q * per + i;
        if (k < nb) {
            s_start[k] = before + base;
            base += s_cnt[k];
        }
    }
    if (q == 0) {  // end of the last run of the range
        int total = before;
        for (int w = 0; w < kThreads / 64; ++w) total += s_part[w];
        s_start[nb] = total;
    }
    __syncthreads();
    EPNET_RS_STAMP(p_c);
    const unsigned div_magic = div == 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)div + 1ull);   // (div == 1: see below)
    auto place = [&](int j, int t, int old) {   // the run is filled from its end: old = entries still to place
        const int slot = s_start[j - j0] + old - 1;
        const unsigned end = (slot + 1 == s_start[j - j0 + 1]) ? 0x80000000u : 0u;
        const int at = slot_to_mem(slot, ept);
        ent[at] = end | ((unsigned)j << 16) | (div == 1 ? (unsigned)t : __umulhi((unsigned)t, div_magic));   // == t / div (t * div < 2^32)
        if (W) wsorted[at] = weight[t];   // (only the workgroup that owns the target reads the weight)
    };
    // unrolled: every step of the kUnroll entries is issued for all of them before any result is used, with no branch in between
    // (an entry of another workgroup's range goes through the motions on the thread's dummy counter): loads, returning atomics, run
    // bounds and weights are one wait each instead of kUnroll chains of four waits
    int *s_dummy = s_bins + 2 * nb + 1 + q;   // (a word per thread: one shared dummy would serialise the atomics of a wave)
    t0 = q;
    for (; t0 + (kUnroll - 1) * kThreads < p; t0 += kThreads * kUnroll) {
        int js[kUnroll], olds[kUnroll], starts[kUnroll], nexts[kUnroll];
        float ws[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            js[u] = min(max(idx[t0 + u * kThreads], 0), n - 1);
            ws[u] = W ? weight[t0 + u * kThreads] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const bool mine = js[u] >= j0 && js[u] < j1;
            js[u] = mine ? js[u] - j0 : -1;
            olds[u] = atomicSub(mine ? &s_cnt[js[u]] : s_dummy, 1);
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            starts[u] = s_start[max(js[u], 0)];
            nexts[u] = s_start[max(js[u], 0) + 1];
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            if (js[u] < 0) continue;
            const int slot = starts[u] + olds[u] - 1;
            const unsigned end = (slot + 1 == nexts[u]) ? 0x80000000u : 0u;
            const int at = slot_to_mem(slot, ept);
            const unsigned t = (unsigned)(t0 + u * kThreads);
            ent[at] = end | ((unsigned)(js[u] + j0) << 16) | (div == 1 ? t : __umulhi(t, div_magic));
            if (W) wsorted[at] = ws[u];
        }
    }
    for (int t = t0; t < p; t += kThreads) {
        const int j = min(max(idx[t], 0), n - 1);
        if (j >= j0 && j < j1) place(j, t, atomicSub(&s_cnt[j - j0], 1));
    }
    EPNET_RS_STAMP(p_d);
#ifdef EPNET_RUNSUM_STATS
    if (q == 0) {
        atomicAdd(&g_pack_stats[0], p_b - p_a); atomicAdd(&g_pack_stats[1], p_c - p_b); atomicAdd(&g_pack_stats[2], p_d - p_c);
        atomicAdd(&g_pack_stats[4], 1ull);
    }
#endif
    if (part == parts - 1)  // padding: entries of a dump target (key n) that read position 0 with weight 0, every one a run
        for (int s = p + q; s < P; s += kThreads) {  // of its own (ONE long run would be handed from thread to thread)
            const int at = slot_to_mem(s, ept);
            ent[at] = 0x80000000u | ((unsigned)n << 16);
            if (W) wsorted[at] = 0.f;
        }
}

// ---- the sums ----------------------------------------------------------------------------------------------------
// grid (workgroups per scene, b); workgroup x takes the row groups [x * per_wg, (x + 1) * per_wg) of its scene.
// grad_out rows of a scene: row_floats floats at grad_out + bs * gstride + row * row_stride (row_stride > row_floats: one tile
// of longer rows); grad_points rows: ((bs * c) + row) * n.
template <int R, bool W, int kPre = 8>   // kPre: 16-byte loads per thread that bring in the R rows (R * row_floats <= kPre * 4096)
__global__ __launch_bounds__(kThreads) void scatter_kernel(int c, int n, int row_floats, int row_stride, int P, int per_wg, int vec_out,
                                                           size_t gstride,
                                                           const float *__restrict__ grad_out,
                                                           const unsigned *__restrict__ ent,
                                                           const float *__restrict__ wsorted,
                                                           float *__restrict__ grad_points) {
    extern __shared__ float s_mem[];
    __shared__ unsigned char s_whole[kThreads];
    const int q = threadIdx.x;
    // Workgroups are dealt round-robin over the 8 XCDs (linear id % 8), each with its own L2. Every row pass of a workgroup re-reads
    // the scene's sorted entries (up to 0.4 MB): the workgroups of a scene therefore go to ONE XCD (a bijective relabelling of the
    // linear id; placement is a matter of speed only), where the entries of its two or three scenes stay in L2 instead of all
    // scenes' entries passing through every L2.
    int wg_x, bs;
    {
        const int nwg = (int)(gridDim.x * gridDim.y), orig = (int)(blockIdx.y * gridDim.x + blockIdx.x);
        const int xcd = orig & 7, qq = nwg >> 3, rr = nwg & 7;
        const int wgid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (orig >> 3);
        bs = wgid / (int)gridDim.x;
        wg_x = wgid - bs * (int)gridDim.x;
    }
    const int n_pad = (n + 1 + 3) / 4 * 4;
    float *s_row = s_mem;                           // R rows of grad_out
    float *s_out = s_row + (size_t)R * row_floats;  // R output rows (+ the dump slot n)
    float *s_first = s_out + (size_t)R * n_pad;     // partial sum of the run a thread's range starts inside of
    float *s_dump = s_first + (size_t)R * kThreads; // a private word per thread and row: where the running sums that nobody reads go
    const int ept = P / kThreads, nq = ept >> 2;
    const uint4 *ent4 = reinterpret_cast<const uint4 *>(ent + (size_t)bs * P);
    const float4 *w4p = W ? reinterpret_cast<const float4 *>(wsorted + (size_t)bs * P) : nullptr;
    const int groups = (c + R - 1) / R;
    const int g_beg = wg_x * per_wg, g_end = min(groups, g_beg + per_wg);
    if (g_beg >= g_end) return;

    // what a thread needs to know about its slice of the sorted entries, the same for every row
    bool open_start = false;
    if (q > 0) open_start = !(ent[(size_t)bs * P + slot_to_mem(q * ept - 1, ept)] >> 31);
    bool any_end = false;
    unsigned last = 0;
    for (int g = 0; g < nq; ++g) {
        const uint4 e = ent4[g * kThreads + q];
        any_end = any_end || ((e.x | e.y | e.z | e.w) >> 31);
        last = e.w;
    }
    const bool open_end = !(last >> 31);
    const bool whole = open_start && !any_end;  // one link of a run that began before and ends after this thread's range
    const bool head = open_end && !whole;       // the run this thread's range ends inside of begins here: finish it
    const int last_key = (int)((last >> 16) & 0x7FFFu);
    const int first_key = q > 0 ? (int)((ent[(size_t)bs * P + slot_to_mem(q * ept - 1, ept)] >> 16) & 0x7FFFu) : 0;
    s_whole[q] = whole ? 1 : 0;
    for (int i = q; i < R * n_pad; i += kThreads) s_out[i] = 0.f;
    __syncthreads();
    // A run that crosses MANY threads (a target with thousands of entries) must not be collected link by link: the thread a
    // run starts in collects at most kNear links; links further down the chain add themselves with an LDS atomic.
    constexpr int kNear = 4;
    bool long_chain = false;   // (head) more than kNear links follow
    if (head) {
        long_chain = true;
        for (int k = q + 1; k <= q + kNear && k < kThreads; ++k)
            if (!s_whole[k]) {
                long_chain = false;
                break;
            }
        if (q + kNear >= kThreads) long_chain = false;  // the chain ends with the last thread at the latest
    }
    bool far = false;          // (open_start) more than kNear links away from the thread the run starts in
    if (open_start) {
        int d = 1;
        for (int k = q - 1; k >= 0 && s_whole[k] && d <= kNear; --k) ++d;
        far = d > kNear;
    }

    const float *go = grad_out + (size_t)bs * gstride;
    float *gp = grad_points + (size_t)bs * c * n;
    const int row4 = row_floats >> 2;
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 pre[kPre];
    auto fetch = [&](int g) {
        const int nr = min(R, c - g * R);
        const float *src = go + (size_t)g * R * row_stride;
        const int total4 = nr * row4;
#pragma unroll
        for (int k = 0; k < kPre; ++k) {
            const int e = k * kThreads + q;
            if (e < total4) {
                const int r = (R == 1 || row_stride == row_floats) ? 0 : e / row4;
                const int col = e - r * row4;
                pre[k] = (R == 1 || row_stride == row_floats) ? reinterpret_cast<const f4 *>(src)[e]
                                                              : reinterpret_cast<const f4 *>(src + (size_t)r * row_stride)[col];
            }
        }
    };
    constexpr int kAhead = R == 1 ? 3 : (R == 2 ? 2 : 1);  // (the wider variants have no registers to spare)
    uint4 ering[kAhead];
    float4 wring[W ? kAhead : 1];
    auto load_group = [&](int slot, int gi) {
        ering[slot] = ent4[gi * kThreads + q];
        if (W) wring[slot] = w4p[gi * kThreads + q];
    };
#pragma unroll
    for (int j = 0; j < kAhead; ++j) load_group(j, min(j, nq - 1));
    fetch(g_beg);
    EPNET_RS_BEGIN;
    for (int g = g_beg; g < g_end; ++g) {
        const int c0 = g * R, nr = min(R, c - c0);
        EPNET_RS_STAMP(t_a);
        {
            const int total4 = nr * row4;
            f4 *dst = reinterpret_cast<f4 *>(s_row);
#pragma unroll
            for (int k = 0; k < kPre; ++k) {
                const int e = k * kThreads + q;
                if (e < total4) dst[e] = pre[k];
            }
        }
        __syncthreads();
        EPNET_RS_STAMP(t_b);
        // the rows of grad_points this thread will add to, requested before the sums so that the read-modify-write at the
        // end does not wait for them
        const int n4 = n >> 2;
        const bool have_old = vec_out && q < nr * n4;
        float4 oldv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (have_old) oldv = reinterpret_cast<const float4 *>(gp + (size_t)(c0 + q / n4) * n)[q - (q / n4) * n4];
        // Branch-free walk over the thread's entries: EVERY entry stores the running sum of its run -- the last entry of a run
        // to the output row slot of its target, an entry
        // inside the run the thread's range began in to the hand-over slot, every other entry to the thread's private dump
        // word. (Neighbouring lanes hold neighbouring targets, a few words apart: with every running sum going to its target's
        // slot the stores of a wave met in a quarter of the banks, 4 to 8 lanes deep, and set the pace of the whole kernel; the
        // private words of a wave are consecutive.) A run's slot in s_out is written by one thread only: the one it starts in.
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.f;
        bool first_open = open_start;
        // where a running sum goes that is not a finished run's: the hand-over slot while inside the run the range began in, the
        // thread's dump word afterwards (kept as one value: a select per entry instead of two)
        const int dump_slot = (int)(s_dump - s_out) + q;
        int priv = open_start ? (int)(s_first - s_out) + q : dump_slot;
        // the entry words of kAhead groups of four are in flight ahead of the sums (they come from L2 at best). The ring's slots
        // are static (the loop is unrolled by its length) and every load is UNCONDITIONAL (past the end it repeats the last
        // group): only then can the compiler count the loads in flight and wait for exactly the group it needs -- with a
        // conditional load it falls back to `s_waitcnt vmcnt(0)` at every use, i.e. no prefetch at all
        auto sum_group = [&](const uint4 &e4, const float4 &w4) __attribute__((always_inline)) {
            const unsigned es[4] = {e4.x, e4.y, e4.z, e4.w};
            const float ws[4] = {w4.x, w4.y, w4.z, w4.w};
            float v[4][R];  // all reads of the group before any store (the stores go to the same LDS array: the compiler
                            // would otherwise keep every read behind the previous entry's store)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int r = 0; r < R; ++r) v[k][r] = s_row[r * row_floats + (int)(es[k] & 0xFFFFu)];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned e = es[k];
                const int key = (int)((e >> 16) & 0x7FFFu);
                const bool end = (int)e < 0;
                const bool real = !first_open && end;
                const int slot = real ? key : priv;   // index relative to s_out, row 0
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    acc[r] += W ? v[k][r] * ws[k] : v[k][r];
                    s_out[(real ? r * n_pad : r * kThreads) + slot] = acc[r];
                    acc[r] = end ? 0.f : acc[r];
                }
                priv = end ? dump_slot : priv;   // (the hand-over slot only up to the end of the run the range began in)
                first_open = first_open && !end;
            }
        };
        const float4 ones = make_float4(1.f, 1.f, 1.f, 1.f);
        int g0 = 0;
        for (; g0 + kAhead <= nq; g0 += kAhead) {
#pragma unroll
            for (int j = 0; j < kAhead; ++j) {
                sum_group(ering[j], W ? wring[j] : ones);              // (summed out of the ring's registers, refilled
                load_group(j, min(g0 + j + kAhead, nq - 1));           //  afterwards: no copies, no rotation)
            }
        }
#pragma unroll
        for (int j = 0; j < kAhead - 1; ++j)   // the nq % kAhead groups left over (their words are in the ring's first slots)
            if (g0 + j < nq) sum_group(ering[j], W ? wring[j] : ones);
        // the run this thread's range ends inside of (and begins in) has no last entry here: its running sum goes to its target's
        // slot now, where the hand-over step below completes it
        if (head) {
#pragma unroll
            for (int r = 0; r < R; ++r) s_out[r * n_pad + last_key] = acc[r];
        }
        // the first groups' words again for the next pass (the same words every pass), ahead of the rows
#pragma unroll
        for (int j = 0; j < kAhead; ++j) load_group(j, min(j, nq - 1));
        // The next group's rows are requested HERE, behind the last entry loads of this pass: vector loads return in order, so a
        // request issued ahead of the entry loop would have every wait for entry words wait for the HBM rows as well (the compiler
        // emitted `s_waitcnt vmcnt(0)` at the head of the loop: no overlap at all). Now they travel during the hand-over steps,
        // the output rows and the next copy into LDS. (The old output values were requested before the loop: pinned here so
        // that nothing issued before the prefetch is waited for after it.)
        if (have_old) asm volatile("" ::"v"(oldv.x), "v"(oldv.y), "v"(oldv.z), "v"(oldv.w));
        if (g + 1 < g_end) fetch(g + 1);
        EPNET_RS_STAMP(t_c);
        __syncthreads();
        EPNET_RS_STAMP(t_d);
        if (head) {  // collect the (near) links of the run this thread's range ends inside of
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float links = 0.f;
                for (int k = q + 1; k <= q + kNear && k < kThreads; ++k) {
                    links += s_first[r * kThreads + k];
                    if (!s_whole[k]) break;
                }
                if (long_chain) atomicAdd(&s_out[r * n_pad + last_key], links);   // the slot holds this thread's own part already
                else s_out[r * n_pad + last_key] = acc[r] + links;
            }
        }
        if (far) {
#pragma unroll
            for (int r = 0; r < R; ++r) atomicAdd(&s_out[r * n_pad + first_key], s_first[r * kThreads + q]);
        }
        __syncthreads();
        EPNET_RS_STAMP(t_e);
        // grad_points += the output rows; the LDS copies go back to zero for the next group
        if (vec_out) {
            for (int i = q; i < nr * n4; i += kThreads) {
                const int r = i / n4, j4 = i - r * n4;
                float4 *so = reinterpret_cast<float4 *>(s_out + r * n_pad) + j4;
                float4 *dst = reinterpret_cast<float4 *>(gp + (size_t)(c0 + r) * n) + j4;
                const float4 a = *so, o = (i == q && have_old) ? oldv : *dst;
                *dst = make_float4(o.x + a.x, o.y + a.y, o.z + a.z, o.w + a.w);
                *so = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            for (int i = q; i < nr * n; i += kThreads) {
                const int r = i / n, j = i - r * n;
                gp[(size_t)(c0 + r) * n + j] += s_out[r * n_pad + j];
                s_out[r * n_pad + j] = 0.f;
            }
        }
        EPNET_RS_STAMP(t_f);
        EPNET_RS_ACC(0, t_a, t_b); EPNET_RS_ACC(1, t_b, t_c); EPNET_RS_ACC(2, t_c, t_d); EPNET_RS_ACC(3, t_d, t_e); EPNET_RS_ACC(4, t_e, t_f);
        EPNET_RS_ACC(5, t_a, t_a + 1);
        // (the next group's rows overwrite s_row only after every thread has passed the barrier before the head step, and
        // its sums start only after the barrier that follows the copy: the zeroing above is complete by then)
    }
    EPNET_RS_END;
}

template <bool W>
inline int launch(int b, int c, int n, int div, int row_floats, const float *grad_out, size_t gstride, const int *idx,
                  const float *weight, float *grad_points, void *workspace, size_t workspace_bytes_given, hipStream_t s,
                  const char *what) {
    if (workspace_bytes_given < workspace_bytes(b, n, div, row_floats, W)) return EPNET_ENOMEM;
    if (b > 65535) return EPNET_ELIMIT;
    if (((uintptr_t)grad_out | (uintptr_t)(gstride * sizeof(float))) & 15) return EPNET_EINVAL;  // callers check alignment first
    const int tile = tile_floats(n, div, row_floats);
    const size_t idx_stride = (size_t)row_floats * div;
    // parts: enough workgroups to spread the counting sort over the chip, each with at least a few hundred targets
    int parts = b >= 128 ? 2 : b >= 32 ? 8 : 16;
    if (const char *e = getenv("EPNET_RUNSUM_PARTS")) parts = atoi(e) > 0 ? atoi(e) : parts;  // (tuning)
    while (parts > 1 && n / parts < 256) parts >>= 1;
    const int nb = div_up(n, parts);
    const int vec_out = ((n & 3) == 0 && ((uintptr_t)grad_points & 15) == 0) ? 1 : 0;
    for (int pos0 = 0; pos0 < row_floats; pos0 += tile) {
        const int len = min(tile, row_floats - pos0);  // a multiple of 4: row_floats and the tile length are
        const int entries = len * div;
        const int P = padded_entries(entries);
        unsigned *ent = (unsigned *)workspace;
        float *wsorted = W ? (float *)(ent + (size_t)b * P) : nullptr;
        hipLaunchKernelGGL((pack_kernel<W>), dim3(parts, b), dim3(kThreads), (size_t)(2 * nb + 1 + kThreads) * sizeof(int), s, n, entries, div,
                           idx_stride, idx + (size_t)pos0 * div, W ? weight + (size_t)pos0 * div : nullptr, ent, wsorted);
        int rc = check_launch("inverse index");
        if (rc) return rc;
        const int rows = pick_rows(b, c, n, len);
        const int groups = div_up(c, rows);
        // about two workgroups per CU over the whole launch (one resident at a time when a row fills the LDS), whole passes
        const size_t lds = lds_bytes(rows, n, len);
        int wgs = (lds > 80 * 1024 ? 256 : 512) / b;
        if (wgs < 1) wgs = 1;
        if (wgs > groups) wgs = groups;
        const int per_wg = div_up(groups, wgs);
        wgs = div_up(groups, per_wg);
        dim3 grid(wgs, b);
#define EPNET_RUNSUM(R_, PRE_)                                                                                                 \
    hipLaunchKernelGGL((scatter_kernel<R_, W, PRE_>), grid, dim3(kThreads), lds, s, c, n, len, row_floats, P, per_wg, vec_out, gstride, \
                       grad_out + pos0, ent, wsorted, grad_points)
        switch (rows) {
            case 8: EPNET_RUNSUM(8, 8); break;
            case 4: EPNET_RUNSUM(4, 8); break;
            case 2: EPNET_RUNSUM(2, 8); break;
            default:
                if (len <= 16384) EPNET_RUNSUM(1, 4);   // (frees the registers of the entry-word ring)
                else EPNET_RUNSUM(1, 8);
                break;
        }
#undef EPNET_RUNSUM
        rc = check_launch(what);
        if (rc) return rc;
    }
    return EPNET_OK;
}

}  // namespace runsum
}  // namespace epnet
