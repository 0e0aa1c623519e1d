// interpolate.hip -- three_nn, three_interpolate and its gradient for gfx950.
//
// Replaces three_nn_kernel_fast, three_interpolate_kernel_fast and three_interpolate_grad_kernel_fast
// (pointnet2_lib/pointnet2/src/interpolate_gpu.cu:9-52, 77-97, 120-142).
//
// three_nn: the reference gives each unknown point to one thread that scans all m known points
// (n/64 waves in flight -- one per CU at n = 16384). Here kSplit = 8 lanes share one unknown,
// each scans every 8th known point out of an LDS tile and keeps its own 3 best; the 8 partial
// lists are merged with xor-shuffles. The reference's strict '<' insertion in index order
// (:37-48) yields exactly the 3 smallest pairs under the lexicographic order (d, k) -- earlier
// index wins a distance tie -- so merging partial lists under that same order is bit-identical,
// whatever the split. The reference's double accumulators initialised to 1e40 (:30) behave like
// float +inf (a float d is never >= 1e40 unless it is inf/nan, and (float)1e40 == inf).
//
// three_interpolate: out = w0*p[i0] + w1*p[i1] + w2*p[i2], left to right, no contraction (:96).
// A thread owns 4 consecutive unknowns, keeps their 12 indices and weights in registers and
// loops over a chunk of channels (the reference re-reads idx and weight for every channel).
#include <limits.h>
#include <math.h>
#include <stdlib.h>

#include "common.h"
#include "dpp.h"
#include "spatial.h"
#include "runsum.h"

namespace epnet {

constexpr int kNnThreads = 256;
constexpr int kNnSplit = 8;                       // lanes per unknown
constexpr int kNnPerBlock = kNnThreads / kNnSplit;  // 32 unknowns per workgroup
constexpr int kNnTile = 4096;                     // known points per LDS tile (48 KiB)

struct Top3 {
    float d0, d1, d2;
    int i0, i1, i2;
};

__device__ __forceinline__ bool pair_less(float da, int ia, float db, int ib) {
    return da < db || (da == db && ia < ib);
}

// insert (d,k) into a list sorted by (d,k); used for merging (k may be smaller than entries)
__device__ __forceinline__ void top3_insert(Top3 &t, float d, int k) {
    if (pair_less(d, k, t.d0, t.i0)) {
        t.d2 = t.d1; t.i2 = t.i1;
        t.d1 = t.d0; t.i1 = t.i0;
        t.d0 = d; t.i0 = k;
    } else if (pair_less(d, k, t.d1, t.i1)) {
        t.d2 = t.d1; t.i2 = t.i1;
        t.d1 = d; t.i1 = k;
    } else if (pair_less(d, k, t.d2, t.i2)) {
        t.d2 = d; t.i2 = k;
    }
}

__global__ __launch_bounds__(kNnThreads) void three_nn_kernel(int n, int m, const float *__restrict__ unknown,
                                                              const float *__restrict__ known,
                                                              float *__restrict__ dist2, int *__restrict__ idx) {
    __shared__ float tile[kNnTile * 3];
    const int bs = blockIdx.y;
    unknown += (size_t)bs * n * 3;
    known += (size_t)bs * m * 3;
    dist2 += (size_t)bs * n * 3;
    idx += (size_t)bs * n * 3;

    const int part = threadIdx.x & (kNnSplit - 1);
    const int pt = blockIdx.x * kNnPerBlock + (threadIdx.x / kNnSplit);
    const bool ok = pt < n;
    const float ux = ok ? unknown[pt * 3 + 0] : 0.f;
    const float uy = ok ? unknown[pt * 3 + 1] : 0.f;
    const float uz = ok ? unknown[pt * 3 + 2] : 0.f;

    const float inf = __builtin_huge_valf();
    const int big = 0x7fffffff;
    Top3 t = {inf, inf, inf, big, big, big};

    for (int t0 = 0; t0 < m; t0 += kNnTile) {
        const int tn = min(kNnTile, m - t0);
        __syncthreads();
        for (int e = threadIdx.x; e < tn * 3; e += kNnThreads) tile[e] = known[(size_t)t0 * 3 + e];
        __syncthreads();
        for (int k = part; k < tn; k += kNnSplit) {
            const float x = tile[k * 3 + 0], y = tile[k * 3 + 1], z = tile[k * 3 + 2];
            const float dx = ux - x, dy = uy - y, dz = uz - z;
            const float d = dx * dx + dy * dy + dz * dz;
            // ascending k within a lane: strict '<' on d alone is the (d,k) order here
            if (d < t.d0) {
                t.d2 = t.d1; t.i2 = t.i1;
                t.d1 = t.d0; t.i1 = t.i0;
                t.d0 = d; t.i0 = t0 + k;
            } else if (d < t.d1) {
                t.d2 = t.d1; t.i2 = t.i1;
                t.d1 = d; t.i1 = t0 + k;
            } else if (d < t.d2) {
                t.d2 = d; t.i2 = t0 + k;
            }
        }
    }

    // merge the kNnSplit partial lists (lanes part = 0..7 are adjacent)
#pragma unroll
    for (int off = 1; off < kNnSplit; off <<= 1) {
        const float od0 = __shfl_xor(t.d0, off, 64), od1 = __shfl_xor(t.d1, off, 64), od2 = __shfl_xor(t.d2, off, 64);
        const int oi0 = __shfl_xor(t.i0, off, 64), oi1 = __shfl_xor(t.i1, off, 64), oi2 = __shfl_xor(t.i2, off, 64);
        top3_insert(t, od0, oi0);
        top3_insert(t, od1, oi1);
        top3_insert(t, od2, oi2);
    }

    if (ok && part == 0) {
        dist2[pt * 3 + 0] = t.d0;
        dist2[pt * 3 + 1] = t.d1;
        dist2[pt * 3 + 2] = t.d2;
        // slots never filled keep the reference's initial index 0 (:31)
        idx[pt * 3 + 0] = t.i0 == big ? 0 : t.i0;
        idx[pt * 3 + 1] = t.i1 == big ? 0 : t.i1;
        idx[pt * 3 + 2] = t.i2 == big ? 0 : t.i2;
    }
}

// ---- indexed three_nn -------------------------------------------------------------------------------
// With caller scratch the known points are cell-sorted into buckets of 64 with boxes (the ball-query
// index, spatial.h). One wave per unknown point u:
//   1. lane b evaluates the lower bound L_b = |clamp(u, box_b) - u|^2 of bucket b (same fp32 expression as
//      the point distance, so d >= L_b for every point of the bucket);
//   2. the bucket with the smallest bound is scanned; the 3rd smallest of its four row minima bounds the
//      true third-nearest distance from above (four distinct points);
//   3. every bucket with L_b <= that bound is scanned; each lane keeps the three best (d, k) pairs it saw;
//   4. three rounds of a 64-bit wave minimum over (bits(d) << 32 | k) pop the global three best.
// A bucket that is not scanned has L_b > bound >= d3, so it holds neither a closer point nor an equal one:
// the result is the reference's (strict '<' in index order == lexicographic (d, k)), bit for bit.
constexpr int kNnIxThreads = 256;
constexpr int kNnIxMinKnown = 512;

__global__ __launch_bounds__(kNnIxThreads) void three_nn_indexed_kernel(int n, int m, int np,
                                                                        const float *__restrict__ unknown,
                                                                        const float4 *__restrict__ sorted,
                                                                        const float *__restrict__ boxes,
                                                                        float *__restrict__ dist2, int *__restrict__ idx) {
    const int lane = threadIdx.x & 63;
    const int bs = blockIdx.y;
    const int pt = blockIdx.x * (kNnIxThreads / 64) + (threadIdx.x >> 6);
    if (pt >= n) return;  // wave-uniform
    sorted += (size_t)bs * np;
    boxes += (size_t)bs * (np / 64) * 6;
    const float *u = unknown + ((size_t)bs * n + pt) * 3;
    const float ux = u[0], uy = u[1], uz = u[2];
    const int nb = np >> 6;
    constexpr unsigned kNone = 0xFFFFFFFFu;  // "no point": above every finite distance's bit pattern

    // per-lane three best (bits(d), k) pairs; d >= +0, so the bit pattern orders like the float
    unsigned d0 = kNone, d1 = kNone, d2 = kNone;
    int i0 = 0x7fffffff, i1 = 0x7fffffff, i2 = 0x7fffffff;
    auto offer = [&](const float4 &p) -> unsigned {
        const float dx = ux - p.x, dy = uy - p.y, dz = uz - p.z;
        const float d = dx * dx + dy * dy + dz * dz;
        const int k = __float_as_int(p.w);
        // padding rows (k < 0) and non-finite distances never enter a list, as in the reference (:37-48)
        const unsigned db = (k >= 0 && d < __builtin_huge_valf()) ? __float_as_uint(d) : kNone;
        const bool lt0 = db < d0 || (db == d0 && k < i0), lt1 = db < d1 || (db == d1 && k < i1),
                   lt2 = db < d2 || (db == d2 && k < i2);
        if (db != kNone) {
            d2 = lt1 ? d1 : (lt2 ? db : d2);  i2 = lt1 ? i1 : (lt2 ? k : i2);
            d1 = lt0 ? d0 : (lt1 ? db : d1);  i1 = lt0 ? i0 : (lt1 ? k : i1);
            d0 = lt0 ? db : d0;               i0 = lt0 ? k : i0;
        }
        return db;
    };
    auto visit = [&](int b) -> unsigned { return offer(sorted[(b << 6) + lane]); };

    unsigned bound = kNone;
    for (int b0 = 0; b0 < nb; b0 += 64) {  // nb <= 64 for m <= 4096: one pass
        const int b = b0 + lane;
        unsigned L = kNone;
        if (b < nb) {
            const float *bx = boxes + b * 6;
            const float px = __builtin_amdgcn_fmed3f(ux, bx[0], bx[1]), py = __builtin_amdgcn_fmed3f(uy, bx[2], bx[3]),
                        pz = __builtin_amdgcn_fmed3f(uz, bx[4], bx[5]);
            const float dx = ux - px, dy = uy - py, dz = uz - pz;
            const float Lf = dx * dx + dy * dy + dz * dz;
            L = Lf < __builtin_huge_valf() ? __float_as_uint(Lf) : kNone;  // all-padding buckets sit at 3e38
        }
        unsigned long long done = 0ull;
        if (b0 == 0) {
            // seed with the bucket whose box is nearest; its exact third-smallest distance bounds d3 from above
            const unsigned mn = wave_min_all(L);
            const int bstart = (int)__builtin_ctzll(__ballot(L == mn));
            unsigned db = visit(bstart);
            done = 1ull << bstart;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const unsigned mnr = wave_min_all(db);
                bound = mnr;
                const unsigned long long at = __ballot(db == mnr);
                if (lane == (int)__builtin_ctzll(at)) db = kNone;  // drop one holder of the minimum
            }
        }
        unsigned long long cand = __ballot(b < nb && L <= bound) & ~done;
        while (cand) {  // the rows of up to 4 buckets are requested together (a chain of dependent loads otherwise)
            float4 p[4];
            int nbk = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (cand) {
                    p[k] = sorted[((b0 + (int)__builtin_ctzll(cand)) << 6) + lane];
                    cand &= cand - 1ull;
                    nbk = k + 1;
                }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < nbk) offer(p[k]);
        }
    }
    // pop the wave's three smallest (d, k) pairs: min distance, then min index among its holders
    unsigned bd[3];
    int bi[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const unsigned md = wave_min_all(d0);
        const unsigned mk = wave_min_all(d0 == md ? (unsigned)i0 : kNone);
        bd[r] = md;
        bi[r] = (int)mk;
        if (d0 == md && (unsigned)i0 == mk && md != kNone) {  // exactly one lane: indices are unique
            d0 = d1; i0 = i1;
            d1 = d2; i1 = i2;
            d2 = kNone; i2 = 0x7fffffff;
        }
    }
    if (lane == 0) {
        float *dd = dist2 + ((size_t)bs * n + pt) * 3;
        int *ii = idx + ((size_t)bs * n + pt) * 3;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const bool have = bd[r] != kNone;
            dd[r] = have ? __uint_as_float(bd[r]) : __builtin_huge_valf();  // unfilled slot: (float)1e40 = inf, index 0
            ii[r] = have ? bi[r] : 0;
        }
    }
}

// ---- three_nn with BOTH point sets indexed -----------------------------------------------------------
// One wave per BUCKET of 64 spatially adjacent unknown points (one unknown per lane, taken from the unknown set's
// own scene index), so the lanes of a wave want almost the same known points:
//   1. the known buckets are visited in order of increasing box-to-box gap to the unknown bucket (the nearest one
//      gives every lane three candidates and with them an upper bound d3 on its third-nearest distance);
//   2. for a known bucket K each lane evaluates the lower bound L = |clamp(u, box_K) - u|^2 (same fp32
//      expression as the distance); K is scanned -- by the whole wave -- if ANY lane has L <= its d3; the walk
//      stops when the smallest remaining gap (a lower bound of every lane's L) exceeds every lane's d3.
// A scan walks the bucket's 64 known points through wave-uniform addresses (scalar loads: the coordinates arrive in SGPRs and feed
// the packed arithmetic directly, no LDS staging); every lane keeps its own three best
// under the lexicographic (d, k) order, so there is no cross-lane merge at all. A bucket that is not scanned has
// L > d3 for every lane: it holds neither a closer point nor an equal one with a smaller index -- the result is
// the reference's, bit for bit. ~2x fewer instructions per unknown than the wave-per-unknown kernel above.
constexpr int kNnTileThreads = 256;

#ifdef EPNET_NN_STATS  // diagnostic build only (profiles/micro/nn_stats.py): per-wave work counters of the tile kernel
__device__ unsigned long long g_nn_stats[16];
#define EPNET_NN_CNT(slot, v) nn_acc[slot] += (unsigned long long)(v)
#else
#define EPNET_NN_CNT(slot, v)
#endif

constexpr int kNnTileMaxBoxes = 1024;  // known-bucket boxes staged in LDS (24 KB): m <= 65536

__global__ __launch_bounds__(kNnTileThreads) void three_nn_tile_kernel(int n, int npu, int npk,
                                                                       const float4 *__restrict__ sorted_u,
                                                                       const float *__restrict__ boxes_u,
                                                                       const float4 *__restrict__ sorted_k,
                                                                       const float *__restrict__ boxes_k,
                                                                       float *__restrict__ dist2, int *__restrict__ idx) {
    extern __shared__ float s_boxes[];                  // the known buckets' boxes, shared by the 4 waves
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int wg_x, bs;
    xcd_scene_map(wg_x, bs);   // a scene's two indices pass through one XCD's L2
    const int ub = wg_x * (kNnTileThreads / 64) + wave;
    sorted_u += (size_t)bs * npu;
    boxes_u += (size_t)bs * (npu >> 6) * 6;
    sorted_k += (size_t)bs * npk;
    boxes_k += (size_t)bs * (npk >> 6) * 6;
    const int nbk = npk >> 6;
    for (int e = threadIdx.x; e < nbk * 6; e += kNnTileThreads) s_boxes[e] = boxes_k[e];
    __syncthreads();
    if (ub >= (npu >> 6)) return;  // wave-uniform; no block-level barrier below
    const float4 u = sorted_u[(ub << 6) + lane];
    const int ku = __float_as_int(u.w);
    const bool valid = ku >= 0;
    if (!__ballot(valid)) return;  // a bucket of padding
    constexpr unsigned kNone = 0xFFFFFFFFu;
#ifdef EPNET_NN_STATS
    unsigned long long nn_acc[8] = {1, 0, 0, 0, 0, 0, 0, 0};
#endif
    // The three best (d, k) pairs of this lane's unknown as 64-bit keys bits(d) << 32 | (k ^ 0x80000000): d >= +0, so the unsigned
    // order of the keys is the lexicographic (d, k) order. An empty slot is (+inf, INT_MIN): the reference's lists start at
    // (float)1e40 = +inf and take a point only if d < best (interpolate_gpu.cu:30-48), so a distance of +inf (padding rows carry
    // 3e38 coordinates: their d overflows) or NaN (bits above +inf's) never enters -- with this start value the plain key order
    // says exactly that, and the distance bits need no validity mask at all.
    // The keys are held and ordered AS DOUBLES. A finite distance or +inf has a high word <= 0x7F800000, below the double exponent
    // of all ones (0x7FF00000): a finite non-negative double, and for those the order of the values is the order of the bit
    // patterns. A float NaN does NOT stay below it (payloads run up to 0x7FFFFFFF / 0xFFFFFFFF, and NaN inputs propagate their
    // payload): as a double NaN it would make v_min_f64 / v_max_f64 return the other operand and duplicate an entry. So the
    // candidate's distance bits are clamped to +inf's first (one v_min_u32: a NaN of either sign is above 0x7F800000 as an
    // unsigned word); the key (+inf, k) then sorts behind the empty slot (+inf, INT_MIN) and never enters. A sorted insertion
    // is then five v_min_f64 / v_max_f64 (full rate, branch-free, exact; a key with d = 0 is a denormal double: the f64
    // denormal mode of a HIP kernel is IEEE) instead of three 64-bit compares and ten selects.
    constexpr unsigned kInf = 0x7F800000u;
    double e0 = __hiloint2double((int)kInf, 0), e1 = e0, e2 = e0;
#define d2 ((unsigned)__double2hiint(e2))
    // four known points at a time: the squared distances two to an instruction (v_pk_add_f32 / v_pk_mul_f32 round like the
    // scalar forms: same (dx*dx + dy*dy) + dz*dz), one vote for the four, then the insertions of those that matter
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 ux2 = {u.x, u.x}, uy2 = {u.y, u.y}, uz2 = {u.z, u.z};
    auto insert = [&](unsigned db, int k) {
        const double e = __hiloint2double((int)min(db, kInf), (int)((unsigned)k ^ 0x80000000u));
        double t0, r0, t1, r1, t2;  // (inline asm: through fmin / fmax the compiler adds a canonicalising v_max_f64 x,x per operand)
        asm("v_min_f64 %0, |%1|, %2" : "=v"(t0) : "v"(e), "v"(e0));
        asm("v_max_f64 %0, |%1|, %2" : "=v"(r0) : "v"(e), "v"(e0));
        asm("v_min_f64 %0, %1, %2" : "=v"(t1) : "v"(r0), "v"(e1));
        asm("v_max_f64 %0, %1, %2" : "=v"(r1) : "v"(r0), "v"(e1));
        asm("v_min_f64 %0, %1, %2" : "=v"(t2) : "v"(r1), "v"(e2));
        e0 = t0;
        e1 = t1;
        e2 = t2;
        EPNET_NN_CNT(5, 1);
    };
    auto offer4 = [&](const float4 *__restrict__ kp4) {  // four consecutive points of a known bucket, wave-uniform address: scalar loads
        const float4 P0 = kp4[0], P1 = kp4[1], P2 = kp4[2], P3 = kp4[3];
        const f2 ax = {P0.x, P1.x}, ay = {P0.y, P1.y}, az = {P0.z, P1.z};
        const f2 bx = {P2.x, P3.x}, by = {P2.y, P3.y}, bz = {P2.z, P3.z};
        const f2 adx = ux2 - ax, ady = uy2 - ay, adz = uz2 - az;
        const f2 bdx = ux2 - bx, bdy = uy2 - by, bdz = uz2 - bz;
        const f2 da = adx * adx + ady * ady + adz * adz;
        const f2 dbv = bdx * bdx + bdy * bdy + bdz * bdz;
        const unsigned b0 = __float_as_uint(da.x), b1 = __float_as_uint(da.y), b2 = __float_as_uint(dbv.x), b3 = __float_as_uint(dbv.y);
        // (ties with the third entry are rare: a conservative '<=' on the distance alone keeps the vote cheap)
        const unsigned nearest = min(min(b0, b1), min(b2, b3));
        EPNET_NN_CNT(3, 1);
        if (!__ballot(nearest <= d2)) return;
        EPNET_NN_CNT(4, 1);
        insert(b0, __float_as_int(P0.w));
        insert(b1, __float_as_int(P1.w));
        insert(b2, __float_as_int(P2.w));
        insert(b3, __float_as_int(P3.w));
    };
    // A known bucket covers a 4x larger region than a bucket of unknowns (the known set is the 4x sparser FPS subset), so most of
    // its 64 points are out of reach even when the bucket's box is not: the box of each ROW of 16 consecutive points (compact:
    // the points are in spatial order) is found on the fly with four DPP steps per coordinate, and a row is offered only if
    // some lane's exact lower bound to that box is within its third distance
    auto scan = [&](int kb) {
        const float4 *__restrict__ const bucket = sorted_k + ((size_t)kb << 6);
        const float4 kp = bucket[lane];  // one point per lane: the row boxes
        EPNET_NN_CNT(2, 1);
        // padding rows carry 3e38: they only widen a row's box (never a wrong skip); non-finite points are never offered
        float rlx, rhx, rly, rhy, rlz, rhz;
        row16_boxes(kp.x, kp.y, kp.z, rlx, rhx, rly, rhy, rlz, rhz);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float blx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rlx), r * 16));
            const float bhx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rhx), r * 16));
            const float bly = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rly), r * 16));
            const float bhy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rhy), r * 16));
            const float blz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rlz), r * 16));
            const float bhz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rhz), r * 16));
            const float px = __builtin_amdgcn_fmed3f(u.x, blx, bhx), py = __builtin_amdgcn_fmed3f(u.y, bly, bhy),
                        pz = __builtin_amdgcn_fmed3f(u.z, blz, bhz);
            const float ex = u.x - px, ey = u.y - py, ez = u.z - pz;
            const float Lf = ex * ex + ey * ey + ez * ez;  // same expression as the distance: a lower bound of every point's
            // (a NaN bound -- non-finite coordinates -- compares false and keeps the row; an empty third slot is +inf)
            if (!__ballot(valid && !(Lf > __uint_as_float(d2)))) continue;
#pragma unroll
            for (int q = r * 4; q < r * 4 + 4; ++q) offer4(bucket + q * 4);
        }
    };

    // box-to-box gap between this unknown bucket and known bucket kb: a lower bound of every lane's L (each term
    // |u - clamp(u)| >= the gap of its axis, and fp32 sub / mul / add are monotone), kNone for a bucket of padding
    const float *ubx = boxes_u + ub * 6;
    const float ulx = ubx[0], uhx = ubx[1], uly = ubx[2], uhy = ubx[3], ulz = ubx[4], uhz = ubx[5];
    auto gap_of = [&](int kb) -> unsigned {
        if (kb >= nbk) return kNone;
        const float *bx = s_boxes + kb * 6;
        const float gx = fmaxf(0.f, fmaxf(bx[0] - uhx, ulx - bx[1])), gy = fmaxf(0.f, fmaxf(bx[2] - uhy, uly - bx[3])),
                    gz = fmaxf(0.f, fmaxf(bx[4] - uhz, ulz - bx[5]));
        const float g2 = gx * gx + gy * gy + gz * gz;
        return (bx[0] < 3.0e38f && g2 < __builtin_huge_valf()) ? __float_as_uint(g2) : kNone;
    };
    // does any lane still need known bucket kb?  (d2 == +inf: its list is not full yet)
    auto wanted = [&](int kb) -> bool {
        const float *bx = s_boxes + kb * 6;  // wave-uniform: broadcast reads
        const float px = __builtin_amdgcn_fmed3f(u.x, bx[0], bx[1]), py = __builtin_amdgcn_fmed3f(u.y, bx[2], bx[3]),
                    pz = __builtin_amdgcn_fmed3f(u.z, bx[4], bx[5]);
        const float dx = u.x - px, dy = u.y - py, dz = u.z - pz;
        const float Lf = dx * dx + dy * dy + dz * dz;
        const unsigned L = Lf < __builtin_huge_valf() ? __float_as_uint(Lf) : kNone;
        return __ballot(valid && L != kNone && L <= d2) != 0ull;
    };
    if (nbk <= 64) {
        // known buckets in order of increasing gap: the lists tighten early, and once the smallest remaining gap
        // exceeds every lane's third distance no remaining bucket can matter
        unsigned g = gap_of(lane);
        for (;;) {
            const unsigned mn = wave_min_all(g);
            if (mn == kNone || !__ballot(valid && mn <= d2)) break;
            const int kb = (int)__builtin_ctzll(__ballot(g == mn));
            if (lane == kb) g = kNone;
            EPNET_NN_CNT(1, 1);
            if (wanted(kb)) scan(kb);
        }
    } else {
        // many known buckets: seed with the nearest one, then all others in index order
        unsigned best = kNone;
        int seed = 0;
        for (int k0 = 0; k0 < nbk; k0 += 64) {
            const unsigned g = gap_of(k0 + lane);
            const unsigned mn = wave_min_all(g);
            if (mn < best) {
                best = mn;
                seed = k0 + (int)__builtin_ctzll(__ballot(g == mn));
            }
        }
        seed = __builtin_amdgcn_readfirstlane(seed);
        if (best != kNone) scan(seed);
        for (int kb = 0; kb < nbk; ++kb) {
            if (kb == seed && best != kNone) continue;
            if (wanted(kb)) scan(kb);
        }
    }
    if (valid) {
        float *dd = dist2 + ((size_t)bs * n + ku) * 3;
        int *ii = idx + ((size_t)bs * n + ku) * 3;
        // unfilled slot: (float)1e40 = inf, index 0 (interpolate_gpu.cu:30-31)
        const double es[3] = {e0, e1, e2};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const unsigned d = (unsigned)__double2hiint(es[r]);
            dd[r] = __uint_as_float(d);
            ii[r] = d != kInf ? (int)((unsigned)__double2loint(es[r]) ^ 0x80000000u) : 0;
        }
    }
#ifdef EPNET_NN_STATS
    if (lane == 0) {
        for (int s_ = 0; s_ < 6; ++s_) atomicAdd(&g_nn_stats[s_], nn_acc[s_]);
        atomicMax(&g_nn_stats[6], nn_acc[2]);                 // most buckets scanned by one wave
        atomicAdd(&g_nn_stats[7], nn_acc[2] * nn_acc[2]);
        atomicAdd(&g_nn_stats[8 + (nn_acc[2] >= 28 ? 7 : nn_acc[2] / 4)], 1ull);   // histogram of scans per wave, bins of 4
    }
#endif
#undef d2
}

#ifdef EPNET_RUNSUM_STATS
extern "C" int epnet_debug_runsum_stats(unsigned long long *host16) {
    (void)hipMemcpyFromSymbol(host16, HIP_SYMBOL(runsum::g_runsum_stats), sizeof(unsigned long long) * 16);
    unsigned long long zero[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(runsum::g_runsum_stats), zero, sizeof(zero));
    return 0;
}
extern "C" int epnet_debug_pack_stats(unsigned long long *host8) {
    (void)hipMemcpyFromSymbol(host8, HIP_SYMBOL(runsum::g_pack_stats), sizeof(unsigned long long) * 8);
    unsigned long long zero[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(runsum::g_pack_stats), zero, sizeof(zero));
    return 0;
}
#endif

#ifdef EPNET_NN_STATS
extern "C" int epnet_debug_nn_stats(unsigned long long *host16) {
    (void)hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_nn_stats), sizeof(unsigned long long) * 16);
    unsigned long long zero[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_nn_stats), zero, sizeof(zero));
    return 0;
}
#endif

constexpr int kTiThreads = 256;
constexpr int kTiChan = 16;

__global__ __launch_bounds__(kTiThreads) void three_interpolate_kernel(int c, int m, int n,
                                                                       const float *__restrict__ points,
                                                                       const int *__restrict__ idx,
                                                                       const float *__restrict__ weight,
                                                                       float *__restrict__ out) {
    int wg_x, wg_y, bs;
    xcd_scene_map3(wg_x, wg_y, bs);   // a scene's idx / weight rows are read by every channel group: through one XCD's L2
    const int c0 = wg_y * kTiChan;
    const int i0 = (wg_x * kTiThreads + threadIdx.x) * 4;
    if (i0 >= n) return;
    const int cnt = min(4, n - i0);
    int ix[4][3];
    float w[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const bool ok = u < cnt;
            ix[u][j] = ok ? idx[((size_t)bs * n + i0 + u) * 3 + j] : 0;
            w[u][j] = ok ? weight[((size_t)bs * n + i0 + u) * 3 + j] : 0.f;
        }
    const int cend = min(c, c0 + kTiChan);
    const bool vec = (cnt == 4) && (n % 4 == 0) && ((uintptr_t)out % 16 == 0);
    for (int ci = c0; ci < cend; ++ci) {
        const float *src = points + ((size_t)bs * c + ci) * m;
        float *dst = out + ((size_t)bs * c + ci) * n + i0;
        float r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = w[u][0] * src[ix[u][0]] + w[u][1] * src[ix[u][1]] + w[u][2] * src[ix[u][2]];
        if (vec) {
            *reinterpret_cast<float4 *>(dst) = make_float4(r[0], r[1], r[2], r[3]);
        } else {
            for (int u = 0; u < cnt; ++u) dst[u] = r[u];
        }
    }
}

// LDS-staged variant (see gather_rows_lds_kernel in group.hip): R whole rows of `points` (R*m floats) are
// copied into LDS with coalesced loads; the 3 random reads per output then come from LDS.
// grid: (tiles, row chunks, scenes); dynamic LDS: rows * m floats; n % 4 == 0.
__global__ __launch_bounds__(kTiThreads) void three_interpolate_lds_kernel(int c, int m, int n, int rows, int tile,
                                                                           const float *__restrict__ points,
                                                                           const int *__restrict__ idx,
                                                                           const float *__restrict__ weight,
                                                                           float *__restrict__ out) {
    extern __shared__ float s_rows[];
    int wg_x, wg_y, bs;
    xcd_scene_map3(wg_x, wg_y, bs);   // a scene's idx / weight rows are read by every row chunk: through one XCD's L2
    const int c0 = wg_y * rows;
    const int nr = min(rows, c - c0);
    const float *src = points + ((size_t)bs * c + c0) * m;
    const int total = nr * m;
    if ((m & 3) == 0 && ((uintptr_t)src & 15) == 0) {
        const float4 *src4 = reinterpret_cast<const float4 *>(src);
        float4 *dst4 = reinterpret_cast<float4 *>(s_rows);
        for (int e = threadIdx.x; e < total / 4; e += kTiThreads) dst4[e] = src4[e];
    } else {
        for (int e = threadIdx.x; e < total; e += kTiThreads) s_rows[e] = src[e];
    }
    __syncthreads();
    const int i_begin = wg_x * tile, i_end = min(n, i_begin + tile);
    float *dst_base = out + ((size_t)bs * c + c0) * n;
    for (int i0 = i_begin + threadIdx.x * 4; i0 < i_end; i0 += kTiThreads * 4) {
        int ix[4][3];
        float w[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                ix[u][j] = idx[((size_t)bs * n + i0 + u) * 3 + j];
                w[u][j] = weight[((size_t)bs * n + i0 + u) * 3 + j];
            }
        const float *row = s_rows;
        float *dst = dst_base + i0;
        for (int r = 0; r < nr; ++r) {
            float o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) o[u] = w[u][0] * row[ix[u][0]] + w[u][1] * row[ix[u][1]] + w[u][2] * row[ix[u][2]];
            store_stream(dst, o[0], o[1], o[2], o[3]);
            row += m;
            dst += n;
        }
    }
}

// Same, with the staged rows interleaved FOUR channels to a known point ([m][4] floats per group of four rows): one 16-byte
// LDS read then serves four channel rows -- a quarter of the LDS instructions and less than half the LDS cycles per output
// byte of the row-major kernel above (random 4-byte reads: two 32-lane groups over 32 banks; random 16-byte reads: four
// 16-lane groups over 16 bank quads), which is what bounded it (3.5 TB/s against 5.2 for a plain gather).
// rows is a multiple of 4 (c % 4 == 0), m % 4 == 0, n % 4 == 0; dynamic LDS: rows * m floats.
template <int U>   // unknowns per thread: 4 (16-byte stores), or 1 for the coarse levels (n < 1024: a thread per unknown fills the workgroup)
__global__ __launch_bounds__(kTiThreads) void three_interpolate_lds4_kernel(int c, int m, int n, int rows, int tile,
                                                                            const float *__restrict__ points,
                                                                            const int *__restrict__ idx,
                                                                            const float *__restrict__ weight,
                                                                            float *__restrict__ out) {
    extern __shared__ float4 s_quad[];  // [rows / 4][m] float4 = the four channels of one known point
    int wg_x, wg_y, bs;
    xcd_scene_map3(wg_x, wg_y, bs);   // a scene's idx / weight rows are read by every row chunk: through one XCD's L2
    const int c0 = wg_y * rows;
    const int nr = min(rows, c - c0);  // multiple of 4
    const int groups = nr >> 2, m4 = m >> 2;
    const float *src = points + ((size_t)bs * c + c0) * m;
    for (int e = threadIdx.x; e < groups * m4; e += kTiThreads) {
        const int g = e / m4, j4 = e - g * m4;
        const float4 *row = reinterpret_cast<const float4 *>(src + (size_t)g * 4 * m) + j4;
        const float4 a = row[0], b4 = row[m4], cc = row[2 * m4], d = row[3 * m4];
        float4 *dst = s_quad + g * m + j4 * 4;
        dst[0] = make_float4(a.x, b4.x, cc.x, d.x);
        dst[1] = make_float4(a.y, b4.y, cc.y, d.y);
        dst[2] = make_float4(a.z, b4.z, cc.z, d.z);
        dst[3] = make_float4(a.w, b4.w, cc.w, d.w);
    }
    __syncthreads();
    const int i_begin = wg_x * tile, i_end = min(n, i_begin + tile);
    float *dst_base = out + ((size_t)bs * c + c0) * n;
    for (int i0 = i_begin + threadIdx.x * U; i0 < i_end; i0 += kTiThreads * U) {
        int ix[U][3];
        float w[U][3];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                ix[u][j] = idx[((size_t)bs * n + i0 + u) * 3 + j];
                w[u][j] = weight[((size_t)bs * n + i0 + u) * 3 + j];
            }
        const float4 *quad = s_quad;
        float *dst = dst_base + i0;
        for (int g = 0; g < groups; ++g) {
            float4 v[U][3];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < 3; ++j) v[u][j] = quad[ix[u][j]];
            // the reference's expression per output: w0 * p0 + w1 * p1 + w2 * p2, left to right (interpolate_gpu.cu:95-96)
#define EPNET_TI_VAL(u_, F_) (w[u_][0] * v[u_][0].F_ + w[u_][1] * v[u_][1].F_ + w[u_][2] * v[u_][2].F_)
#define EPNET_TI_ROW(F_)                                                                                          \
    if constexpr (U == 4) store_stream(dst, EPNET_TI_VAL(0, F_), EPNET_TI_VAL(1, F_), EPNET_TI_VAL(2, F_), EPNET_TI_VAL(3, F_)); \
    else __builtin_nontemporal_store(EPNET_TI_VAL(0, F_), dst);                                                    \
    dst += n;
            EPNET_TI_ROW(x)
            EPNET_TI_ROW(y)
            EPNET_TI_ROW(z)
            EPNET_TI_ROW(w)
#undef EPNET_TI_ROW
#undef EPNET_TI_VAL
            quad += m;
        }
    }
}

// gradient: rows of grad_points (m floats each) accumulated in LDS, see group.hip
__global__ __launch_bounds__(kTiThreads) void three_interpolate_grad_lds_kernel(int c, int n, int m, int rows,
                                                                                const float *__restrict__ grad_out,
                                                                                const int *__restrict__ idx,
                                                                                const float *__restrict__ weight,
                                                                                float *__restrict__ grad_points) {
    extern __shared__ float acc[];
    const int bs = blockIdx.y;
    const int c0 = blockIdx.x * rows;
    const int nr = min(rows, c - c0);
    for (int e = threadIdx.x; e < nr * m; e += kTiThreads) acc[e] = 0.f;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += kTiThreads) {
        const int *ix = idx + ((size_t)bs * n + i) * 3;
        const float *w = weight + ((size_t)bs * n + i) * 3;
        const int i0 = ix[0], i1 = ix[1], i2 = ix[2];
        const float w0 = w[0], w1 = w[1], w2 = w[2];
        for (int r = 0; r < nr; ++r) {
            const float g = grad_out[((size_t)bs * c + c0 + r) * n + i];
            atomicAdd(&acc[r * m + i0], g * w0);
            atomicAdd(&acc[r * m + i1], g * w1);
            atomicAdd(&acc[r * m + i2], g * w2);
        }
    }
    __syncthreads();
    float *gp = grad_points + ((size_t)bs * c + c0) * m;
    for (int e = threadIdx.x; e < nr * m; e += kTiThreads) gp[e] += acc[e];
}

__global__ __launch_bounds__(kTiThreads) void three_interpolate_grad_atomic_kernel(int c, int n, int m,
                                                                                   const float *__restrict__ grad_out,
                                                                                   const int *__restrict__ idx,
                                                                                   const float *__restrict__ weight,
                                                                                   float *__restrict__ grad_points) {
    const int bs = blockIdx.z, ci = blockIdx.y;
    const int i = blockIdx.x * kTiThreads + threadIdx.x;
    if (i >= n) return;
    const float g = grad_out[((size_t)bs * c + ci) * n + i];
    const int *ix = idx + ((size_t)bs * n + i) * 3;
    const float *w = weight + ((size_t)bs * n + i) * 3;
    float *gp = grad_points + ((size_t)bs * c + ci) * m;
    atomicAdd(gp + ix[0], g * w[0]);
    atomicAdd(gp + ix[1], g * w[1]);
    atomicAdd(gp + ix[2], g * w[2]);
}

}  // namespace epnet

using namespace epnet;

extern "C" int epnet_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx,
                              epnet_stream_t stream) {
    EPNET_REQUIRE(b >= 0 && n >= 0 && m >= 0);
    if (b == 0 || n == 0) return EPNET_OK;
    EPNET_REQUIRE(unknown && dist2 && idx && (known || m == 0));
    EPNET_REQUIRE(b <= 65535);
    dim3 grid(div_up(n, kNnPerBlock), b);
    hipLaunchKernelGGL(three_nn_kernel, grid, dim3(kNnThreads), 0, (hipStream_t)stream, n, m, unknown, known, dist2, idx);
    return check_launch("three_nn");
}

extern "C" int epnet_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                                       const float *weight, float *out, epnet_stream_t stream) {
    EPNET_REQUIRE(b >= 0 && c >= 0 && m >= 0 && n >= 0);
    if (b == 0 || c == 0 || n == 0) return EPNET_OK;
    EPNET_REQUIRE(points && idx && weight && out);
    if (b > 65535 || div_up(c, kTiChan) > 65535) return EPNET_ELIMIT;
    constexpr int kLdsBudget = 64 * 1024;
    const bool quad_ok = (c & 3) == 0 && (m & 3) == 0 && ((uintptr_t)points & 15) == 0;
    if (c >= 8 && (n % 4) == 0 && n >= 1024 && (size_t)m * 4 <= kLdsBudget && ((uintptr_t)out % 16) == 0) {
        int rows = kLdsBudget / (m * 4);
        if (rows > c) rows = c;
        if (rows > 32) rows = 32;
        if ((c & 3) == 0 && rows >= 4) rows &= ~3;  // whole groups of four channels for the interleaved kernel
        const int chunks = div_up(c, rows);
        int tiles = div_up(1024, b * chunks);
        const int max_tiles = max(1, n / 2048);
        if (tiles > max_tiles) tiles = max_tiles;
        if (tiles < 1) tiles = 1;
        int tile = div_up(n, tiles);
        tile = (tile + 1023) / 1024 * 1024;
        tiles = div_up(n, tile);
        if (chunks <= 65535) {
            if (quad_ok && (rows & 3) == 0)
                hipLaunchKernelGGL(three_interpolate_lds4_kernel<4>, dim3(tiles, chunks, b), dim3(kTiThreads), (size_t)rows * m * 4,
                                   (hipStream_t)stream, c, m, n, rows, tile, points, idx, weight, out);
            else
                hipLaunchKernelGGL(three_interpolate_lds_kernel, dim3(tiles, chunks, b), dim3(kTiThreads), (size_t)rows * m * 4,
                                   (hipStream_t)stream, c, m, n, rows, tile, points, idx, weight, out);
            return check_launch("three_interpolate");
        }
    }
    // the coarse levels of the FP pyramid (64 -> 256, 256 -> 1024 points, 512 - 1024 channels): a thread per unknown, more channel
    // rows per workgroup (the known rows are short). The direct kernel below runs these shapes at 1.7 TB/s
    if (quad_ok && c >= 8 && n >= 64 && n < 1024 && m >= 4 && (size_t)m * 16 <= kLdsBudget) {
        int rows = kLdsBudget / (m * 4);
        if (rows > c) rows = c;
        if (rows > 64) rows = 64;
        rows &= ~3;
        const int chunks = div_up(c, rows);
        const int tile = (n + kTiThreads - 1) / kTiThreads * kTiThreads;   // one tile: n < 1024 = 4 passes of the workgroup at most
        if (rows >= 4 && chunks <= 65535) {
            hipLaunchKernelGGL(three_interpolate_lds4_kernel<1>, dim3(1, chunks, b), dim3(kTiThreads), (size_t)rows * m * 4,
                               (hipStream_t)stream, c, m, n, rows, tile, points, idx, weight, out);
            return check_launch("three_interpolate");
        }
    }
    dim3 grid(div_up(div_up(n, 4), kTiThreads), div_up(c, kTiChan), b);
    hipLaunchKernelGGL(three_interpolate_kernel, grid, dim3(kTiThreads), 0, (hipStream_t)stream, c, m, n, points, idx,
                       weight, out);
    return check_launch("three_interpolate");
}

extern "C" int epnet_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                            const float *weight, float *grad_points, epnet_stream_t stream) {
    EPNET_REQUIRE(b >= 0 && c >= 0 && m >= 0 && n >= 0);
    if (b == 0 || c == 0 || n == 0 || m == 0) return EPNET_OK;
    EPNET_REQUIRE(grad_out && idx && weight && grad_points);
    if (b > 65535) return EPNET_ELIMIT;
    constexpr int kLdsBudget = 64 * 1024;
    hipStream_t s = (hipStream_t)stream;
    if ((size_t)m * 4 <= kLdsBudget) {
        int rows = kLdsBudget / (m * 4);
        if (rows > 8) rows = 8;
        if (rows > c) rows = c;
        dim3 grid(div_up(c, rows), b);
        hipLaunchKernelGGL(three_interpolate_grad_lds_kernel, grid, dim3(kTiThreads), (size_t)rows * m * 4, s, c, n, m,
                           rows, grad_out, idx, weight, grad_points);
    } else {
        if (c > 65535) return EPNET_ELIMIT;
        dim3 grid(div_up(n, kTiThreads), c, b);
        hipLaunchKernelGGL(three_interpolate_grad_atomic_kernel, grid, dim3(kTiThreads), 0, s, c, n, m, grad_out, idx,
                           weight, grad_points);
    }
    return check_launch("three_interpolate_grad");
}

static int nn_padded(int m) { return (m + 63) / 64 * 64; }

extern "C" size_t epnet_three_nn_workspace_bytes(int b, int n, int m) {
    if (b <= 0 || n <= 0 || m < kNnIxMinKnown || m > 65536) return 0;  // few known points: the direct scan is used
    const size_t np = (size_t)nn_padded(m);
    return (size_t)b * (np * sizeof(float4) + (np / 64) * 6 * sizeof(float));
}

extern "C" int epnet_three_nn_ws(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx,
                                 void *workspace, size_t workspace_bytes, epnet_stream_t stream) {
    const size_t need = epnet_three_nn_workspace_bytes(b, n, m);
    if (need == 0) return epnet_three_nn(b, n, m, unknown, known, dist2, idx, stream);
    EPNET_REQUIRE(unknown && known && dist2 && idx && workspace);
    if (workspace_bytes < need) return EPNET_ENOMEM;
    if (((uintptr_t)workspace & 15) != 0) return EPNET_EINVAL;
    EPNET_REQUIRE(b <= 65535);
    hipStream_t s = (hipStream_t)stream;
    const int np = nn_padded(m);
    float4 *sorted = (float4 *)workspace;
    float *boxes = (float *)(sorted + (size_t)b * np);
    int rc = spatial_index_launch(b, m, np, known, sorted, boxes, nullptr, s);
    if (rc) return rc;
    dim3 grid(div_up(n, kNnIxThreads / 64), b);
    hipLaunchKernelGGL(three_nn_indexed_kernel, grid, dim3(kNnIxThreads), 0, s, n, m, np, unknown, sorted, boxes, dist2, idx);
    return check_launch("three_nn indexed");
}

// atomic-free gradient (runsum.h): the 3n (unknown, neighbour) pairs grouped by their known point once, then equal shares
// of the sorted pairs summed per thread (w * grad_out) out of LDS-staged grad_out rows
extern "C" size_t epnet_three_interpolate_grad_workspace_bytes(int b, int n, int m) {
    if (b <= 0 || !runsum::usable(m, 3, (long long)n * 3, n)) return 0;
    return runsum::workspace_bytes(b, m, 3, n, true);
}

extern "C" int epnet_three_interpolate_grad_ws(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                               const float *weight, float *grad_points, void *workspace,
                                               size_t workspace_bytes, epnet_stream_t stream) {
    const size_t need = epnet_three_interpolate_grad_workspace_bytes(b, n, m);
    if (need == 0 || ((uintptr_t)grad_out & 15))
        return epnet_three_interpolate_grad(b, c, n, m, grad_out, idx, weight, grad_points, stream);
    EPNET_REQUIRE(b >= 0 && c >= 0);
    if (b == 0 || c == 0) return EPNET_OK;
    EPNET_REQUIRE(grad_out && idx && weight && grad_points && workspace);
    return runsum::launch<true>(b, c, m, 3, n, grad_out, (size_t)c * n, idx, weight, grad_points, workspace, workspace_bytes,
                                (hipStream_t)stream, "three_interpolate_grad");
}

// three_nn over scene indices built beforehand (epnet_scene_index_build): of the known set, and optionally of the
// unknown set as well. Same results as epnet_three_nn.
extern "C" int epnet_three_nn_indexed(int b, int n, int m, const float *unknown, const float *known,
                                      const void *unknown_index, size_t unknown_index_bytes, const void *known_index,
                                      size_t known_index_bytes, float *dist2, int *idx, epnet_stream_t stream) {
    const size_t need_k = scene_index_bytes(b, m);
    if (need_k == 0 || !known_index) return epnet_three_nn(b, n, m, unknown, known, dist2, idx, stream);
    EPNET_REQUIRE(b >= 0 && n >= 0);
    if (b == 0 || n == 0) return EPNET_OK;
    EPNET_REQUIRE(dist2 && idx);
    if (known_index_bytes < need_k) return EPNET_ENOMEM;
    EPNET_REQUIRE(b <= 65535);
    hipStream_t s = (hipStream_t)stream;
    const int npk = scene_index_np(m);
    const float4 *sorted_k = (const float4 *)known_index;
    const float *boxes_k = (const float *)(sorted_k + (size_t)b * npk);
    const size_t need_u = scene_index_bytes(b, n);
    // the bucket-of-unknowns kernel does a long serial walk per wave: it wins once there are enough buckets to fill
    // the chip (measured: 4096 buckets 0.20 vs 0.26 ms, 1024 buckets 0.12 vs 0.06 ms)
    const char *env_min = getenv("EPNET_NN_TILE_MIN_BUCKETS");  // tests force either kernel
    const long long min_buckets = env_min ? atoll(env_min) : 4096;
    if (need_u != 0 && unknown_index && (npk >> 6) <= kNnTileMaxBoxes && (long long)b * (scene_index_np(n) >> 6) >= min_buckets) {
        if (unknown_index_bytes < need_u) return EPNET_ENOMEM;
        const int npu = scene_index_np(n);
        const float4 *sorted_u = (const float4 *)unknown_index;
        const float *boxes_u = (const float *)(sorted_u + (size_t)b * npu);
        dim3 grid(div_up(npu >> 6, kNnTileThreads / 64), b);
        hipLaunchKernelGGL(three_nn_tile_kernel, grid, dim3(kNnTileThreads), (size_t)(npk >> 6) * 6 * sizeof(float), s, n, npu,
                           npk, sorted_u, boxes_u, sorted_k, boxes_k, dist2, idx);
        return check_launch("three_nn");
    }
    EPNET_REQUIRE(unknown);
    dim3 grid(div_up(n, kNnIxThreads / 64), b);
    hipLaunchKernelGGL(three_nn_indexed_kernel, grid, dim3(kNnIxThreads), 0, s, n, m, npk, unknown, sorted_k, boxes_k, dist2, idx);
    return check_launch("three_nn");
}
