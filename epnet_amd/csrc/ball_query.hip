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
#include <stdlib.h>

#include "common.h"
#include "dpp.h"
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

// T threads, 2^BITS grid cells: <1024, 14> for big scenes; <256, 12> for n <= 4096, where 4096 cells are plenty and a
// workgroup of 4 waves with 18 KB of LDS finds room on a CU that the wide kernels of the pipelined stack occupy
#ifdef EPNET_BQ_STATS  // diagnostic build only (profiles/micro/bq_stats.py): per-wave counters of bq_query2_kernel
__device__ unsigned long long g_bq_stats[16];
#endif
#ifdef EPNET_IX_STATS  // diagnostic build only (profiles/micro/ix_stats.py): phase counters of the index build (wave 0)
__device__ unsigned long long g_ix_stats[8];
#define EPNET_IX_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define EPNET_IX_STAMP(var)
#endif

template <int kIxThreads, int BITS>
__global__ __launch_bounds__(kIxThreads) void bq_index_kernel(int n, int np, const float *__restrict__ xyz,
                                                              float4 *__restrict__ sorted, float *__restrict__ boxes,
                                                              float *__restrict__ qboxes, int n_src = 0,
                                                              const int *__restrict__ src_idx = nullptr,
                                                              float *__restrict__ gathered = nullptr) {
    constexpr int kCells = 1 << BITS;
    extern __shared__ int s_hist[];  // cell histogram / running offsets (spatial.h)
    __shared__ float s_box[6][16];
    __shared__ int s_part[16];
    const int q = threadIdx.x, lane = q & 63, wave = q >> 6;
    // one workgroup per scene on the sampling chain of a level: beside the wide kernels of the pipelined stack (32 waves of a ball
    // query per CU) its waves would get a ninth of the issue slots -- a 23 us build took 204 us there
    __builtin_amdgcn_s_setprio(3);
    // src_idx (n <= kIxThreads * 16 only): the scene's points are rows src_idx[0 .. n) of a cloud of n_src points -- the centres a
    // sampling has just picked -- and are written out in that order as well (gathered): the centre gather of an SA level and the
    // index build of the next one in one dispatch
    xyz += (size_t)blockIdx.x * (src_idx ? n_src : n) * 3;
    if (src_idx) {
        src_idx += (size_t)blockIdx.x * n;
        gathered += (size_t)blockIdx.x * n * 3;
    }
    sorted += (size_t)blockIdx.x * np;
    boxes += (size_t)blockIdx.x * (np / 64) * 6;
    // counting sort by cell, scattering the points (with their original index) straight to global memory.
    // Up to kIxPerThread points per thread the cloud is read ONCE into registers (bounding box, cell codes and the
    // scatter all work from there); bigger clouds re-read it per phase.
    constexpr int per = kCells / kIxThreads;
    constexpr int per_shift = per == 16 ? 4 : per == 8 ? 3 : per == 4 ? 2 : -1;
    static_assert(per_shift > 0, "scan layout");
    constexpr int kIxPerThread = 16;
    const bool in_regs = n <= kIxThreads * kIxPerThread;  // block-uniform
    float px[kIxPerThread], py[kIxPerThread], pz[kIxPerThread];
    float lo[3], ext[3];
    EPNET_IX_STAMP(t_0);
    if (in_regs) {
        float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
        // all of a thread's points are requested before the first one is used, and nothing below branches on `k < n`: with the
        // box update under an `if` every iteration was a basic block of its own and its load was waited for before the next
        // was issued -- 16 serial trips to HBM at the head of the sampling chain of every level
        int row[kIxPerThread];
#pragma unroll
        for (int i = 0; i < kIxPerThread; ++i) {
            const int k = q + i * kIxThreads;
            const int kk = k < n ? k : 0;
            row[i] = src_idx ? src_idx[kk] : kk;
        }
#pragma unroll
        for (int i = 0; i < kIxPerThread; ++i) {
            px[i] = xyz[row[i] * 3 + 0];
            py[i] = xyz[row[i] * 3 + 1];
            pz[i] = xyz[row[i] * 3 + 2];
        }
        if (src_idx) {
#pragma unroll
            for (int i = 0; i < kIxPerThread; ++i) {
                const int k = q + i * kIxThreads;
                if (k < n) {
                    gathered[k * 3 + 0] = px[i];
                    gathered[k * 3 + 1] = py[i];
                    gathered[k * 3 + 2] = pz[i];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < kIxPerThread; ++i) {
            const bool ok = q + i * kIxThreads < n;
            mn[0] = fminf(mn[0], ok ? px[i] : 3.4e38f); mx[0] = fmaxf(mx[0], ok ? px[i] : -3.4e38f);
            mn[1] = fminf(mn[1], ok ? py[i] : 3.4e38f); mx[1] = fmaxf(mx[1], ok ? py[i] : -3.4e38f);
            mn[2] = fminf(mn[2], ok ? pz[i] : 3.4e38f); mx[2] = fmaxf(mx[2], ok ? pz[i] : -3.4e38f);
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float l = wave_minf_all(mn[a]), h = wave_maxf_all(mx[a]);
            if (lane == 0) {
                s_box[a][wave] = l;
                s_box[3 + a][wave] = h;
            }
        }
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float l = s_box[a][0], h = s_box[3 + a][0];
            for (int w = 1; w < kIxThreads / 64; ++w) {
                l = fminf(l, s_box[a][w]);
                h = fmaxf(h, s_box[3 + a][w]);
            }
            lo[a] = l;
            ext[a] = h - l;
        }
    } else {
        block_bbox3(xyz, n, s_box, lo, ext);
    }
    const CellGrid g = make_cell_grid(lo, ext, BITS);
    EPNET_IX_STAMP(t_1);
    for (int i = q; i < kCells + kCells / per + 64; i += kIxThreads) s_hist[i] = 0;
    __syncthreads();
    EPNET_IX_STAMP(t_2);
    // in_regs: the histogram pass keeps what its atomics return -- a point's rank inside its cell -- so the scatter below needs
    // no second round of atomics on the same (hot: a cell next to the sensor holds hundreds of points) words: 46 % of the kernel
    int code[kIxPerThread], rank[kIxPerThread];
    if (in_regs) {
#pragma unroll
        for (int i = 0; i < kIxPerThread; ++i) code[i] = hist_at((int)cell_code(g, px[i], py[i], pz[i]), per_shift);
#pragma unroll
        for (int i = 0; i < kIxPerThread; ++i)   // (a slot beyond n counts on one of the 64 spare words behind the histogram)
            rank[i] = atomicAdd(&s_hist[q + i * kIxThreads < n ? code[i] : kCells + kCells / per + lane], 1);
    } else {
        for (int k = q; k < n; k += kIxThreads)
            atomicAdd(&s_hist[hist_at((int)cell_code(g, xyz[k * 3 + 0], xyz[k * 3 + 1], xyz[k * 3 + 2]), per_shift)], 1);
    }
    __syncthreads();
    EPNET_IX_STAMP(t_3);
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
    EPNET_IX_STAMP(t_4);
    auto bucket_box = [&](int p, const float4 v) {  // one wave handles one bucket at a time (all 64 lanes active)
        const bool real = p < n;
        const float cx_ = canonical(v.x), cy_ = canonical(v.y), cz_ = canonical(v.z);   // (quiet NaNs drop out of the box, as with fminf)
        float mnx = real ? cx_ : 3.4e38f, mxx = real ? cx_ : -3.4e38f, mny = real ? cy_ : 3.4e38f, mxy = real ? cy_ : -3.4e38f,
              mnz = real ? cz_ : 3.4e38f, mxz = real ? cz_ : -3.4e38f;
        wave_box(mnx, mxx, mny, mxy, mnz, mxz);
        if (lane == 0) {
            float *bx = boxes + (p >> 6) * 6;
            const bool any = mnx <= mxx;  // an all-padding bucket gets a box no ball can reach
            bx[0] = any ? mnx : 3.0e38f; bx[1] = any ? mxx : 3.0e38f;
            bx[2] = any ? mny : 3.0e38f; bx[3] = any ? mxy : 3.0e38f;
            bx[4] = any ? mnz : 3.0e38f; bx[5] = any ? mxz : 3.0e38f;
        }
    };
    const bool staged = in_regs && (np & 255) == 0;  // (three_nn's own index of a known set pads to 64 only: no whole-bucket quarters)
    if (staged) {
        // The sorted order goes through LDS, a quarter of the scene at a time: the points land in the staging buffer (scattered
        // 16-byte LDS writes), leave for global memory as whole rows (the scattered 16-byte global stores this replaces were
        // 46 % of the kernel: 4 M partial-line writes per 256 scenes) and give their bucket boxes on the way out (no read-back
        // of what was just written: another 20 %).
        float4 *s_stage = reinterpret_cast<float4 *>(s_hist + kCells + kCells / per + 64);
        int pos[kIxPerThread];
#pragma unroll
        for (int i = 0; i < kIxPerThread; ++i) pos[i] = s_hist[code[i]] + rank[i];   // start of the cell + rank inside it
        // a quarter of the scene per round; the 256-thread variant at most 512 points (8 KB): with 26 KB of LDS in all it still finds
        // room on a CU that holds two 64 KB workgroups of a row gather -- at 34 KB it waited for the whole gather to drain (260 us)
        // (a multiple of 64: np is a multiple of 256 here. A scene index has a power of two, three_nn's own index of a known set any
        // multiple of 64 -- 2304, 2816, 3328, 3840 are not whole rounds of 512: the last round is shorter)
        const int quarter = kIxThreads == 256 ? min(np >> 2, 512) : np >> 2;
        for (int h = 0; h * quarter < np; ++h) {
            const int base = h * quarter, len = min(quarter, np - base);
            for (int p = q; p < len; p += kIxThreads)   // padding rows: never inside a ball
                if (base + p >= n) s_stage[p] = make_float4(3.0e38f, 3.0e38f, 3.0e38f, __int_as_float(-1));
#pragma unroll
            for (int i = 0; i < kIxPerThread; ++i) {
                const int k = q + i * kIxThreads;
                if (k < n && (unsigned)(pos[i] - base) < (unsigned)quarter)
                    s_stage[pos[i] - base] = make_float4(px[i], py[i], pz[i], __int_as_float(k));
            }
            __syncthreads();
            for (int p = q; p < len; p += kIxThreads) {
                const float4 v = s_stage[p];
                sorted[base + p] = v;
                bucket_box(base + p, v);
            }
            __syncthreads();
        }
    } else {
        if (in_regs) {
#pragma unroll
            for (int i = 0; i < kIxPerThread; ++i) {
                const int k = q + i * kIxThreads;
                if (k < n) sorted[s_hist[code[i]] + rank[i]] = make_float4(px[i], py[i], pz[i], __int_as_float(k));
            }
        } else {
            for (int k = q; k < n; k += kIxThreads) {
                const float x = xyz[k * 3 + 0], y = xyz[k * 3 + 1], z = xyz[k * 3 + 2];
                const int pos = atomicAdd(&s_hist[hist_at((int)cell_code(g, x, y, z), per_shift)], 1);
                sorted[pos] = make_float4(x, y, z, __int_as_float(k));
            }
        }
        for (int p = n + q; p < np; p += kIxThreads)  // padding: never inside a ball
            sorted[p] = make_float4(3.0e38f, 3.0e38f, 3.0e38f, __int_as_float(-1));
        __threadfence_block();
        __syncthreads();  // the scattered points are read back by other waves of this workgroup below
        for (int p = q; p < np; p += kIxThreads) bucket_box(p, sorted[p]);
    }
#ifdef EPNET_IX_STATS
    {
        EPNET_IX_STAMP(t_6);
        if (q == 0) {
            atomicAdd(&g_ix_stats[0], t_1 - t_0); atomicAdd(&g_ix_stats[1], t_2 - t_1); atomicAdd(&g_ix_stats[2], t_3 - t_2);
            atomicAdd(&g_ix_stats[3], t_4 - t_3); atomicAdd(&g_ix_stats[4], t_6 - t_4);
            atomicAdd(&g_ix_stats[7], 1ull);
        }
    }
#endif
    if (!qboxes) return;
    // second level: one box per 4 consecutive buckets (256 sorted points)
    qboxes += (size_t)blockIdx.x * (np / 256) * 6;
    __threadfence_block();
    __syncthreads();
    for (int g = q; g < np / 256; g += kIxThreads) {
        float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
        for (int k = 0; k < 4; ++k) {
            const float *bx = boxes + (g * 4 + k) * 6;
            if (bx[0] < 3.0e38f)  // not an all-padding bucket
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    mn[a] = fminf(mn[a], bx[2 * a]);
                    mx[a] = fmaxf(mx[a], bx[2 * a + 1]);
                }
        }
        const bool any = mn[0] <= mx[0];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            qboxes[g * 6 + 2 * a] = any ? mn[a] : 3.0e38f;
            qboxes[g * 6 + 2 * a + 1] = any ? mx[a] : 3.0e38f;
        }
    }
}

constexpr int kQThreads = 256;

// |clamp(c, box) - c|^2 < r^2, box = (min x, max x, min y, max y, min z, max z)
__device__ __forceinline__ bool box_near(const float *__restrict__ bx, float cx, float cy, float cz, float radius2) {
    const float2 bxx = *reinterpret_cast<const float2 *>(bx), byy = *reinterpret_cast<const float2 *>(bx + 2),
                 bzz = *reinterpret_cast<const float2 *>(bx + 4);
    const float px = __builtin_amdgcn_fmed3f(cx, bxx.x, bxx.y), py = __builtin_amdgcn_fmed3f(cy, byy.x, byy.y),
                pz = __builtin_amdgcn_fmed3f(cz, bzz.x, bzz.y);
    const float dx = cx - px, dy = cy - py, dz = cz - pz;
    return (dx * dx + dy * dy + dz * dz) < radius2;
}

// One wave per centre. Two levels of boxes: the lanes test the quad boxes (256 points each), then -- 16 candidate
// quads at a time -- the 4 bucket boxes of each candidate quad, then the 64 points of every near bucket. A ball holds
// a handful of points (median 1-2 on KITTI-like scenes), so the hits are appended to a 64-entry list and ordered
// by counting; only a ball with more than 64 hits takes a second walk through the N-bit bitmap.
// The K scales of an MSG level (same centres, same points, nested balls) share ONE walk: boxes are tested against
// the largest radius and every distance is computed once.
// DPL = bitmap dwords per lane = np / 2048 (np >= 2048)

// the candidate walk: visit(p, d2) for the 64 points p (one per lane) of every bucket whose box is nearer than r2
template <int DPL, typename F>
__device__ __forceinline__ void bq_walk(int lane, int np, float r2, float cx, float cy, float cz,
                                        const float4 *__restrict__ sorted, const float *__restrict__ boxes,
                                        const float *__restrict__ qboxes, int *quads, F &&visit) {
    // The kernel is bound by its chain of dependent loads (boxes -> boxes -> points), not by arithmetic: the rows of
    // up to 4 near buckets are requested together before any of them is tested.
    auto scan_buckets = [&](unsigned long long cand, auto &&bucket_of) {
        while (cand) {
            float4 p[4];
            int nb = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (cand) {
                    const int bb = bucket_of((int)__builtin_ctzll(cand));
                    cand &= cand - 1ull;
                    p[k] = sorted[(bb << 6) + lane];
                    nb = k + 1;
                }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < nb) {
                    const float dx = cx - p[k].x, dy = cy - p[k].y, dz = cz - p[k].z;
                    visit(p[k], dx * dx + dy * dy + dz * dz);
                }
        }
    };
    if constexpr (DPL <= 2) {  // <= 64 buckets: one round over the bucket boxes, no second level
        const bool bnear = lane < (np >> 6) && box_near(boxes + lane * 6, cx, cy, cz, r2);
        scan_buckets(__ballot(bnear), [](int l) { return l; });
    } else {
        const int nq = np >> 8;
        for (int q0 = 0; q0 < nq; q0 += 64) {
            const int qd = q0 + lane;
            const bool qnear = qd < nq && box_near(qboxes + qd * 6, cx, cy, cz, r2);
            const unsigned long long qmask = __ballot(qnear);
            if (!qmask) continue;
            const int nqc = (int)__popcll(qmask);
            if (qnear) quads[popc_below(qmask)] = qd;
            __builtin_amdgcn_wave_barrier();
            for (int s0 = 0; s0 < nqc; s0 += 16) {
                const int slot = s0 + (lane >> 2);
                int bid = 0;
                bool bnear = false;
                if (slot < nqc) {
                    bid = quads[slot] * 4 + (lane & 3);
                    bnear = box_near(boxes + bid * 6, cx, cy, cz, r2);
                }
                scan_buckets(__ballot(bnear), [&](int l) { return __builtin_amdgcn_readlane(bid, l); });
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// <= 64 hits in `hits`: order them by original index (rank = number of smaller indices; they are distinct) and write
// "the first nsample in index order, padded with the first" (ball_query_gpu.cu:29-43); an empty ball is all zeros
__device__ __forceinline__ void bq_emit_list(int lane, const int *hits, int cnt, int nsample, int *__restrict__ out) {
    __builtin_amdgcn_wave_barrier();
    const int mine = lane < cnt ? hits[lane] : 0x7fffffff;
    int rank = 0;
    for (int j = 0; j < cnt; ++j) rank += __builtin_amdgcn_readlane(mine, j) < mine ? 1 : 0;
    if (lane < cnt && rank < nsample) out[rank] = mine;
    int first = 0;
    if (cnt) first = __builtin_amdgcn_readlane(mine, (int)__builtin_ctzll(__ballot(lane < cnt && rank == 0)));
    for (int l = cnt + lane; l < nsample; l += 64) out[l] = first;
}

// 64 < cnt <= 64 * RM hits in `hits`, nsample <= 64: only the nsample SMALLEST original indices are wanted. They are found
// without a second walk: the indices are distinct integers below np, so count(hits < T) grows by at most one per unit of T and a
// binary search over T finds the threshold below which exactly nsample of them lie (the lanes hold the list in RM registers; a
// count is RM ballots); those are compacted to the head of the list, which bq_emit_list then orders.
template <int RM>
__device__ __forceinline__ int bq_select_smallest(int lane, int *hits, int cnt, int nsample, int np) {
    __builtin_amdgcn_wave_barrier();
    int key[RM];
#pragma unroll
    for (int r = 0; r < RM; ++r) key[r] = (r * 64 < cnt && r * 64 + lane < cnt) ? hits[r * 64 + lane] : 0x7fffffff;
    int lo = 0, hi = np;   // count(hits < lo) < nsample <= count(hits < hi)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        int c = 0;
#pragma unroll
        for (int r = 0; r < RM; ++r)
            if (r * 64 < cnt) c += (int)__popcll(__ballot(key[r] < mid));
        if (c >= nsample) hi = mid;
        else lo = mid;
    }
    __builtin_amdgcn_wave_barrier();
    int base = 0;
#pragma unroll
    for (int r = 0; r < RM; ++r)
        if (r * 64 < cnt) {
            const bool sel = key[r] < hi;
            const unsigned long long sm = __ballot(sel);
            if (sel) hits[base + popc_below(sm)] = key[r];
            base += (int)__popcll(sm);
        }
    return hi;   // every kept index is below, every dropped one at or above
}

// a crowded ball: every hit sets bit `original index` of an N-bit bitmap, which is then read back in order
template <int DPL>
__device__ __forceinline__ void bq_bitmap_search(int lane, int np, float r2, int nsample, float cx, float cy, float cz,
                                                 const float4 *__restrict__ sorted, const float *__restrict__ boxes,
                                                 const float *__restrict__ qboxes, int *__restrict__ out, unsigned *bits,
                                                 int *quads) {
#pragma unroll
    for (int w = 0; w < DPL; ++w) bits[lane * DPL + w] = 0u;
    __builtin_amdgcn_wave_barrier();
    bq_walk<DPL>(lane, np, r2, cx, cy, cz, sorted, boxes, qboxes, quads, [&](const float4 &p, float d2) {
        if (d2 < r2) {
            const int k = __float_as_int(p.w);
            atomicOr(&bits[k >> 5], 1u << (k & 31));
        }
    });
    __builtin_amdgcn_wave_barrier();
    // lane l owns bits [l*32*DPL, (l+1)*32*DPL)
    int bc = 0;
#pragma unroll 8
    for (int i = 0; i < DPL; ++i) bc += __popc(bits[lane * DPL + i]);
    const int incl = wave_inclusive_scan(bc);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    int pos = incl - bc;
    int mine_first = 0x7fffffff;
    if (bc > 0 && (pos < nsample || pos == 0)) {
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
    const unsigned long long have = __ballot(bc > 0);
    int first = 0;
    if (have) first = __builtin_amdgcn_readlane(mine_first, (int)__builtin_ctzll(have));
    for (int l = total + lane; l < nsample; l += 64) out[l] = first;
}

template <int K>
struct BqScales {
    float r2[K];   // radius * radius (ball_query_gpu.cu:23)
    int nsample[K];
    int *idx[K];   // (B, M, nsample[k])
};

// order (or NULL) / npc: the scene index of the CENTRES (npc entries per scene, the m real ones first): wave w then serves the
// w-th centre in that spatial order instead of centre w -- consecutive waves walk the same quads, buckets and rows, which are
// still in the CU's L1 / the XCD's L2 from the neighbour (FPS order scatters consecutive centres over the whole scene)
template <int DPL, int K>
__global__ __launch_bounds__(kQThreads) void bq_query_kernel(int np, int m, BqScales<K> sc,
                                                             const float *__restrict__ new_xyz,
                                                             const float4 *__restrict__ sorted,
                                                             const float *__restrict__ boxes,
                                                             const float *__restrict__ qboxes,
                                                             const float4 *__restrict__ order, int npc) {
    __shared__ unsigned s_bits[kQThreads / 64][64 * DPL];
    __shared__ int s_quads[kQThreads / 64][64];
    __shared__ int s_hits[kQThreads / 64][K][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int wg_x, bs;
    xcd_scene_map(wg_x, bs);   // a scene's index passes through one XCD's L2
    int ci = wg_x * (kQThreads / 64) + wave;
    if (ci >= m) return;  // wave-uniform; no block-level barrier below
    sorted += (size_t)bs * np;
    boxes += (size_t)bs * (np / 64) * 6;
    qboxes += (size_t)bs * (np / 256) * 6;
    float cx, cy, cz;
    if (order) {
        const float4 e = order[(size_t)bs * npc + ci];
        ci = __float_as_int(e.w);
        cx = e.x; cy = e.y; cz = e.z;
    } else {
        const float *c = new_xyz + ((size_t)bs * m + ci) * 3;
        cx = c[0]; cy = c[1]; cz = c[2];
    }
    float r2max = sc.r2[0];
#pragma unroll
    for (int k = 1; k < K; ++k) r2max = fmaxf(r2max, sc.r2[k]);
    int cnt[K];
#pragma unroll
    for (int k = 0; k < K; ++k) cnt[k] = 0;
    bq_walk<DPL>(lane, np, r2max, cx, cy, cz, sorted, boxes, qboxes, s_quads[wave], [&](const float4 &p, float d2) {
        if (!__ballot(d2 < r2max)) return;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const bool hit = d2 < sc.r2[k];
            const unsigned long long hm = __ballot(hit);
            if (hm) {
                const int pos = cnt[k] + popc_below(hm);
                if (hit && pos < 64) s_hits[wave][k][pos] = __float_as_int(p.w);
                cnt[k] += (int)__popcll(hm);
            }
        }
    });
#pragma unroll
    for (int k = 0; k < K; ++k) {
        int *out = sc.idx[k] + ((size_t)bs * m + ci) * sc.nsample[k];
        if (cnt[k] <= 64)
            bq_emit_list(lane, s_hits[wave][k], cnt[k], sc.nsample[k], out);
        else
            bq_bitmap_search<DPL>(lane, np, sc.r2[k], sc.nsample[k], cx, cy, cz, sorted, boxes, qboxes, out, s_bits[wave],
                                  s_quads[wave]);
    }
}

// TWO centres per wave. The search is a chain of dependent loads (quad boxes -> bucket boxes -> rows) and the chip
// holds at most 8 waves per SIMD, so a wave that walks the chain for two centres at once -- quad boxes loaded once
// and tested against both, the bucket boxes of centre 0 in lanes 0-31 and of centre 1 in lanes 32-63, one candidate
// row of each per step -- halves the latency per centre. Same results as bq_query_kernel.
// kStream (every nsample <= 64; the launcher takes it for scenes of more than 16384 points): no bitmap. A list that runs full
// during the walk is cut down to its nsample smallest indices on the spot and from then on only takes hits below the threshold
// of that cut (nothing at or above it can be among the first nsample in index order): ONE walk however crowded the ball, and
// 4 KB of LDS per wave instead of the N-bit bitmap's 8 KB at 65536 points, which held the kernel at half the waves a CU can
// keep (the walk is a chain of dependent loads: 0.54 -> 0.32 ms at 32 scenes of BASELINE config 5). The lists are looked at once
// per step of the walk (<= 2 rows = 128 appends per list).
template <int DPL, int K, bool kStream>
__global__ __launch_bounds__(kQThreads) void bq_query2_kernel(int np, int m, BqScales<K> sc,
                                                              const float *__restrict__ new_xyz,
                                                              const float4 *__restrict__ sorted,
                                                              const float *__restrict__ boxes,
                                                              const float *__restrict__ qboxes,
                                                              const float4 *__restrict__ order, int npc) {
    constexpr int CAP = kStream ? (K == 1 ? 512 : 256) : 64, kStepAppends = 128;
    constexpr int kBitWords = kStream ? 0 : 64 * DPL;
    __shared__ unsigned s_pool[kQThreads / 64][kBitWords + 2 * K * CAP];   // (the bitmap of a crowded ball,) the hit lists
    __shared__ int s_quads[kQThreads / 64][2][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int wg_x, bs;
    xcd_scene_map(wg_x, bs);   // a scene's index passes through one XCD's L2
    const int ci0 = (wg_x * (kQThreads / 64) + wave) * 2;
    if (ci0 >= m) return;  // wave-uniform; no block-level barrier below
    const bool two = ci0 + 1 < m;
    sorted += (size_t)bs * np;
    boxes += (size_t)bs * (np / 64) * 6;
    qboxes += (size_t)bs * (np / 256) * 6;
    float ax, ay, az, bx_, by_, bz_;
    int out0 = ci0, out1 = ci0 + 1;   // the rows of idx the two centres write
    if (order) {   // the two neighbours in the centres' spatial order (see bq_query_kernel): they mostly want the same rows
        const float4 e0 = order[(size_t)bs * npc + ci0], e1 = order[(size_t)bs * npc + ci0 + (two ? 1 : 0)];
        ax = e0.x; ay = e0.y; az = e0.z;
        bx_ = e1.x; by_ = e1.y; bz_ = e1.z;
        out0 = __float_as_int(e0.w);
        out1 = __float_as_int(e1.w);
    } else {
        const float *c0 = new_xyz + ((size_t)bs * m + ci0) * 3;
        const float *c1 = two ? c0 + 3 : c0;  // an odd tail repeats centre 0 (its second copy is never written)
        ax = c0[0]; ay = c0[1]; az = c0[2];
        bx_ = c1[0]; by_ = c1[1]; bz_ = c1[2];
    }
    const int half = lane >> 5;  // which centre this lane serves in the bucket-box stage
    const float hx = half ? bx_ : ax, hy = half ? by_ : ay, hz = half ? bz_ : az;
    float r2max = sc.r2[0];
#pragma unroll
    for (int k = 1; k < K; ++k) r2max = fmaxf(r2max, sc.r2[k]);
    int cnt[2][K];    // hits so far (kStream: entries in the list)
    int below[2][K];  // (kStream) the list only takes original indices below this
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int k = 0; k < K; ++k) {
            cnt[e][k] = 0;
            below[e][k] = 0x7fffffff;
        }
    int *hits_of = reinterpret_cast<int *>(s_pool[wave]) + kBitWords;
#ifdef EPNET_BQ_STATS
    const unsigned long long st_0 = __builtin_amdgcn_s_memtime();
    unsigned st_rows = 0;
    bool st_sel = false, st_again = false;
#endif

    auto visit = [&](int e, const float4 &p) {  // e is compile-time at the call sites
#ifdef EPNET_BQ_STATS
        ++st_rows;
#endif
        const float cx = e ? bx_ : ax, cy = e ? by_ : ay, cz = e ? bz_ : az;
        const float dx = cx - p.x, dy = cy - p.y, dz = cz - p.z;
        const float d2 = dx * dx + dy * dy + dz * dz;
        if (!__ballot(d2 < r2max)) return;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const bool hit = d2 < sc.r2[k] && (!kStream || __float_as_int(p.w) < below[e][k]);
            const unsigned long long hm = __ballot(hit);
            if (hm) {
                const int pos = cnt[e][k] + popc_below(hm);
                if (hit && pos < CAP) hits_of[(e * K + k) * CAP + pos] = __float_as_int(p.w);
                cnt[e][k] += (int)__popcll(hm);
            }
        }
    };
    // (kStream) between two steps of the walk: room for the next step's appends in every list
    auto make_room = [&]() {
        if constexpr (kStream) {
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int k = 0; k < K; ++k)
                    if (cnt[e][k] > CAP - kStepAppends) {
                        below[e][k] = bq_select_smallest<CAP / 64>(lane, hits_of + (e * K + k) * CAP, cnt[e][k], sc.nsample[k], np);
                        cnt[e][k] = sc.nsample[k];
#ifdef EPNET_BQ_STATS
                        st_sel = true;
#endif
                    }
        }
    };
    // one candidate row of each centre per step (both loads in flight together); bucket_of(e, bit)
    auto scan_pairs = [&](unsigned long long cand0, unsigned long long cand1, auto &&bucket_of) {
        while (cand0 | cand1) {
            make_room();
            float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), p1 = p0, p2 = p0, p3 = p0;
            const bool h0 = cand0 != 0ull, h1 = cand1 != 0ull;
            if (h0) { p0 = sorted[(bucket_of(0, (int)__builtin_ctzll(cand0)) << 6) + lane]; cand0 &= cand0 - 1ull; }
            if (h1) { p1 = sorted[(bucket_of(1, (int)__builtin_ctzll(cand1)) << 6) + lane]; cand1 &= cand1 - 1ull; }
            const bool h2 = cand0 != 0ull, h3 = cand1 != 0ull;
            if (h2) { p2 = sorted[(bucket_of(0, (int)__builtin_ctzll(cand0)) << 6) + lane]; cand0 &= cand0 - 1ull; }
            if (h3) { p3 = sorted[(bucket_of(1, (int)__builtin_ctzll(cand1)) << 6) + lane]; cand1 &= cand1 - 1ull; }
            if (h0) visit(0, p0);
            if (h1) visit(1, p1);
            if (h2) visit(0, p2);
            if (h3) visit(1, p3);
        }
    };
    if constexpr (DPL <= 2) {  // <= 64 buckets: every bucket box once, tested against both centres
        bool n0 = false, n1 = false;
        if (lane < (np >> 6)) {
            n0 = box_near(boxes + lane * 6, ax, ay, az, r2max);
            n1 = box_near(boxes + lane * 6, bx_, by_, bz_, r2max);
        }
        scan_pairs(__ballot(n0), __ballot(n1), [](int, int bit) { return bit; });
    } else {
        const int nq = np >> 8;
        for (int q0 = 0; q0 < nq; q0 += 64) {
            const int qd = q0 + lane;
            bool n0 = false, n1 = false;
            if (qd < nq) {
                n0 = box_near(qboxes + qd * 6, ax, ay, az, r2max);
                n1 = box_near(qboxes + qd * 6, bx_, by_, bz_, r2max);
            }
            const unsigned long long qm0 = __ballot(n0), qm1 = __ballot(n1);
            if (!(qm0 | qm1)) continue;
            const int nq0 = (int)__popcll(qm0), nq1 = (int)__popcll(qm1);
            if (n0) s_quads[wave][0][popc_below(qm0)] = qd;
            if (n1) s_quads[wave][1][popc_below(qm1)] = qd;
            __builtin_amdgcn_wave_barrier();
            const int nmine = half ? nq1 : nq0;
            for (int s0 = 0; s0 < max(nq0, nq1); s0 += 8) {  // 8 candidate quads of each centre per round
                const int slot = s0 + ((lane & 31) >> 2);
                int bid = 0;
                bool bnear = false;
                if (slot < nmine) {
                    bid = s_quads[wave][half][slot] * 4 + (lane & 3);
                    bnear = box_near(boxes + bid * 6, hx, hy, hz, r2max);
                }
                const unsigned long long cand = __ballot(bnear);
                scan_pairs(cand & 0xFFFFFFFFull, cand >> 32,
                           [&](int e, int bit) { return __builtin_amdgcn_readlane(bid, bit + 32 * e); });
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
#ifdef EPNET_BQ_STATS
    const unsigned long long st_1 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        if (e == 1 && !two) break;
        const float cx = e ? bx_ : ax, cy = e ? by_ : ay, cz = e ? bz_ : az;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            int *out = sc.idx[k] + ((size_t)bs * m + (e ? out1 : out0)) * sc.nsample[k];
            int *hits = hits_of + (e * K + k) * CAP;
            if constexpr (kStream) {
                if (cnt[e][k] > 64) {   // (a list that was cut before holds nsample or more candidates: the same selection)
                    bq_select_smallest<CAP / 64>(lane, hits, cnt[e][k], sc.nsample[k], np);
                    cnt[e][k] = sc.nsample[k];
#ifdef EPNET_BQ_STATS
                    st_sel = true;
#endif
                }
                bq_emit_list(lane, hits, cnt[e][k], sc.nsample[k], out);
            } else {
                if (cnt[e][k] <= 64) {
                    bq_emit_list(lane, hits, cnt[e][k], sc.nsample[k], out);
                } else {
#ifdef EPNET_BQ_STATS
                    st_again = true;
#endif
                    bq_bitmap_search<DPL>(lane, np, sc.r2[k], sc.nsample[k], cx, cy, cz, sorted, boxes, qboxes, out, s_pool[wave],
                                          s_quads[wave][0]);
                }
            }
        }
    }
#ifdef EPNET_BQ_STATS
    {
        const unsigned long long st_3 = __builtin_amdgcn_s_memtime();
        if (lane == 0) {
            const int o = st_again ? 8 : (st_sel ? 4 : 0);   // plain waves / waves that selected / waves that walked again
            atomicAdd(&g_bq_stats[o + 0], 1ull);
            atomicAdd(&g_bq_stats[o + 1], st_1 - st_0);
            atomicAdd(&g_bq_stats[o + 2], st_3 - st_1);
            atomicAdd(&g_bq_stats[o + 3], (unsigned long long)st_rows);
            atomicMax(&g_bq_stats[12], st_3 - st_0);
        }
    }
#endif
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


// shared with three_nn (interpolate.hip): cell-sorted float4 copy (x, y, z, original index) of n points padded
// to np (a multiple of 64) per scene, plus one box (6 floats) per 64 sorted points
int epnet::spatial_index_launch(int b, int n, int np, const float *xyz, float4 *sorted, float *boxes, float *qboxes,
                                hipStream_t s, int n_src, const int *src_idx, float *gathered) {
    if (n <= 4096) {
        constexpr int T = 256, cells = 1 << 12;
        // (+ the staging buffer of a quarter of the sorted scene, at most 512 points = 8 KB here; 64 KB below)
        const size_t stage = (np & 255) == 0 ? (size_t)(np / 4 < 512 ? np / 4 : 512) * sizeof(float4) : 0;
        hipLaunchKernelGGL((bq_index_kernel<T, 12>), dim3(b), dim3(T), (size_t)(cells + cells / (cells / T) + 64) * sizeof(int) + stage, s,
                           n, np, xyz, sorted, boxes, qboxes, n_src, src_idx, gathered);
    } else {
        constexpr int T = 1024, cells = 1 << 14;
        const size_t stage = (n <= T * 16 && (np & 255) == 0) ? (size_t)(np / 4) * sizeof(float4) : 0;   // (bigger scenes scatter straight to global memory)
        hipLaunchKernelGGL((bq_index_kernel<T, 14>), dim3(b), dim3(T), (size_t)(cells + cells / (cells / T) + 64) * sizeof(int) + stage, s,
                           n, np, xyz, sorted, boxes, qboxes, n_src, src_idx, gathered);
    }
    return check_launch("spatial index");
}

template <int K>
static int bq_query_launch(int b, int np, int m, const BqScales<K> &sc, const float *new_xyz, const float4 *sorted,
                           hipStream_t s, const float4 *order = nullptr, int npc = 0) {
    const float *boxes = (const float *)(sorted + (size_t)b * np);
    const float *qboxes = boxes + (size_t)b * (np / 64) * 6;
    // enough centres to fill the chip: two per wave (half the latency per centre); EPNET_BQ_PAIR=0/1 forces either
    const char *pair_str = getenv("EPNET_BQ_PAIR");
    const int pair_env = pair_str ? atoi(pair_str) : -1;
    const bool pair = pair_env >= 0 ? pair_env != 0 : (long long)b * m >= 65536;
    // EPNET_BQ_PAD_KB: LDS the launch reserves without using it -- fewer workgroups per CU, i.e. wave slots left for the short
    // one-workgroup-per-scene kernels of the sampling chain that run beside a full-chip query (an experiment knob; default 0)
    const char *pad_str = getenv("EPNET_BQ_PAD_KB");
    const size_t pad = pad_str ? (size_t)atoi(pad_str) * 1024 : 0;
    if (pair) {
        dim3 grid(div_up(div_up(m, 2), kQThreads / 64), b);
        // streaming lists instead of the bitmap (see the kernel): where the bitmap costs occupancy, i.e. above 16384 points;
        // EPNET_BQ_STREAM=0/1 forces either (nsample <= 64 is a precondition of the streaming variant)
        bool stream = np > 16384;
        if (const char *e = getenv("EPNET_BQ_STREAM")) stream = atoi(e) != 0;
        for (int k = 0; k < K; ++k) stream = stream && sc.nsample[k] <= 64;
#define EPNET_BQ2(D_)                                                                                                                   \
    do {                                                                                                                                \
        if (stream)                                                                                                                     \
            hipLaunchKernelGGL((bq_query2_kernel<D_, K, true>), grid, dim3(kQThreads), pad, s, np, m, sc, new_xyz, sorted, boxes, qboxes, \
                               order, npc);                                                                                             \
        else                                                                                                                            \
            hipLaunchKernelGGL((bq_query2_kernel<D_, K, false>), grid, dim3(kQThreads), pad, s, np, m, sc, new_xyz, sorted, boxes,       \
                               qboxes, order, npc);                                                                                     \
    } while (0)
        switch (np / 2048) {
            case 1: EPNET_BQ2(1); break;
            case 2: EPNET_BQ2(2); break;
            case 4: EPNET_BQ2(4); break;
            case 8: EPNET_BQ2(8); break;
            case 16: EPNET_BQ2(16); break;
            default: EPNET_BQ2(32); break;
        }
#undef EPNET_BQ2
        return check_launch("ball_query query");
    }
    dim3 grid(div_up(m, kQThreads / 64), b);
#define EPNET_BQ(D_) hipLaunchKernelGGL((bq_query_kernel<D_, K>), grid, dim3(kQThreads), pad, s, np, m, sc, new_xyz, sorted, boxes, qboxes, order, npc)
    switch (np / 2048) {
        case 1: EPNET_BQ(1); break;
        case 2: EPNET_BQ(2); break;
        case 4: EPNET_BQ(4); break;
        case 8: EPNET_BQ(8); break;
        case 16: EPNET_BQ(16); break;
        default: EPNET_BQ(32); break;
    }
#undef EPNET_BQ
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
    float *boxes = (float *)(sorted + (size_t)b * np);
    return spatial_index_launch(b, n, np, xyz, sorted, boxes, boxes + (size_t)b * (np / 64) * 6, (hipStream_t)stream);
}

// the scene index of the n points xyz_src[idx[0 .. n)] of every scene (rows of a cloud of n_src points: the centres a sampling has
// just picked), which are written to `gathered` (b, n, 3) in that order as well: gather_points + epnet_scene_index_build of an SA
// level's centres in one dispatch. 1024 <= n <= 16384.
extern "C" int epnet_scene_index_build_gathered(int b, int n_src, int n, const float *xyz_src, const int *idx, float *gathered,
                                                void *index, size_t index_bytes, epnet_stream_t stream) {
    const size_t need = scene_index_bytes(b, n);
    EPNET_REQUIRE(need != 0 && n <= 16384 && n_src >= 1 && xyz_src && idx && gathered && index);
    if (index_bytes < need) return EPNET_ENOMEM;
    if (((uintptr_t)index & 15) != 0) return EPNET_EINVAL;
    EPNET_REQUIRE(b <= 65535);
    const int np = scene_index_np(n);
    float4 *sorted = (float4 *)index;
    float *boxes = (float *)(sorted + (size_t)b * np);
    return spatial_index_launch(b, n, np, xyz_src, sorted, boxes, boxes + (size_t)b * (np / 64) * 6, (hipStream_t)stream, n_src, idx,
                                gathered);
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
    BqScales<1> sc;
    sc.r2[0] = radius * radius;  // ball_query_gpu.cu:23
    sc.nsample[0] = nsample;
    sc.idx[0] = idx;
    return bq_query_launch<1>(b, np, m, sc, new_xyz, (const float4 *)index, (hipStream_t)stream);
}

// the nscales ball queries of an MSG level (same centres, same points) in one launch: boxes are tested against the
// largest radius and every distance is computed once. idx[k]: (b, m, nsample[k]). Same results as nscales calls.
extern "C" int epnet_ball_query_indexed_multi(int b, int n, int m, int nscales, const float *radii, const int *nsamples,
                                              const float *new_xyz, const float *xyz, const void *index,
                                              size_t index_bytes, int *const *idx, epnet_stream_t stream) {
    EPNET_REQUIRE(nscales >= 0 && (nscales == 0 || (radii && nsamples && idx)));
    const size_t need = scene_index_bytes(b, n);
    bool fused = nscales == 2 && need != 0 && index && m > 0 && nsamples[0] > 0 && nsamples[1] > 0;
    if (!fused) {
        for (int k = 0; k < nscales; ++k) {
            const int rc = epnet_ball_query_indexed(b, n, m, radii[k], nsamples[k], new_xyz, xyz, index, index_bytes, idx[k], stream);
            if (rc) return rc;
        }
        return EPNET_OK;
    }
    EPNET_REQUIRE(new_xyz && idx[0] && idx[1]);
    if (index_bytes < need) return EPNET_ENOMEM;
    EPNET_REQUIRE(b <= 65535);
    BqScales<2> sc;
    for (int k = 0; k < 2; ++k) {
        sc.r2[k] = radii[k] * radii[k];  // ball_query_gpu.cu:23
        sc.nsample[k] = nsamples[k];
        sc.idx[k] = idx[k];
    }
    return bq_query_launch<2>(b, scene_index_np(n), m, sc, new_xyz, (const float4 *)index, (hipStream_t)stream);
}

// the nscales (1 or 2) ball queries of a level with the centres served in THEIR spatial order: centre_index is the scene index of
// the m centres (new_xyz = the cloud it was built from, in that order). Same results as epnet_ball_query_indexed_multi.
extern "C" int epnet_ball_query_ordered(int b, int n, int m, int nscales, const float *radii, const int *nsamples,
                                        const float *new_xyz, const float *xyz, const void *index, size_t index_bytes,
                                        const void *centre_index, size_t centre_index_bytes, int *const *idx,
                                        epnet_stream_t stream) {
    EPNET_REQUIRE(nscales >= 0 && (nscales == 0 || (radii && nsamples && idx)));
    const size_t need = scene_index_bytes(b, n), need_c = scene_index_bytes(b, m);
    bool ordered = (nscales == 1 || nscales == 2) && need != 0 && need_c != 0 && index && centre_index && m > 0;
    for (int k = 0; ordered && k < nscales; ++k) ordered = nsamples[k] > 0 && idx[k] != nullptr;
    // what the order buys is cache hits on the point rows: nothing up to 16384 points (the scene's rows stay in L2 whatever the order:
    // 256 scenes, 16384 x 4096: 0.458 = 0.455 ms; 4096 x 1024: 0.108 -> 0.125), 31 % at 65536 (6.68 -> 4.58 ms). EPNET_BQ_ORDERED=0/1 forces.
    const char *force = getenv("EPNET_BQ_ORDERED");
    if (force ? atoi(force) == 0 : n <= 16384) ordered = false;
    if (!ordered) return epnet_ball_query_indexed_multi(b, n, m, nscales, radii, nsamples, new_xyz, xyz, index, index_bytes, idx, stream);
    if (index_bytes < need || centre_index_bytes < need_c) return EPNET_ENOMEM;
    EPNET_REQUIRE(b <= 65535);
    const int np = scene_index_np(n), npc = scene_index_np(m);
    if (nscales == 1) {
        BqScales<1> sc;
        sc.r2[0] = radii[0] * radii[0];  // ball_query_gpu.cu:23
        sc.nsample[0] = nsamples[0];
        sc.idx[0] = idx[0];
        return bq_query_launch<1>(b, np, m, sc, new_xyz, (const float4 *)index, (hipStream_t)stream, (const float4 *)centre_index, npc);
    }
    BqScales<2> sc;
    for (int k = 0; k < 2; ++k) {
        sc.r2[k] = radii[k] * radii[k];
        sc.nsample[k] = nsamples[k];
        sc.idx[k] = idx[k];
    }
    return bq_query_launch<2>(b, np, m, sc, new_xyz, (const float4 *)index, (hipStream_t)stream, (const float4 *)centre_index, npc);
}

extern "C" size_t epnet_ball_query_workspace_bytes(int b, int n, int m) {
    if (n < 1024 || m <= 0) return 0;  // small or huge scenes: direct scan, no scratch
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

#ifdef EPNET_BQ_STATS
extern "C" int epnet_debug_bq_stats(unsigned long long *host16) {
    (void)hipMemcpyFromSymbol(host16, HIP_SYMBOL(epnet::g_bq_stats), sizeof(unsigned long long) * 16);
    unsigned long long zero[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(epnet::g_bq_stats), zero, sizeof(zero));
    return 0;
}
#endif
#ifdef EPNET_IX_STATS
extern "C" int epnet_debug_ix_stats(unsigned long long *host8) {
    (void)hipMemcpyFromSymbol(host8, HIP_SYMBOL(epnet::g_ix_stats), sizeof(unsigned long long) * 8);
    unsigned long long zero[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(epnet::g_ix_stats), zero, sizeof(zero));
    return 0;
}
#endif
