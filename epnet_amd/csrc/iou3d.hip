// iou3d.hip -- rotated-box BEV overlap / IoU and NMS (rotated and axis-aligned) for gfx950.
//
// Replaces lib/utils/iou3d/src/iou3d_kernel.cu (box_overlap :108-212, iou_bev :214-221,
// boxes_overlap_kernel :223-234, boxes_iou_bev_kernel :236-248, nms_kernel :250-292, iou_normal
// :295-303, nms_normal_kernel :306-348) and the host half of lib/utils/iou3d/src/iou3d.cpp:73-170.
//
// Differences in structure, not in results:
//   * box_overlap evaluates each polygon vertex's atan2 once and bubble-sorts on the stored
//     angles (the reference recomputes two atan2 per comparison, :104-106, :188-196): the same
//     deterministic function of the same arguments, hence the same swaps in the same order;
//   * the NMS bit-mask is computed for the upper triangle only, one wave per (row, 64-column tile) -- the host sweep
//     (iou3d.cpp:111-114) never reads a word left of the diagonal;
//   * the greedy sweep runs on the device (one workgroup): no cudaMalloc / 5 MB device->host
//     copy / host loop per call (iou3d.cpp:87-116).
// Trigonometry: correctly rounded float cos/sin/atan2 via the double-precision functions, the
// same definition the oracle uses (DESIGN.md "Parity definition").
#include <math.h>

#include "common.h"

namespace epnet {

constexpr float kIouEps = 1e-8f;  // iou3d_kernel.cu:13

__device__ __forceinline__ float cr_cosf(float x) { return (float)cos((double)x); }
__device__ __forceinline__ float cr_sinf(float x) { return (float)sin((double)x); }
__device__ __forceinline__ float cr_atan2f(float y, float x) { return (float)atan2((double)y, (double)x); }

struct P2 {
    float x, y;
};

__device__ __forceinline__ float cross3(P2 p1, P2 p2, P2 p0) {  // :38-40
    return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

__device__ __forceinline__ bool rect_cross(P2 p1, P2 p2, P2 q1, P2 q2) {  // :42-48
    return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
           fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}

// check_in_box2d (:50-65) with the box's cos(-ry), sin(-ry) passed in
__device__ __forceinline__ bool in_box2d(const float *box, float ncos, float nsin, P2 p) {
    const float MARGIN = 1e-5f;
    const float center_x = (box[0] + box[2]) / 2;
    const float center_y = (box[1] + box[3]) / 2;
    const float rot_x = (p.x - center_x) * ncos + (p.y - center_y) * nsin + center_x;
    const float rot_y = -(p.x - center_x) * nsin + (p.y - center_y) * ncos + center_y;
    return rot_x > box[0] - MARGIN && rot_x < box[2] + MARGIN && rot_y > box[1] - MARGIN && rot_y < box[3] + MARGIN;
}

__device__ __forceinline__ bool seg_intersection(P2 p1, P2 p0, P2 q1, P2 q0, P2 &ans) {  // :67-96
    if (!rect_cross(p0, p1, q0, q1)) return false;
    const float s1 = cross3(q0, p1, p0);
    const float s2 = cross3(p1, q1, p0);
    const float s3 = cross3(p0, q1, q0);
    const float s4 = cross3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
    const float s5 = cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > kIouEps) {
        ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        const float D = a0 * b1 - a1 * b0;
        ans.x = (b0 * c1 - b1 * c0) / D;
        ans.y = (a1 * c0 - a0 * c1) / D;
    }
    return true;
}

__device__ __forceinline__ void rot_center(P2 center, float c, float s, P2 &p) {  // :98-102
    const float nx = (p.x - center.x) * c + (p.y - center.y) * s + center.x;
    const float ny = -(p.x - center.x) * s + (p.y - center.y) * c + center.y;
    p.x = nx;
    p.y = ny;
}

// per-box trigonometry: cos/sin of +ry (corner rotation) and of -ry (point-in-box test)
struct BoxTrig {
    float c, s, nc, ns;
};

__device__ __forceinline__ BoxTrig box_trig(float ry) {
    BoxTrig t;
    t.c = cr_cosf(ry);
    t.s = cr_sinf(ry);
    t.nc = cr_cosf(-ry);
    t.ns = cr_sinf(-ry);
    return t;
}

__device__ float box_overlap(const float *box_a, const float *box_b, BoxTrig ta, BoxTrig tb) {  // :108-212
    const float a_x1 = box_a[0], a_y1 = box_a[1], a_x2 = box_a[2], a_y2 = box_a[3];
    const float b_x1 = box_b[0], b_y1 = box_b[1], b_x2 = box_b[2], b_y2 = box_b[3];
    const P2 center_a = {(a_x1 + a_x2) / 2, (a_y1 + a_y2) / 2};
    const P2 center_b = {(b_x1 + b_x2) / 2, (b_y1 + b_y2) / 2};
    P2 ac[5] = {{a_x1, a_y1}, {a_x2, a_y1}, {a_x2, a_y2}, {a_x1, a_y2}, {0.f, 0.f}};
    P2 bc[5] = {{b_x1, b_y1}, {b_x2, b_y1}, {b_x2, b_y2}, {b_x1, b_y2}, {0.f, 0.f}};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        rot_center(center_a, ta.c, ta.s, ac[k]);
        rot_center(center_b, tb.c, tb.s, bc[k]);
    }
    ac[4] = ac[0];
    bc[4] = bc[0];

    P2 cp[24];  // the reference has 16 slots; two rectangles cannot produce more than 16 entries
    float ang[24];
    P2 pc = {0.f, 0.f};
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            P2 r;
            if (seg_intersection(ac[i + 1], ac[i], bc[j + 1], bc[j], r)) {
                pc.x = pc.x + r.x;
                pc.y = pc.y + r.y;
                cp[cnt++] = r;
            }
        }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (in_box2d(box_a, ta.nc, ta.ns, bc[k])) {
            pc.x = pc.x + bc[k].x;
            pc.y = pc.y + bc[k].y;
            cp[cnt++] = bc[k];
        }
        if (in_box2d(box_b, tb.nc, tb.ns, ac[k])) {
            pc.x = pc.x + ac[k].x;
            pc.y = pc.y + ac[k].y;
            cp[cnt++] = ac[k];
        }
    }
    pc.x /= cnt;
    pc.y /= cnt;

    for (int i = 0; i < cnt; ++i) ang[i] = cr_atan2f(cp[i].y - pc.y, cp[i].x - pc.x);
    for (int j = 0; j < cnt - 1; ++j)
        for (int i = 0; i < cnt - j - 1; ++i)
            if (ang[i] > ang[i + 1]) {
                const P2 tp = cp[i];
                cp[i] = cp[i + 1];
                cp[i + 1] = tp;
                const float ta_ = ang[i];
                ang[i] = ang[i + 1];
                ang[i + 1] = ta_;
            }

    float area = 0.f;
    for (int k = 0; k < cnt - 1; ++k) {
        const float ax = cp[k].x - cp[0].x, ay = cp[k].y - cp[0].y;
        const float bx = cp[k + 1].x - cp[0].x, by = cp[k + 1].y - cp[0].y;
        area += ax * by - ay * bx;
    }
    return fabsf(area) / 2.0f;
}

__device__ __forceinline__ float iou_bev(const float *box_a, const float *box_b, BoxTrig ta, BoxTrig tb) {  // :214-221
    const float sa = (box_a[2] - box_a[0]) * (box_a[3] - box_a[1]);
    const float sb = (box_b[2] - box_b[0]) * (box_b[3] - box_b[1]);
    const float s_overlap = box_overlap(box_a, box_b, ta, tb);
    return s_overlap / fmaxf(sa + sb - s_overlap, kIouEps);
}

__device__ __forceinline__ float iou_normal(const float *a, const float *b) {  // :295-303
    const float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
    const float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
    const float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
    const float interS = width * height;
    const float Sa = (a[2] - a[0]) * (a[3] - a[1]);
    const float Sb = (b[2] - b[0]) * (b[3] - b[1]);
    return interS / fmaxf(Sa + Sb - interS, kIouEps);
}

// one thread per (a, b) pair; 64 b's x 4 a's per workgroup, the b boxes' trig staged in LDS
template <bool IOU>
__global__ __launch_bounds__(256) void pairwise_bev_kernel(int num_a, const float *__restrict__ boxes_a, int num_b,
                                                           const float *__restrict__ boxes_b, float *__restrict__ ans) {
    const int b_idx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int a_idx = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (a_idx >= num_a || b_idx >= num_b) return;
    float ba[5], bb[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        ba[i] = boxes_a[a_idx * 5 + i];
        bb[i] = boxes_b[b_idx * 5 + i];
    }
    const BoxTrig ta = box_trig(ba[4]), tb = box_trig(bb[4]);
    ans[(size_t)a_idx * num_b + b_idx] = IOU ? iou_bev(ba, bb, ta, tb) : box_overlap(ba, bb, ta, tb);
}

// ---- fused 3-D IoU (boxes_iou3d_gpu of lib/utils/iou3d/iou3d_utils.py:21-53 in ONE launch) -------------
// The reference builds the BEV boxes (kitti_utils.boxes3d_to_bev_torch :137-150), calls the overlap kernel and
// finishes with ~10 small torch kernels (height overlap, volumes, clamp, divide). Every step is an fp32
// elementwise op, reproduced here in the same order, so the values equal that composition.
// boxes: (.,7) [x, y, z, h, w, l, ry], y = bottom centre, camera coordinates.
__device__ __forceinline__ void bev_of(const float *b7, float *bev) {
    const float half_l = b7[5] / 2, half_w = b7[4] / 2;
    bev[0] = b7[0] - half_l;
    bev[1] = b7[2] - half_w;
    bev[2] = b7[0] + half_l;
    bev[3] = b7[2] + half_w;
    bev[4] = b7[6];
}

__device__ __forceinline__ float iou3d_pair(const float *a7, const float *b7) {
    float ba[5], bb[5];
    bev_of(a7, ba);
    bev_of(b7, bb);
    const float overlaps_bev = box_overlap(ba, bb, box_trig(ba[4]), box_trig(bb[4]));
    const float a_min = a7[1] - a7[3], a_max = a7[1], b_min = b7[1] - b7[3], b_max = b7[1];
    const float max_of_min = fmaxf(a_min, b_min), min_of_max = fminf(a_max, b_max);
    const float overlaps_h = fmaxf(min_of_max - max_of_min, 0.f);
    const float overlaps_3d = overlaps_bev * overlaps_h;
    const float vol_a = a7[3] * a7[4] * a7[5], vol_b = b7[3] * b7[4] * b7[5];
    return overlaps_3d / fmaxf(vol_a + vol_b - overlaps_3d, 1e-7f);
}

__global__ __launch_bounds__(256) void iou3d_matrix_kernel(int num_a, const float *__restrict__ boxes_a, int num_b,
                                                           const float *__restrict__ boxes_b, float *__restrict__ ans) {
    const int b_idx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int a_idx = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (a_idx >= num_a || b_idx >= num_b) return;
    float a7[7], b7[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        a7[i] = boxes_a[a_idx * 7 + i];
        b7[i] = boxes_b[b_idx * 7 + i];
    }
    ans[(size_t)a_idx * num_b + b_idx] = iou3d_pair(a7, b7);
}

__global__ __launch_bounds__(256) void iou3d_pairs_kernel(int k, const float *__restrict__ boxes_a,
                                                          const float *__restrict__ boxes_b, float *__restrict__ ans) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= k) return;
    float a7[7], b7[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        a7[j] = boxes_a[i * 7 + j];
        b7[j] = boxes_b[i * 7 + j];
    }
    ans[i] = iou3d_pair(a7, b7);
}

// ---- ROI augmentation by noise (lib/rpn/proposal_target_layer.py:220-247, aug_roi_by_noise_torch) ----------
// The reference walks the sampled ROIs one by one on the host: per ROI up to aug_times tries, each a host coin
// (np.random.rand() < 0.2 keeps the ROI as it is), a noisy box built by five small torch kernels
// (random_aug_box3d :250-275), a single-pair boxes_iou3d_gpu (another ~15 launches) and a device->host read of the
// IoU for the loop condition -- up to 640 such round trips per scene. Here the random draws of ALL tries are
// the caller's tables (drawn on the device with no sync), every (ROI, try) IoU is evaluated by its own lane, and
// the first try whose IoU ends the reference's loop (`not (iou < pos_thresh)`) is found with one ballot:
// the same boxes and IoUs as the loop fed with draw [k][cnt] at try cnt of ROI k.
//   keep_draw (k, aug_times) u8: 1 = "keep the original ROI" (:232-234)
//   noise (k, aug_times, 7) f32: pos_shift[3], hwl_scale[3], angle_rot -- aug = [xyz + shift, hwl * scale, ry + rot] (:259,274)
__device__ __forceinline__ void aug_box_of(const float *roi, const float *nz, bool keep, float *aug) {
#pragma unroll
    for (int j = 0; j < 7; ++j) aug[j] = roi[j];
    if (!keep) {
#pragma unroll
        for (int j = 0; j < 3; ++j) aug[j] = roi[j] + nz[j];
#pragma unroll
        for (int j = 3; j < 6; ++j) aug[j] = roi[j] * nz[j];
        aug[6] = roi[6] + nz[6];
    }
}

// TP = tries per ROI rounded up to a power of two (<= 64): 64 / TP ROIs per wave, lane = (roi, try)
__global__ __launch_bounds__(256) void aug_roi_wave_kernel(int k, int aug_times, int tp, float pos_thresh,
                                                           float *__restrict__ rois, const float *__restrict__ gts,
                                                           const float *__restrict__ iou_src, const int *__restrict__ tries,
                                                           const unsigned char *__restrict__ keep_draw,
                                                           const float *__restrict__ noise, float *__restrict__ iou_out) {
    const int lane = lane_id();
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int per_wave = 64 / tp;
    const int grp = lane / tp, t = lane - grp * tp;
    const int r = wave * per_wave + grp;
    const int n_try = r < k ? (tries ? min(max(tries[r], 0), aug_times) : aug_times) : 0;
    const bool active = t < n_try;
    float roi[7], gt[7], aug[7], nz[7];
    bool keep = true;
    float iou = 0.f;
    if (active) {
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            roi[j] = rois[(size_t)r * 7 + j];
            gt[j] = gts[(size_t)r * 7 + j];
            nz[j] = noise[((size_t)r * aug_times + t) * 7 + j];
        }
        keep = keep_draw[(size_t)r * aug_times + t] != 0;
        aug_box_of(roi, nz, keep, aug);
        iou = iou3d_pair(aug, gt);
    }
    // the loop runs try t+1 only while temp_iou < pos_thresh (:231): a try ends it when that comparison is false
    const unsigned long long ends = __ballot(active && !(iou < pos_thresh));
    const unsigned long long group_mask = (tp == 64 ? ~0ull : ((1ull << tp) - 1)) << (grp * tp);
    const unsigned long long mine = ends & group_mask;
    const int last = mine ? (__builtin_ctzll(mine) - grp * tp) : (n_try - 1);
    __syncthreads();  // every lane has read its ROI before the winners overwrite rows (waves of a block share no ROI,
                      // lanes of a wave are in lockstep; the barrier only orders the stores after the loads)
    if (active && t == last) {
#pragma unroll
        for (int j = 0; j < 7; ++j) rois[(size_t)r * 7 + j] = aug[j];
        iou_out[r] = keep ? iou_src[r] : iou;  // :243-246
    }
    if (r < k && n_try == 0 && t == 0) iou_out[r] = iou_src[r];  // no try ran (cnt == 0, :243): box untouched
}

// any number of tries: one thread walks the loop of one ROI
__global__ __launch_bounds__(64) void aug_roi_serial_kernel(int k, int aug_times, float pos_thresh, float *__restrict__ rois,
                                                            const float *__restrict__ gts, const float *__restrict__ iou_src,
                                                            const int *__restrict__ tries,
                                                            const unsigned char *__restrict__ keep_draw,
                                                            const float *__restrict__ noise, float *__restrict__ iou_out) {
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= k) return;
    const int n_try = tries ? min(max(tries[r], 0), aug_times) : aug_times;
    float roi[7], gt[7], aug[7], nz[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        roi[j] = rois[(size_t)r * 7 + j];
        gt[j] = gts[(size_t)r * 7 + j];
        aug[j] = roi[j];
    }
    float temp_iou = 0.f;
    int cnt = 0;
    bool keep = true;
    while (temp_iou < pos_thresh && cnt < n_try) {
#pragma unroll
        for (int j = 0; j < 7; ++j) nz[j] = noise[((size_t)r * aug_times + cnt) * 7 + j];
        keep = keep_draw[(size_t)r * aug_times + cnt] != 0;
        aug_box_of(roi, nz, keep, aug);
        temp_iou = iou3d_pair(aug, gt);
        ++cnt;
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) rois[(size_t)r * 7 + j] = aug[j];
    iou_out[r] = (cnt == 0 || keep) ? iou_src[r] : temp_iou;
}

// per-box trigonometry, once per box instead of once per pair
// Batched form of all three NMS kernels: group = blockIdx.z (y for the 1-D ones) with `cap` boxes of room per group and
// the group's box count read from device memory (counts[group], clamped to cap) -- a caller whose counts are produced
// on the device (the proposal layer) never has to read them back. counts == NULL: one group of boxes_num boxes.
__global__ __launch_bounds__(256) void box_trig_kernel(int n, const int *__restrict__ counts, int cap,
                                                       const float *__restrict__ boxes, BoxTrig *__restrict__ trig) {
    const int grp = blockIdx.y;
    if (counts) n = min(counts[grp], cap);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) trig[(size_t)grp * cap + i] = box_trig(boxes[((size_t)grp * cap + i) * 5 + 4]);
}

// ---- rotated NMS mask (nms_kernel, iou3d_kernel.cu:250-292) ------------------------------------------------------
// Everything of a pair's overlap that depends on ONE box is computed once per box: the rotated corners (same expressions
// as box_overlap), their axis-aligned hull, cos / sin of -ry for the point-in-box test, the box area.
struct BoxRot {
    float x1, y1, x2, y2;      // the BEV box itself
    float nc, ns;              // cos(-ry), sin(-ry): check_in_box2d (:50-65)
    float cx[4], cy[4];        // corners after rotate_around_center (:98-102, :141-146)
    float minx, maxx, miny, maxy;
    float area, pad;
};
static_assert(sizeof(BoxRot) == 80, "BoxRot layout");

__global__ __launch_bounds__(256) void box_rot_kernel(int n, const int *__restrict__ counts, int cap,
                                                      const float *__restrict__ boxes, BoxRot *__restrict__ rot) {
    const int grp = blockIdx.y;
    if (counts) n = min(counts[grp], cap);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *b = boxes + ((size_t)grp * cap + i) * 5;
    BoxRot r;
    r.x1 = b[0]; r.y1 = b[1]; r.x2 = b[2]; r.y2 = b[3];
    const BoxTrig t = box_trig(b[4]);
    r.nc = t.nc;
    r.ns = t.ns;
    const P2 center = {(r.x1 + r.x2) / 2, (r.y1 + r.y2) / 2};
    P2 c[4] = {{r.x1, r.y1}, {r.x2, r.y1}, {r.x2, r.y2}, {r.x1, r.y2}};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        rot_center(center, t.c, t.s, c[k]);
        r.cx[k] = c[k].x;
        r.cy[k] = c[k].y;
    }
    r.minx = fminf(fminf(c[0].x, c[1].x), fminf(c[2].x, c[3].x));
    r.maxx = fmaxf(fmaxf(c[0].x, c[1].x), fmaxf(c[2].x, c[3].x));
    r.miny = fminf(fminf(c[0].y, c[1].y), fminf(c[2].y, c[3].y));
    r.maxy = fmaxf(fmaxf(c[0].y, c[1].y), fmaxf(c[2].y, c[3].y));
    r.area = (r.x2 - r.x1) * (r.y2 - r.y1);
    r.pad = 0.f;
    rot[(size_t)grp * cap + i] = r;
}

// box_overlap (:108-212) over precomputed corners, the candidate points of the intersection polygon kept in LDS
// ([slot][thread]: per-lane dynamic slots without scratch memory, conflict-free) instead of per-thread arrays
constexpr int kRotThreads = 256;
constexpr int kRotSlots = 16;  // the reference's own capacity (cross_points[16], :125)
__device__ float box_overlap_lds(const BoxRot &A, const BoxRot &B, float *s_x, float *s_y, float *s_a) {
    const int t = threadIdx.x;
    P2 ac[5], bc[5];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        ac[k] = {A.cx[k], A.cy[k]};
        bc[k] = {B.cx[k], B.cy[k]};
    }
    ac[4] = ac[0];
    bc[4] = bc[0];
    P2 pc = {0.f, 0.f};
    int cnt = 0;
    auto push = [&](P2 r) {
        pc.x = pc.x + r.x;
        pc.y = pc.y + r.y;
        if (cnt < kRotSlots) {
            s_x[cnt * kRotThreads + t] = r.x;
            s_y[cnt * kRotThreads + t] = r.y;
        }
        ++cnt;
    };
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            P2 r;
            if (seg_intersection(ac[i + 1], ac[i], bc[j + 1], bc[j], r)) push(r);
        }
    const float box_a[4] = {A.x1, A.y1, A.x2, A.y2}, box_b[4] = {B.x1, B.y1, B.x2, B.y2};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (in_box2d(box_a, A.nc, A.ns, bc[k])) push(bc[k]);
        if (in_box2d(box_b, B.nc, B.ns, ac[k])) push(ac[k]);
    }
    pc.x /= cnt;
    pc.y /= cnt;
    cnt = min(cnt, kRotSlots);
    for (int i = 0; i < cnt; ++i)
        s_a[i * kRotThreads + t] = cr_atan2f(s_y[i * kRotThreads + t] - pc.y, s_x[i * kRotThreads + t] - pc.x);
    // the reference's bubble sort (:183-192): adjacent swaps on a strict '>' -- stable, so equal angles keep their order
    for (int j = 0; j < cnt - 1; ++j)
        for (int i = 0; i < cnt - j - 1; ++i) {
            const float a0 = s_a[i * kRotThreads + t], a1 = s_a[(i + 1) * kRotThreads + t];
            if (a0 > a1) {
                const float x0 = s_x[i * kRotThreads + t], y0 = s_y[i * kRotThreads + t];
                s_x[i * kRotThreads + t] = s_x[(i + 1) * kRotThreads + t];
                s_y[i * kRotThreads + t] = s_y[(i + 1) * kRotThreads + t];
                s_a[i * kRotThreads + t] = a1;
                s_x[(i + 1) * kRotThreads + t] = x0;
                s_y[(i + 1) * kRotThreads + t] = y0;
                s_a[(i + 1) * kRotThreads + t] = a0;
            }
        }
    float area = 0.f;
    const float x0 = s_x[t], y0 = s_y[t];
    for (int k = 0; k < cnt - 1; ++k) {
        const float ax = s_x[k * kRotThreads + t] - x0, ay = s_y[k * kRotThreads + t] - y0;
        const float bx = s_x[(k + 1) * kRotThreads + t] - x0, by = s_y[(k + 1) * kRotThreads + t] - y0;
        area += ax * by - ay * bx;
    }
    return fabsf(area) / 2.0f;
}

// One workgroup per 64 x 64 tile of the upper triangle. Phase 1: every thread tests 16 pairs against the axis-aligned hulls
// of the rotated boxes -- hulls more than 1e-3 apart cannot share a segment crossing (check_rect_cross needs overlapping
// segment boxes) or an inside corner (margin 1e-5), so the reference's cnt is 0 and its overlap exactly 0 -- and queues the
// survivors (a few per cent of the pairs of a proposal set). Phase 2: the queue is worked off one pair per lane, dense,
// and the suppression bits are OR-ed into the tile's 64 mask words in LDS. 6300 boxes: 0.95 -> ... ms.
__global__ __launch_bounds__(kRotThreads) void nms_mask_rot_kernel(int boxes_num, const int *__restrict__ counts, int cap,
                                                                    float thresh, const BoxRot *__restrict__ rot,
                                                                    unsigned long long *__restrict__ mask) {
    __shared__ BoxRot s_row[64], s_col[64];
    __shared__ unsigned long long s_bits[64];
    __shared__ unsigned short s_queue[64 * 64];
    __shared__ int s_count;
    __shared__ float s_px[kRotSlots * kRotThreads], s_py[kRotSlots * kRotThreads], s_pa[kRotSlots * kRotThreads];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int row_blk = blockIdx.x, col_blk = blockIdx.y, grp = blockIdx.z;
    if (counts) boxes_num = min(counts[grp], cap);
    const int stride = counts ? (cap + 63) / 64 : (boxes_num + 63) / 64;  // mask words per row
    if (col_blk < row_blk || row_blk * 64 >= boxes_num || col_blk * 64 >= boxes_num) return;  // tiles left of the diagonal are never read
    rot += (size_t)grp * cap;
    mask += (size_t)grp * cap * stride;
    {   // the tile's 128 boxes: 20 floats each, copied as floats (coalesced)
        const float *src_r = reinterpret_cast<const float *>(rot + (size_t)row_blk * 64);
        const float *src_c = reinterpret_cast<const float *>(rot + (size_t)col_blk * 64);
        const int nr = min(64, boxes_num - row_blk * 64) * 20, nc = min(64, boxes_num - col_blk * 64) * 20;
        for (int e = t; e < nr; e += kRotThreads) reinterpret_cast<float *>(s_row)[e] = src_r[e];
        for (int e = t; e < nc; e += kRotThreads) reinterpret_cast<float *>(s_col)[e] = src_c[e];
    }
    if (t < 64) s_bits[t] = 0ull;
    if (t == 0) s_count = 0;
    __syncthreads();
    const float kSlack = 1e-3f;
    const int col = col_blk * 64 + lane;
    const BoxRot &cb = s_col[lane];
    for (int rr = wave; rr < 64; rr += kRotThreads / 64) {  // wave-uniform row: ballot + prefix give the queue slots
        const int row = row_blk * 64 + rr;
        bool cand = false;
        if (row < boxes_num && col < boxes_num && col > row) {  // on the diagonal tile only the bits right of the row itself (:281-283)
            const BoxRot &rb = s_row[rr];
            cand = !(rb.minx > cb.maxx + kSlack || cb.minx > rb.maxx + kSlack || rb.miny > cb.maxy + kSlack || cb.miny > rb.maxy + kSlack);
        }
        const unsigned long long m = __ballot(cand);
        if (m) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_count, (int)__builtin_popcountll(m));
            base = __builtin_amdgcn_readfirstlane(base);
            if (cand) s_queue[base + popc_below(m)] = (unsigned short)((rr << 6) | lane);
        }
    }
    __syncthreads();
    const int total = s_count;
    for (int k = t; k < total; k += kRotThreads) {
        const unsigned e = s_queue[k];
        const int rr = (int)(e >> 6), cl = (int)(e & 63u);
        const BoxRot &A = s_row[rr], &B = s_col[cl];   // box_a = the row box (cur_box), box_b = the column box (:285-287)
        const float s_overlap = box_overlap_lds(A, B, s_px, s_py, s_pa);
        const float v = s_overlap / fmaxf(A.area + B.area - s_overlap, kIouEps);   // iou_bev (:214-221)
        if (v > thresh) atomicOr(&s_bits[rr], 1ull << cl);
    }
    __syncthreads();
    if (t < 64) {
        const int row = row_blk * 64 + t;
        if (row < boxes_num) mask[(size_t)row * stride + col_blk] = s_bits[t];
    }
}

// suppression bit-mask, upper triangle only: one WAVE per (row, 64-column tile) -- each lane owns one
// pair and the 64-bit mask word is the wave's ballot (the reference gives a thread a whole row of 64
// pairs, :281-290)
template <bool ROTATED>
__global__ __launch_bounds__(256) void nms_mask_kernel(int boxes_num, const int *__restrict__ counts, int cap, float thresh,
                                                       const float *__restrict__ boxes, const BoxTrig *__restrict__ trig,
                                                       unsigned long long *__restrict__ mask) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int col_blk = blockIdx.y;
    const int grp = blockIdx.z;
    if (counts) boxes_num = min(counts[grp], cap);
    const int stride = counts ? (cap + 63) / 64 : (boxes_num + 63) / 64;  // mask words per row
    boxes += (size_t)grp * cap * 5;
    trig += (size_t)grp * cap;
    mask += (size_t)grp * cap * stride;
    if (row >= boxes_num || col_blk < (row >> 6) || col_blk * 64 >= boxes_num) return;  // wave-uniform: tiles left of the diagonal are never read
    const int col = col_blk * 64 + lane;
    bool over = false;
    if (col < boxes_num && col > row) {  // on the diagonal tile only the bits right of the row itself (:281-283)
        float cb[5], ob[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            cb[i] = boxes[row * 5 + i];
            ob[i] = boxes[col * 5 + i];
        }
        const float v = ROTATED ? iou_bev(cb, ob, trig[row], trig[col]) : iou_normal(cb, ob);
        over = v > thresh;
    }
    const unsigned long long bits = __ballot(over);
    if (lane == 0) mask[(size_t)row * stride + col_blk] = bits;
}

// Greedy sweep of iou3d.cpp:100-116 on the device, one workgroup. For each 64-box tile: wave 0
// resolves the tile against its diagonal mask words (fixed-point iteration on wave-wide ORs, see
// below) and appends the kept positions; then ALL threads spread the (kept row, later column) mask
// words of the tile over themselves -- independent, coalesced loads -- and OR them into the
// removed-set words in LDS.
constexpr int kSweepThreads = 1024;
__global__ __launch_bounds__(kSweepThreads) void nms_sweep_kernel(int boxes_num, const int *__restrict__ counts, int cap,
                                                                  const unsigned long long *__restrict__ mask,
                                                                  long long *__restrict__ keep,
                                                                  int *__restrict__ num_keep) {
    extern __shared__ unsigned long long remv[];  // col_blocks words
    __shared__ unsigned long long kept_bits;
    __shared__ int kept_total;
    __shared__ unsigned char kept_rows[64];
    const int grp = blockIdx.x;
    if (counts) boxes_num = min(counts[grp], cap);
    const int col_blocks = (boxes_num + 63) / 64;
    const int stride = counts ? (cap + 63) / 64 : col_blocks;  // mask words per row
    mask += (size_t)grp * cap * stride;
    keep += (size_t)grp * cap;
    num_keep += grp;
    for (int j = threadIdx.x; j < col_blocks; j += kSweepThreads) remv[j] = 0ull;
    if (threadIdx.x == 0) kept_total = 0;
    __syncthreads();
    // wave 0 owns the tile's diagonal mask words, one row per lane; the next tile's are requested a tile ahead
    // (they do not depend on the removed set)
    unsigned long long diag_next = 0ull;
    if (threadIdx.x < 64 && threadIdx.x < min(boxes_num, 64)) diag_next = mask[(size_t)threadIdx.x * stride];
    for (int blk = 0; blk < col_blocks; ++blk) {
        const int size = min(boxes_num - blk * 64, 64);
        if (threadIdx.x < 64) {
            const int lane = threadIdx.x;
            const unsigned long long diag = diag_next;  // bits > lane only (upper triangle)
            if (blk + 1 < col_blocks && lane < min(boxes_num - (blk + 1) * 64, 64))
                diag_next = mask[(size_t)((blk + 1) * 64 + lane) * stride + blk + 1];
            unsigned long long alive = ~remv[blk];
            if (size < 64) alive &= (1ull << size) - 1ull;
            // greedy selection inside the tile: kept_i = alive_i and no kept j < i suppresses i. Solved by iterating
            // kept <- alive & ~OR{diag_j : j in kept} from kept = alive to its fixed point: bit i is final after i+1
            // rounds, and overlap chains are short, so this takes a handful of wave-wide ORs instead of a 64-step
            // chain of cross-lane reads
            unsigned long long kept = alive;
            for (int round = 0; round < 64; ++round) {
                unsigned long long sup = ((kept >> lane) & 1ull) ? diag : 0ull;
                unsigned lo = (unsigned)sup, hi = (unsigned)(sup >> 32);
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    lo |= (unsigned)__shfl_xor((int)lo, off, 64);
                    hi |= (unsigned)__shfl_xor((int)hi, off, 64);
                }
                const unsigned long long next = alive & ~(((unsigned long long)hi << 32) | (unsigned long long)lo);
                const unsigned long long next_u = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(next >> 32)) << 32) |
                                                  (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)next);
                if (next_u == kept) break;
                kept = next_u;
            }
            const int base = kept_total;
            if ((kept >> lane) & 1ull) {
                const int pos = (int)__popcll(kept & ((1ull << lane) - 1ull));
                keep[base + pos] = blk * 64 + lane;
                kept_rows[pos] = (unsigned char)lane;
            }
            if (lane == 0) {
                kept_bits = kept;
                kept_total = base + (int)__popcll(kept);
            }
        }
        __syncthreads();
        const int nk = (int)__popcll(kept_bits);
        const int ncols = col_blocks - (blk + 1);  // columns to the right of the tile
        const int work = nk * ncols;
        for (int w = threadIdx.x; w < work; w += kSweepThreads) {
            const int r = w / ncols, j = blk + 1 + (w - r * ncols);
            const unsigned long long m = mask[(size_t)(blk * 64 + kept_rows[r]) * stride + j];
            if (m) atomicOr(&remv[j], m);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *num_keep = kept_total;
}

template <bool ROTATED>
static int nms_impl(const float *boxes, int boxes_num, float thresh, void *workspace, size_t workspace_bytes,
                    int64_t *keep, int *num_keep, hipStream_t s) {
    if (boxes_num < 0) return EPNET_EINVAL;
    if (!num_keep) return EPNET_EINVAL;
    if (boxes_num == 0) {
        hipError_t e = hipMemsetAsync(num_keep, 0, sizeof(int), s);
        return e == hipSuccess ? EPNET_OK : record_hip_error(e, "nms memset");
    }
    if (!(boxes && keep && workspace)) return EPNET_EINVAL;
    if (workspace_bytes < epnet_nms_workspace_bytes(boxes_num)) return EPNET_ENOMEM;
    const int col_blocks = (boxes_num + 63) / 64;
    if (col_blocks > 65535 || (size_t)col_blocks * 8 > 60 * 1024) return EPNET_ELIMIT;
    unsigned long long *mask = (unsigned long long *)workspace;
    BoxRot *rot = (BoxRot *)(mask + (size_t)boxes_num * col_blocks);
    if (ROTATED) {
        hipLaunchKernelGGL(box_rot_kernel, dim3(div_up(boxes_num, 256)), dim3(256), 0, s, boxes_num, (const int *)nullptr, boxes_num,
                           boxes, rot);
        int rc0 = check_launch("nms_rot");
        if (rc0) return rc0;
        hipLaunchKernelGGL(nms_mask_rot_kernel, dim3(col_blocks, col_blocks), dim3(kRotThreads), 0, s, boxes_num, (const int *)nullptr,
                           boxes_num, thresh, rot, mask);
    } else {
        hipLaunchKernelGGL(nms_mask_kernel<false>, dim3(div_up(boxes_num, 4), col_blocks), dim3(256), 0, s, boxes_num,
                           (const int *)nullptr, boxes_num, thresh, boxes, (const BoxTrig *)nullptr, mask);
    }
    int rc = check_launch("nms_mask");
    if (rc) return rc;
    hipLaunchKernelGGL(nms_sweep_kernel, dim3(1), dim3(kSweepThreads), (size_t)col_blocks * 8, s, boxes_num, (const int *)nullptr,
                       boxes_num, mask, (long long *)keep, num_keep);
    return check_launch("nms_sweep");
}

// `groups` independent NMS problems of at most `cap` boxes each, counts on the device (see the kernels)
static int nms_groups(bool rotated, int groups, int cap, const int *counts, const float *boxes, float thresh, BoxRot *rot,
                      unsigned long long *mask, long long *keep, int *num_keep, hipStream_t s) {
    const int col_blocks = (cap + 63) / 64;
    if (col_blocks > 65535 || groups > 65535 || (size_t)col_blocks * 8 > 60 * 1024) return EPNET_ELIMIT;
    if (rotated) {
        hipLaunchKernelGGL(box_rot_kernel, dim3(div_up(cap, 256), groups), dim3(256), 0, s, 0, counts, cap, boxes, rot);
        int rc0 = check_launch("nms_rot");
        if (rc0) return rc0;
        hipLaunchKernelGGL(nms_mask_rot_kernel, dim3(col_blocks, col_blocks, groups), dim3(kRotThreads), 0, s, 0, counts, cap, thresh,
                           rot, mask);
    } else {
        hipLaunchKernelGGL(nms_mask_kernel<false>, dim3(div_up(cap, 4), col_blocks, groups), dim3(256), 0, s, 0, counts, cap, thresh,
                           boxes, (const BoxTrig *)nullptr, mask);
    }
    int rc = check_launch("nms_mask");
    if (rc) return rc;
    hipLaunchKernelGGL(nms_sweep_kernel, dim3(groups), dim3(kSweepThreads), (size_t)col_blocks * 8, s, 0, counts, cap, mask, keep,
                       num_keep);
    return check_launch("nms_sweep");
}

// ---- proposal layer (lib/rpn/proposal_layer.py:15-142) ---------------------------------------
// The reference walks the scenes on the host: boolean-mask indexing by distance bin (a sync each), top-K slices, NMS (mask
// to the host, host sweep), more slices, torch.cat, copy into the zero-padded result -- about ten host syncs per scene.
// Here: (1) one workgroup per scene walks the score-sorted order and compacts the members of each distance bin, in order,
// up to the bin's pre-NMS budget, and writes their BEV boxes (kitti_utils.boxes3d_to_bev_torch :137-150); (2) the NMS of
// all (scene, bin) groups runs as one batched mask + sweep with the group sizes read from device memory; (3) one kernel
// gathers the first post-NMS survivors of bin 0, then of bin 1, into the zero-padded (b, post, 7) / (b, post) results.
// No value leaves the device. bins == 1 is score_based_proposal (:121-142): one bin that takes every box.
constexpr int kBinThreads = 1024;

__device__ __forceinline__ int bin_of(float z, int bins) {  // nms_range_list = [0, 40, 80] (:65, :77-80)
    if (bins == 1) return 0;
    if (z > 0.f && z <= 40.f) return 0;
    if (z > 40.f && z <= 80.f) return 1;
    return -1;
}

// sel (b, 2, tot): positions IN THE SORTED ORDER of the members of each bin; counts (b*2): members kept for the NMS
__global__ __launch_bounds__(kBinThreads) void proposal_bin_kernel(int n, int bins, int pre0, int pre1, int cap,
                                                                   const float *__restrict__ proposals,
                                                                   const long long *__restrict__ order, int *__restrict__ sel,
                                                                   int *__restrict__ counts, float *__restrict__ bev) {
    __shared__ int wave_cnt[2][kBinThreads / 64];
    __shared__ int base[2];
    const int scene = blockIdx.x, lane = lane_id(), wave = threadIdx.x >> 6;
    const int tot = pre0 + pre1;
    proposals += (size_t)scene * n * 7;
    order += (size_t)scene * n;
    int *sel0 = sel + (size_t)scene * 2 * tot, *sel1 = sel0 + tot;
    if (threadIdx.x < 2) base[threadIdx.x] = 0;
    __syncthreads();
    for (int start = 0; start < n; start += kBinThreads) {
        const int i = start + threadIdx.x;
        int bin = -1;
        if (i < n) bin = bin_of(proposals[(size_t)order[i] * 7 + 2], bins);
        const unsigned long long m0 = __ballot(bin == 0), m1 = __ballot(bin == 1);
        if (lane == 0) {
            wave_cnt[0][wave] = __popcll(m0);
            wave_cnt[1][wave] = __popcll(m1);
        }
        __syncthreads();
        int before0 = base[0], before1 = base[1];
        for (int w = 0; w < wave; ++w) {
            before0 += wave_cnt[0][w];
            before1 += wave_cnt[1][w];
        }
        // bin 0 keeps pre0 + pre1 members: the tail serves bin 1 when that area holds no box at all (:92-100)
        if (bin == 0) {
            const int pos = before0 + popc_below(m0);
            if (pos < tot) sel0[pos] = i;
        } else if (bin == 1) {
            const int pos = before1 + popc_below(m1);
            if (pos < pre1) sel1[pos] = i;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int t0 = 0, t1 = 0;
            for (int w = 0; w < kBinThreads / 64; ++w) {
                t0 += wave_cnt[0][w];
                t1 += wave_cnt[1][w];
            }
            base[0] += t0;
            base[1] += t1;
        }
        __syncthreads();
    }
    const int c0 = base[0], c1 = base[1];
    const int n0 = min(c0, pre0);
    int n1 = min(c1, pre1);
    if (bins == 2 && c1 == 0) {  // "this area doesn't have any points, so use rois of first area" (:92-100)
        n1 = max(min(c0, tot) - pre0, 0);
        for (int j = threadIdx.x; j < n1; j += kBinThreads) sel1[j] = sel0[pre0 + j];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        counts[scene * 2] = n0;
        counts[scene * 2 + 1] = bins == 2 ? n1 : 0;
    }
    // BEV boxes of the members, in order: [x - l/2, z - w/2, x + l/2, z + w/2, ry]
    for (int bin = 0; bin < bins; ++bin) {
        const int cnt = bin ? n1 : n0;
        const int *src = bin ? sel1 : sel0;
        float *dst = bev + ((size_t)scene * 2 + bin) * cap * 5;
        for (int j = threadIdx.x; j < cnt; j += kBinThreads) {
            const float *p = proposals + (size_t)order[src[j]] * 7;
            const float half_l = p[5] / 2, half_w = p[4] / 2;
            dst[j * 5 + 0] = p[0] - half_l;
            dst[j * 5 + 1] = p[2] - half_w;
            dst[j * 5 + 2] = p[0] + half_l;
            dst[j * 5 + 3] = p[2] + half_w;
            dst[j * 5 + 4] = p[6];
        }
    }
}

__global__ __launch_bounds__(256) void proposal_emit_kernel(int n, int pre0, int pre1, int cap, int post0, int post1,
                                                            const float *__restrict__ proposals, const float *__restrict__ scores,
                                                            const long long *__restrict__ order, const int *__restrict__ sel,
                                                            const long long *__restrict__ keep, const int *__restrict__ num_keep,
                                                            float *__restrict__ ret_bbox3d, float *__restrict__ ret_scores,
                                                            int *__restrict__ ret_count) {
    const int scene = blockIdx.x;
    const int tot = pre0 + pre1, post = post0 + post1;
    const int k0 = min(num_keep[scene * 2], post0), k1 = min(num_keep[scene * 2 + 1], post1);
    if (threadIdx.x == 0 && ret_count) ret_count[scene] = k0 + k1;
    for (int j = threadIdx.x; j < post; j += 256) {
        float row[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float sc = 0.f;
        if (j < k0 + k1) {  // the kept boxes of bin 0, then those of bin 1 (torch.cat, :117-118)
            const int bin = j < k0 ? 0 : 1, pos = j < k0 ? j : j - k0;
            const int member = (int)keep[((size_t)scene * 2 + bin) * cap + pos];
            const long long src = order[(size_t)scene * n + sel[((size_t)scene * 2 + bin) * tot + member]];
#pragma unroll
            for (int q = 0; q < 7; ++q) row[q] = proposals[((size_t)scene * n + src) * 7 + q];
            sc = scores[(size_t)scene * n + src];
        }
#pragma unroll
        for (int q = 0; q < 7; ++q) ret_bbox3d[((size_t)scene * post + j) * 7 + q] = row[q];
        ret_scores[(size_t)scene * post + j] = sc;
    }
}

}  // namespace epnet

using namespace epnet;

extern "C" int epnet_boxes_overlap_bev(int num_a, const float *boxes_a, int num_b, const float *boxes_b, float *ans,
                                       epnet_stream_t stream) {
    EPNET_REQUIRE(num_a >= 0 && num_b >= 0);
    if (num_a == 0 || num_b == 0) return EPNET_OK;
    EPNET_REQUIRE(boxes_a && boxes_b && ans);
    if (div_up(num_a, 4) > 65535) return EPNET_ELIMIT;
    hipLaunchKernelGGL(pairwise_bev_kernel<false>, dim3(div_up(num_b, 64), div_up(num_a, 4)), dim3(256), 0,
                       (hipStream_t)stream, num_a, boxes_a, num_b, boxes_b, ans);
    return check_launch("boxes_overlap_bev");
}

extern "C" int epnet_boxes_iou_bev(int num_a, const float *boxes_a, int num_b, const float *boxes_b, float *ans,
                                   epnet_stream_t stream) {
    EPNET_REQUIRE(num_a >= 0 && num_b >= 0);
    if (num_a == 0 || num_b == 0) return EPNET_OK;
    EPNET_REQUIRE(boxes_a && boxes_b && ans);
    if (div_up(num_a, 4) > 65535) return EPNET_ELIMIT;
    hipLaunchKernelGGL(pairwise_bev_kernel<true>, dim3(div_up(num_b, 64), div_up(num_a, 4)), dim3(256), 0,
                       (hipStream_t)stream, num_a, boxes_a, num_b, boxes_b, ans);
    return check_launch("boxes_iou_bev");
}

extern "C" size_t epnet_nms_workspace_bytes(int boxes_num) {
    if (boxes_num <= 0) return 0;
    const size_t col_blocks = ((size_t)boxes_num + 63) / 64;
    return (size_t)boxes_num * col_blocks * sizeof(unsigned long long) + (size_t)boxes_num * sizeof(BoxRot);
}

extern "C" int epnet_nms(const float *boxes, int boxes_num, float thresh, void *workspace, size_t workspace_bytes,
                         int64_t *keep, int *num_keep, epnet_stream_t stream) {
    return nms_impl<true>(boxes, boxes_num, thresh, workspace, workspace_bytes, keep, num_keep, (hipStream_t)stream);
}

extern "C" int epnet_nms_normal(const float *boxes, int boxes_num, float thresh, void *workspace,
                                size_t workspace_bytes, int64_t *keep, int *num_keep, epnet_stream_t stream) {
    return nms_impl<false>(boxes, boxes_num, thresh, workspace, workspace_bytes, keep, num_keep, (hipStream_t)stream);
}

extern "C" int epnet_boxes_iou3d(int num_a, const float *boxes_a, int num_b, const float *boxes_b, float *ans,
                                 epnet_stream_t stream) {
    EPNET_REQUIRE(num_a >= 0 && num_b >= 0);
    if (num_a == 0 || num_b == 0) return EPNET_OK;
    EPNET_REQUIRE(boxes_a && boxes_b && ans);
    if (div_up(num_a, 4) > 65535) return EPNET_ELIMIT;
    hipLaunchKernelGGL(iou3d_matrix_kernel, dim3(div_up(num_b, 64), div_up(num_a, 4)), dim3(256), 0, (hipStream_t)stream,
                       num_a, boxes_a, num_b, boxes_b, ans);
    return check_launch("boxes_iou3d");
}

extern "C" int epnet_boxes_iou3d_pairs(int k, const float *boxes_a, const float *boxes_b, float *ans,
                                       epnet_stream_t stream) {
    EPNET_REQUIRE(k >= 0);
    if (k == 0) return EPNET_OK;
    EPNET_REQUIRE(boxes_a && boxes_b && ans);
    hipLaunchKernelGGL(iou3d_pairs_kernel, dim3(div_up(k, 256)), dim3(256), 0, (hipStream_t)stream, k, boxes_a, boxes_b, ans);
    return check_launch("boxes_iou3d_pairs");
}

extern "C" int epnet_aug_roi_by_noise(int k, int aug_times, float pos_thresh, float *roi_boxes3d, const float *gt_boxes3d,
                                      const float *iou3d_src, const int *tries, const unsigned char *keep_draw,
                                      const float *noise, float *iou_of_rois, epnet_stream_t stream) {
    EPNET_REQUIRE(k >= 0 && aug_times >= 0);
    if (k == 0) return EPNET_OK;
    EPNET_REQUIRE(roi_boxes3d && gt_boxes3d && iou3d_src && iou_of_rois);
    EPNET_REQUIRE(aug_times == 0 || (keep_draw && noise));
    hipStream_t s = (hipStream_t)stream;
    if (aug_times >= 1 && aug_times <= 64 && 0.f < pos_thresh) {  // (the loop starts from temp_iou = 0, :225)
        int tp = 1;
        while (tp < aug_times) tp *= 2;
        const int per_wave = 64 / tp;
        const int waves = div_up(k, per_wave);
        hipLaunchKernelGGL(aug_roi_wave_kernel, dim3(div_up(waves, 4)), dim3(256), 0, s, k, aug_times, tp, pos_thresh,
                           roi_boxes3d, gt_boxes3d, iou3d_src, tries, keep_draw, noise, iou_of_rois);
    } else {
        hipLaunchKernelGGL(aug_roi_serial_kernel, dim3(div_up(k, 64)), dim3(64), 0, s, k, aug_times, pos_thresh, roi_boxes3d,
                           gt_boxes3d, iou3d_src, tries, keep_draw, noise, iou_of_rois);
    }
    return check_launch("aug_roi_by_noise");
}

namespace {
struct ProposalPlan {
    int bins, pre0, pre1, post0, post1, cap, groups;
    size_t off_counts, off_num_keep, off_sel, off_bev, off_trig, off_keep, off_mask, bytes;
};

inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

// pre / post budgets of the two distance bins: int(0.7 * total) and the rest (proposal_layer.py:66-69)
ProposalPlan proposal_plan(int b, int distance_based, int pre_nms_top_n, int post_nms_top_n) {
    ProposalPlan p;
    p.bins = distance_based ? 2 : 1;
    p.pre0 = distance_based ? (int)(pre_nms_top_n * 0.7) : pre_nms_top_n;
    p.pre1 = pre_nms_top_n - p.pre0;
    p.post0 = distance_based ? (int)(post_nms_top_n * 0.7) : post_nms_top_n;
    p.post1 = post_nms_top_n - p.post0;
    p.cap = p.pre0 > p.pre1 ? p.pre0 : p.pre1;
    if (p.cap < 1) p.cap = 1;
    p.groups = b * 2;
    const size_t g = (size_t)p.groups, cap = (size_t)p.cap, tot = (size_t)(p.pre0 + p.pre1);
    size_t off = 0;
    p.off_counts = off;   off = align16(off + g * sizeof(int));
    p.off_num_keep = off; off = align16(off + g * sizeof(int));
    p.off_sel = off;      off = align16(off + g * tot * sizeof(int));
    p.off_bev = off;      off = align16(off + g * cap * 5 * sizeof(float));
    p.off_trig = off;     off = align16(off + g * cap * sizeof(epnet::BoxRot));
    p.off_keep = off;     off = align16(off + g * cap * sizeof(long long));
    p.off_mask = off;     off = align16(off + g * cap * ((cap + 63) / 64) * sizeof(unsigned long long));
    p.bytes = off;
    return p;
}
}  // namespace

extern "C" size_t epnet_rpn_proposals_workspace_bytes(int b, int distance_based, int pre_nms_top_n, int post_nms_top_n) {
    if (b <= 0 || pre_nms_top_n < 0 || post_nms_top_n < 0) return 0;
    return proposal_plan(b, distance_based, pre_nms_top_n, post_nms_top_n).bytes;
}

extern "C" int epnet_rpn_proposals(int b, int n, const float *proposals, const float *scores, const int64_t *order,
                                   int distance_based, int pre_nms_top_n, int post_nms_top_n, float nms_thresh, int rotated,
                                   void *workspace, size_t workspace_bytes, float *ret_bbox3d, float *ret_scores, int *ret_count,
                                   epnet_stream_t stream) {
    EPNET_REQUIRE(b >= 0 && n >= 0 && pre_nms_top_n >= 0 && post_nms_top_n >= 0);
    if (b == 0 || post_nms_top_n == 0) return EPNET_OK;
    EPNET_REQUIRE(ret_bbox3d && ret_scores && workspace);
    EPNET_REQUIRE(n == 0 || (proposals && scores && order));
    const ProposalPlan p = proposal_plan(b, distance_based, pre_nms_top_n, post_nms_top_n);
    if (workspace_bytes < p.bytes) return EPNET_ENOMEM;
    if (b > 32767) return EPNET_ELIMIT;
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)workspace;
    int *counts = (int *)(ws + p.off_counts), *num_keep = (int *)(ws + p.off_num_keep), *sel = (int *)(ws + p.off_sel);
    float *bev = (float *)(ws + p.off_bev);
    BoxRot *trig = (BoxRot *)(ws + p.off_trig);
    long long *keep = (long long *)(ws + p.off_keep);
    unsigned long long *mask = (unsigned long long *)(ws + p.off_mask);
    hipLaunchKernelGGL(proposal_bin_kernel, dim3(b), dim3(kBinThreads), 0, s, n, p.bins, p.pre0, p.pre1, p.cap, proposals,
                       (const long long *)order, sel, counts, bev);
    int rc = check_launch("rpn_proposals bin");
    if (rc) return rc;
    rc = nms_groups(rotated != 0, p.groups, p.cap, counts, bev, nms_thresh, trig, mask, keep, num_keep, s);
    if (rc) return rc;
    hipLaunchKernelGGL(proposal_emit_kernel, dim3(b), dim3(256), 0, s, n, p.pre0, p.pre1, p.cap, p.post0, p.post1, proposals, scores,
                       (const long long *)order, sel, keep, num_keep, ret_bbox3d, ret_scores, ret_count);
    return check_launch("rpn_proposals emit");
}

