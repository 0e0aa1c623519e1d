/*
 * epnet_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Single-threaded CPU restatement of every device kernel on EPNet's point-cloud geometry hot
 * path, written from the text of the reference's .cu/.cpp files (cited per function as
 * path:line below /root/reference). Same loop order, same strict / non-strict comparisons,
 * same tie-breaks. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this; the shipped path (libepnet_hip.so) never does.
 *
 * Parity status: the reference ships NO tests, fixtures or golden vectors,
 * and its pointnet2 / iou3d kernels cannot be built here (nvcc absent) -> for those ops this
 * oracle is "PARITY UNPINNED" by reference tests; it is pinned by analytic known-answer tests
 * (tests/test_oracle_kat.py). The roipool3d functions ARE pinned: they are checked against the
 * reference's own CPU ops compiled from /root/reference (oracle/_ref, tests/test_oracle_vs_ref.py)
 * and against fixtures captured from that build (tests/golden/roipool3d_ref_*.npz).
 *
 * Arithmetic definition (DESIGN.md "Parity definition"):
 *   - IEEE-754 binary32, expression order as written in the reference, NO fused contraction
 *     (build with -ffp-contract=off; nvcc's --fmad default is a compiler liberty, not source
 *     semantics), correctly rounded division;
 *   - cos/sin/atan2 of float arguments are the correctly rounded float results, obtained as
 *     (float)cos((double)x) etc. -- CUDA's cosf/sinf/atan2f are only specified to 1-2 ulp, so
 *     their exact bits are not part of the reference's observable contract.
 */
#include "epnet_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline float cr_cosf(float x) { return (float)cos((double)x); }
static inline float cr_sinf(float x) { return (float)sin((double)x); }
static inline float cr_atan2f(float y, float x) { return (float)atan2((double)y, (double)x); }
static inline float f_min(float a, float b) { return a < b ? a : b; }
static inline float f_max(float a, float b) { return a > b ? a : b; }

/* ------------------------------------------------------------------------------------------
 * pointnet2: sampling
 * ---------------------------------------------------------------------------------------- */

/* pointnet2_lib/pointnet2/src/cuda_utils.h:10-14 */
int oracle_opt_n_threads(int work_size) {
    const int pow_2 = (int)(log((double)work_size) / log(2.0));
    int v = 1 << pow_2;
    if (v > 1024) v = 1024;
    if (v < 1) v = 1;
    return v;
}

/* __update, sampling_gpu.cu:86-91 */
static void fps_update(float *dists, int *dists_i, int idx1, int idx2) {
    const float v1 = dists[idx1], v2 = dists[idx2];
    const int i1 = dists_i[idx1], i2 = dists_i[idx2];
    dists[idx1] = f_max(v1, v2);
    dists_i[idx1] = v2 > v1 ? i2 : i1;
}

/* furthest_point_sampling_kernel<block_size>, sampling_gpu.cu:94-209; launcher :211-253.
 * The block of `bs` threads is simulated thread by thread, then the shared-memory tree. */
void oracle_furthest_point_sampling(int b, int n, int m, const float *dataset, float *temp, int *idxs) {
    if (m <= 0) return;
    const int bs = oracle_opt_n_threads(n);
    float dists[1024];
    int dists_i[1024];
    for (int batch = 0; batch < b; batch++) {
        const float *ds = dataset + (size_t)batch * n * 3;
        float *tp = temp + (size_t)batch * n;
        int *out = idxs + (size_t)batch * m;
        int old = 0;
        out[0] = old;
        for (int j = 1; j < m; j++) {
            const float x1 = ds[old * 3 + 0], y1 = ds[old * 3 + 1], z1 = ds[old * 3 + 2];
            for (int tid = 0; tid < bs; tid++) {
                int besti = 0;
                float best = -1;
                for (int k = tid; k < n; k += bs) {
                    const float x2 = ds[k * 3 + 0], y2 = ds[k * 3 + 1], z2 = ds[k * 3 + 2];
                    const float d = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) + (z2 - z1) * (z2 - z1);
                    const float d2 = f_min(d, tp[k]);
                    tp[k] = d2;
                    besti = d2 > best ? k : besti;
                    best = d2 > best ? d2 : best;
                }
                dists[tid] = best;
                dists_i[tid] = besti;
            }
            for (int s = bs / 2; s >= 1; s >>= 1)
                for (int tid = 0; tid < s; tid++) fps_update(dists, dists_i, tid, tid + s);
            old = dists_i[0];
            out[j] = old;
        }
    }
}

/* gather_points_kernel_fast, sampling_gpu.cu:8-24 */
void oracle_gather_points(int b, int c, int n, int m, const float *points, const int *idx, float *out) {
    for (int bi = 0; bi < b; bi++)
        for (int ci = 0; ci < c; ci++)
            for (int pi = 0; pi < m; pi++)
                out[((size_t)bi * c + ci) * m + pi] = points[((size_t)bi * c + ci) * n + idx[(size_t)bi * m + pi]];
}

/* gather_points_grad_kernel_fast, sampling_gpu.cu:46-63 (atomicAdd order is unspecified in the
 * reference; the oracle adds in ascending output index) */
void oracle_gather_points_grad(int b, int c, int n, int m, const float *grad_out, const int *idx,
                               float *grad_points) {
    for (int bi = 0; bi < b; bi++)
        for (int ci = 0; ci < c; ci++)
            for (int pi = 0; pi < m; pi++)
                grad_points[((size_t)bi * c + ci) * n + idx[(size_t)bi * m + pi]] +=
                    grad_out[((size_t)bi * c + ci) * m + pi];
}

/* ------------------------------------------------------------------------------------------
 * pointnet2: ball query / grouping
 * ---------------------------------------------------------------------------------------- */

/* ball_query_kernel_fast, ball_query_gpu.cu:9-45. idx is NOT cleared here (the caller zero-fills
 * it, pointnet2_utils.py:218), exactly as in the reference. */
void oracle_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                       const float *xyz, int *idx) {
    const float radius2 = radius * radius;
    for (int bi = 0; bi < b; bi++)
        for (int pi = 0; pi < m; pi++) {
            const float *c = new_xyz + ((size_t)bi * m + pi) * 3;
            const float *p = xyz + (size_t)bi * n * 3;
            int *o = idx + ((size_t)bi * m + pi) * nsample;
            const float new_x = c[0], new_y = c[1], new_z = c[2];
            int cnt = 0;
            for (int k = 0; k < n; ++k) {
                const float x = p[k * 3 + 0], y = p[k * 3 + 1], z = p[k * 3 + 2];
                const float d2 = (new_x - x) * (new_x - x) + (new_y - y) * (new_y - y) + (new_z - z) * (new_z - z);
                if (d2 < radius2) {
                    if (cnt == 0)
                        for (int l = 0; l < nsample; ++l) o[l] = k;
                    o[cnt] = k;
                    ++cnt;
                    if (cnt >= nsample) break;
                }
            }
        }
}

/* group_points_kernel_fast, group_points_gpu.cu:47-66 */
void oracle_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                         const int *idx, float *out) {
    const size_t p = (size_t)npoints * nsample;
    for (int bi = 0; bi < b; bi++)
        for (int ci = 0; ci < c; ci++)
            for (size_t q = 0; q < p; q++)
                out[((size_t)bi * c + ci) * p + q] = points[((size_t)bi * c + ci) * n + idx[(size_t)bi * p + q]];
}

/* group_points_grad_kernel_fast, group_points_gpu.cu:8-25 */
void oracle_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                              const int *idx, float *grad_points) {
    const size_t p = (size_t)npoints * nsample;
    for (int bi = 0; bi < b; bi++)
        for (int ci = 0; ci < c; ci++)
            for (size_t q = 0; q < p; q++)
                grad_points[((size_t)bi * c + ci) * n + idx[(size_t)bi * p + q]] +=
                    grad_out[((size_t)bi * c + ci) * p + q];
}

/* ------------------------------------------------------------------------------------------
 * pointnet2: interpolation
 * ---------------------------------------------------------------------------------------- */

/* three_nn_kernel_fast, interpolate_gpu.cu:9-52 (double accumulators initialised to 1e40) */
void oracle_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx) {
    for (int bi = 0; bi < b; bi++)
        for (int pi = 0; pi < n; pi++) {
            const float *u = unknown + ((size_t)bi * n + pi) * 3;
            const float *kn = known + (size_t)bi * m * 3;
            const float ux = u[0], uy = u[1], uz = u[2];
            double best1 = 1e40, best2 = 1e40, best3 = 1e40;
            int besti1 = 0, besti2 = 0, besti3 = 0;
            for (int k = 0; k < m; ++k) {
                const float x = kn[k * 3 + 0], y = kn[k * 3 + 1], z = kn[k * 3 + 2];
                const float d = (ux - x) * (ux - x) + (uy - y) * (uy - y) + (uz - z) * (uz - z);
                if (d < best1) {
                    best3 = best2; besti3 = besti2;
                    best2 = best1; besti2 = besti1;
                    best1 = d; besti1 = k;
                } else if (d < best2) {
                    best3 = best2; besti3 = besti2;
                    best2 = d; besti2 = k;
                } else if (d < best3) {
                    best3 = d; besti3 = k;
                }
            }
            float *dd = dist2 + ((size_t)bi * n + pi) * 3;
            int *ii = idx + ((size_t)bi * n + pi) * 3;
            dd[0] = (float)best1; dd[1] = (float)best2; dd[2] = (float)best3;
            ii[0] = besti1; ii[1] = besti2; ii[2] = besti3;
        }
}

/* three_interpolate_kernel_fast, interpolate_gpu.cu:77-97 */
void oracle_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                              const float *weight, float *out) {
    for (int bi = 0; bi < b; bi++)
        for (int ci = 0; ci < c; ci++) {
            const float *pt = points + ((size_t)bi * c + ci) * m;
            for (int pi = 0; pi < n; pi++) {
                const float *w = weight + ((size_t)bi * n + pi) * 3;
                const int *ix = idx + ((size_t)bi * n + pi) * 3;
                out[((size_t)bi * c + ci) * n + pi] = w[0] * pt[ix[0]] + w[1] * pt[ix[1]] + w[2] * pt[ix[2]];
            }
        }
}

/* three_interpolate_grad_kernel_fast, interpolate_gpu.cu:120-142 */
void oracle_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                   const float *weight, float *grad_points) {
    for (int bi = 0; bi < b; bi++)
        for (int ci = 0; ci < c; ci++) {
            float *gp = grad_points + ((size_t)bi * c + ci) * m;
            for (int pi = 0; pi < n; pi++) {
                const float *w = weight + ((size_t)bi * n + pi) * 3;
                const int *ix = idx + ((size_t)bi * n + pi) * 3;
                const float g = grad_out[((size_t)bi * c + ci) * n + pi];
                gp[ix[0]] += g * w[0];
                gp[ix[1]] += g * w[1];
                gp[ix[2]] += g * w[2];
            }
        }
}

/* ------------------------------------------------------------------------------------------
 * iou3d: rotated rectangle overlap (lib/utils/iou3d/src/iou3d_kernel.cu)
 * ---------------------------------------------------------------------------------------- */

static const float IOU_EPS = 1e-8f; /* iou3d_kernel.cu:13 */

typedef struct { float x, y; } pt2; /* Point, iou3d_kernel.cu:14-32 */

static inline pt2 pt_add(pt2 a, pt2 b) { pt2 r = {a.x + b.x, a.y + b.y}; return r; }
static inline pt2 pt_sub(pt2 a, pt2 b) { pt2 r = {a.x - b.x, a.y - b.y}; return r; }

/* cross(a,b), iou3d_kernel.cu:34-36 */
static inline float cross2(pt2 a, pt2 b) { return a.x * b.y - a.y * b.x; }
/* cross(p1,p2,p0), iou3d_kernel.cu:38-40 */
static inline float cross3(pt2 p1, pt2 p2, pt2 p0) {
    return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

/* check_rect_cross, iou3d_kernel.cu:42-48 */
static int check_rect_cross(pt2 p1, pt2 p2, pt2 q1, pt2 q2) {
    return f_min(p1.x, p2.x) <= f_max(q1.x, q2.x) && f_min(q1.x, q2.x) <= f_max(p1.x, p2.x) &&
           f_min(p1.y, p2.y) <= f_max(q1.y, q2.y) && f_min(q1.y, q2.y) <= f_max(p1.y, p2.y);
}

/* check_in_box2d, iou3d_kernel.cu:50-65 */
static int check_in_box2d(const float *box, pt2 p) {
    const float MARGIN = 1e-5f;
    const float center_x = (box[0] + box[2]) / 2;
    const float center_y = (box[1] + box[3]) / 2;
    const float angle_cos = cr_cosf(-box[4]), angle_sin = cr_sinf(-box[4]);
    const float rot_x = (p.x - center_x) * angle_cos + (p.y - center_y) * angle_sin + center_x;
    const float rot_y = -(p.x - center_x) * angle_sin + (p.y - center_y) * angle_cos + center_y;
    return (rot_x > box[0] - MARGIN && rot_x < box[2] + MARGIN && rot_y > box[1] - MARGIN &&
            rot_y < box[3] + MARGIN);
}

/* intersection, iou3d_kernel.cu:67-96 */
static int intersection(pt2 p1, pt2 p0, pt2 q1, pt2 q0, pt2 *ans) {
    if (check_rect_cross(p0, p1, q0, q1) == 0) return 0;
    const float s1 = cross3(q0, p1, p0);
    const float s2 = cross3(p1, q1, p0);
    const float s3 = cross3(p0, q1, q0);
    const float s4 = cross3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
    const float s5 = cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > IOU_EPS) {
        ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        const float D = a0 * b1 - a1 * b0;
        ans->x = (b0 * c1 - b1 * c0) / D;
        ans->y = (a1 * c0 - a0 * c1) / D;
    }
    return 1;
}

/* rotate_around_center, iou3d_kernel.cu:98-102 */
static void rotate_around_center(pt2 center, float angle_cos, float angle_sin, pt2 *p) {
    const float new_x = (p->x - center.x) * angle_cos + (p->y - center.y) * angle_sin + center.x;
    const float new_y = -(p->x - center.x) * angle_sin + (p->y - center.y) * angle_cos + center.y;
    p->x = new_x;
    p->y = new_y;
}

/* point_cmp, iou3d_kernel.cu:104-106 */
static int point_cmp(pt2 a, pt2 b, pt2 center) {
    return cr_atan2f(a.y - center.y, a.x - center.x) > cr_atan2f(b.y - center.y, b.x - center.x);
}

/* box_overlap, iou3d_kernel.cu:108-212 */
float oracle_box_overlap(const float *box_a, const float *box_b) {
    const float a_x1 = box_a[0], a_y1 = box_a[1], a_x2 = box_a[2], a_y2 = box_a[3], a_angle = box_a[4];
    const float b_x1 = box_b[0], b_y1 = box_b[1], b_x2 = box_b[2], b_y2 = box_b[3], b_angle = box_b[4];
    pt2 center_a = {(a_x1 + a_x2) / 2, (a_y1 + a_y2) / 2};
    pt2 center_b = {(b_x1 + b_x2) / 2, (b_y1 + b_y2) / 2};

    pt2 ac[5] = {{a_x1, a_y1}, {a_x2, a_y1}, {a_x2, a_y2}, {a_x1, a_y2}, {0, 0}};
    pt2 bc[5] = {{b_x1, b_y1}, {b_x2, b_y1}, {b_x2, b_y2}, {b_x1, b_y2}, {0, 0}};

    const float a_angle_cos = cr_cosf(a_angle), a_angle_sin = cr_sinf(a_angle);
    const float b_angle_cos = cr_cosf(b_angle), b_angle_sin = cr_sinf(b_angle);
    for (int k = 0; k < 4; k++) {
        rotate_around_center(center_a, a_angle_cos, a_angle_sin, &ac[k]);
        rotate_around_center(center_b, b_angle_cos, b_angle_sin, &bc[k]);
    }
    ac[4] = ac[0];
    bc[4] = bc[0];

    /* the reference declares cross_points[16]; 24 slots here only so that a (geometrically
     * impossible) overflow cannot corrupt the oracle's stack */
    pt2 cross_points[24];
    pt2 poly_center = {0, 0};
    int cnt = 0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            int flag = intersection(ac[i + 1], ac[i], bc[j + 1], bc[j], &cross_points[cnt]);
            if (flag) {
                poly_center = pt_add(poly_center, cross_points[cnt]);
                cnt++;
            }
        }
    for (int k = 0; k < 4; k++) {
        if (check_in_box2d(box_a, bc[k])) {
            poly_center = pt_add(poly_center, bc[k]);
            cross_points[cnt] = bc[k];
            cnt++;
        }
        if (check_in_box2d(box_b, ac[k])) {
            poly_center = pt_add(poly_center, ac[k]);
            cross_points[cnt] = ac[k];
            cnt++;
        }
    }
    poly_center.x /= cnt;
    poly_center.y /= cnt;

    for (int j = 0; j < cnt - 1; j++)
        for (int i = 0; i < cnt - j - 1; i++)
            if (point_cmp(cross_points[i], cross_points[i + 1], poly_center)) {
                pt2 t = cross_points[i];
                cross_points[i] = cross_points[i + 1];
                cross_points[i + 1] = t;
            }

    float area = 0;
    for (int k = 0; k < cnt - 1; k++)
        area += cross2(pt_sub(cross_points[k], cross_points[0]), pt_sub(cross_points[k + 1], cross_points[0]));
    return (float)(fabsf(area) / 2.0);
}

/* iou_bev, iou3d_kernel.cu:214-221 */
float oracle_iou_bev(const float *box_a, const float *box_b) {
    const float sa = (box_a[2] - box_a[0]) * (box_a[3] - box_a[1]);
    const float sb = (box_b[2] - box_b[0]) * (box_b[3] - box_b[1]);
    const float s_overlap = oracle_box_overlap(box_a, box_b);
    return s_overlap / fmaxf(sa + sb - s_overlap, IOU_EPS);
}

/* iou_normal, iou3d_kernel.cu:295-303 */
float oracle_iou_normal(const float *a, const float *b) {
    const float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
    const float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
    const float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
    const float interS = width * height;
    const float Sa = (a[2] - a[0]) * (a[3] - a[1]);
    const float Sb = (b[2] - b[0]) * (b[3] - b[1]);
    return interS / fmaxf(Sa + Sb - interS, IOU_EPS);
}

/* boxes_overlap_kernel, iou3d_kernel.cu:223-234 */
void oracle_boxes_overlap_bev(int num_a, const float *boxes_a, int num_b, const float *boxes_b, float *ans) {
    for (int a = 0; a < num_a; a++)
        for (int b = 0; b < num_b; b++)
            ans[(size_t)a * num_b + b] = oracle_box_overlap(boxes_a + a * 5, boxes_b + b * 5);
}

/* boxes_iou_bev_kernel, iou3d_kernel.cu:236-248 */
void oracle_boxes_iou_bev(int num_a, const float *boxes_a, int num_b, const float *boxes_b, float *ans) {
    for (int a = 0; a < num_a; a++)
        for (int b = 0; b < num_b; b++)
            ans[(size_t)a * num_b + b] = oracle_iou_bev(boxes_a + a * 5, boxes_b + b * 5);
}

/* nms_kernel / nms_normal_kernel, iou3d_kernel.cu:250-292 / :306-348: every (row block, col block)
 * pair is evaluated, lower triangle included (the `row_start > col_start` test is commented out). */
void oracle_nms_mask(int boxes_num, float thresh, const float *boxes, unsigned long long *mask, int rotated) {
    const int col_blocks = (boxes_num + 63) / 64;
    for (int row_start = 0; row_start < col_blocks; row_start++)
        for (int col_start = 0; col_start < col_blocks; col_start++) {
            int row_size = boxes_num - row_start * 64; if (row_size > 64) row_size = 64;
            int col_size = boxes_num - col_start * 64; if (col_size > 64) col_size = 64;
            for (int t = 0; t < row_size; t++) {
                const int cur_box_idx = 64 * row_start + t;
                const float *cur_box = boxes + cur_box_idx * 5;
                unsigned long long bits = 0;
                int start = 0;
                if (row_start == col_start) start = t + 1;
                for (int i = start; i < col_size; i++) {
                    const float *other = boxes + (64 * col_start + i) * 5;
                    const float v = rotated ? oracle_iou_bev(cur_box, other) : oracle_iou_normal(cur_box, other);
                    if (v > thresh) bits |= 1ULL << i;
                }
                mask[(size_t)cur_box_idx * col_blocks + col_start] = bits;
            }
        }
}

/* host greedy sweep, lib/utils/iou3d/src/iou3d.cpp:100-116 (and :150-166) */
int oracle_nms_sweep(int boxes_num, const unsigned long long *mask, long long *keep) {
    const int col_blocks = (boxes_num + 63) / 64;
    unsigned long long *remv = (unsigned long long *)calloc((size_t)col_blocks + 1, sizeof(unsigned long long));
    int num_to_keep = 0;
    for (int i = 0; i < boxes_num; i++) {
        const int nblock = i / 64, inblock = i % 64;
        if (!(remv[nblock] & (1ULL << inblock))) {
            keep[num_to_keep++] = i;
            const unsigned long long *p = mask + (size_t)i * col_blocks;
            for (int j = nblock; j < col_blocks; j++) remv[j] |= p[j];
        }
    }
    free(remv);
    return num_to_keep;
}

/* nms_gpu / nms_normal_gpu, iou3d.cpp:73-120 / :123-170 */
int oracle_nms(int boxes_num, float thresh, const float *boxes, long long *keep, int rotated) {
    const int col_blocks = (boxes_num + 63) / 64;
    unsigned long long *mask = (unsigned long long *)malloc(sizeof(unsigned long long) * (size_t)boxes_num * col_blocks + 8);
    oracle_nms_mask(boxes_num, thresh, boxes, mask, rotated);
    const int r = oracle_nms_sweep(boxes_num, mask, keep);
    free(mask);
    return r;
}

/* ------------------------------------------------------------------------------------------
 * roipool3d (lib/utils/roipool3d/src/roipool3d_kernel.cu, roipool3d.cpp)
 * ---------------------------------------------------------------------------------------- */

/* pt_in_box3d, roipool3d_kernel.cu:14-28 == pt_in_box3d_cpu, roipool3d.cpp:82-95 (max_dis = 10).
 * The double-typed sub-expressions of the reference (h / 2.0, -l / 2.0 ...) are kept. */
int oracle_pt_in_box3d(float x, float y, float z, float cx, float bottom_y, float cz, float h, float w,
                       float l, float angle) {
    const float max_dis = 10.0f;
    float x_rot, z_rot, cosa, sina, cy;
    cy = (float)(bottom_y - h / 2.0);
    if ((fabsf(x - cx) > max_dis) || (fabsf(y - cy) > h / 2.0) || (fabsf(z - cz) > max_dis)) return 0;
    cosa = cr_cosf(angle);
    sina = cr_sinf(angle);
    x_rot = (x - cx) * cosa + (z - cz) * (-sina);
    z_rot = (x - cx) * sina + (z - cz) * cosa;
    return (x_rot >= -l / 2.0) & (x_rot <= l / 2.0) & (z_rot >= -w / 2.0) & (z_rot <= w / 2.0);
}

/* assign_pts_to_box3d + get_pooled_idx + roipool3d_forward, roipool3d_kernel.cu:97-194
 * (launcher :209-237). pooled_features / pooled_empty_flag are NOT cleared (caller zero-fills). */
void oracle_roipool3d(int batch_size, int pts_num, int boxes_num, int feature_in_len, int sampled_pts_num,
                      const float *xyz, const float *boxes3d, const float *pts_feature,
                      float *pooled_features, int *pooled_empty_flag) {
    int *pts_idx = (int *)malloc(sizeof(int) * (size_t)sampled_pts_num + 4);
    const int row = 3 + feature_in_len;
    for (int bs = 0; bs < batch_size; bs++)
        for (int box = 0; box < boxes_num; box++) {
            const float *bx = boxes3d + ((size_t)bs * boxes_num + box) * 7;
            int cnt = 0;
            for (int k = 0; k < pts_num; k++) {
                const float *p = xyz + ((size_t)bs * pts_num + k) * 3;
                if (oracle_pt_in_box3d(p[0], p[1], p[2], bx[0], bx[1], bx[2], bx[3], bx[4], bx[5], bx[6])) {
                    if (cnt < sampled_pts_num) {
                        pts_idx[cnt] = k;
                        cnt++;
                    } else
                        break;
                }
            }
            if (cnt == 0) {
                pooled_empty_flag[(size_t)bs * boxes_num + box] = 1;
                continue; /* roipool3d_forward returns early for empty boxes, :177-179 */
            }
            for (int k = cnt; k < sampled_pts_num; k++) pts_idx[k] = pts_idx[k % cnt];
            for (int s = 0; s < sampled_pts_num; s++) {
                const int src = pts_idx[s];
                float *dst = pooled_features + (((size_t)bs * boxes_num + box) * sampled_pts_num + s) * row;
                for (int j = 0; j < 3; j++) dst[j] = xyz[((size_t)bs * pts_num + src) * 3 + j];
                for (int j = 0; j < feature_in_len; j++)
                    dst[3 + j] = pts_feature[((size_t)bs * pts_num + src) * feature_in_len + j];
            }
        }
    free(pts_idx);
}

/* pts_in_boxes3d_cpu, roipool3d.cpp:97-125 */
void oracle_pts_in_boxes3d(long long *pts_flag, const float *pts, const float *boxes3d, long boxes_num, long pts_num) {
    memset(pts_flag, 0, (size_t)boxes_num * pts_num * sizeof(long long));
    for (long i = 0; i < boxes_num; i++)
        for (long j = 0; j < pts_num; j++)
            pts_flag[i * pts_num + j] = oracle_pt_in_box3d(
                pts[j * 3], pts[j * 3 + 1], pts[j * 3 + 2], boxes3d[i * 7], boxes3d[i * 7 + 1], boxes3d[i * 7 + 2],
                boxes3d[i * 7 + 3], boxes3d[i * 7 + 4], boxes3d[i * 7 + 5], boxes3d[i * 7 + 6]);
}

/* roipool3d_cpu, roipool3d.cpp:127-195 */
void oracle_roipool3d_cpu(const float *pts, const float *boxes3d, const float *pts_feature, float *pooled_pts,
                          float *pooled_features, long long *pooled_empty_flag, long boxes_num, long pts_num,
                          long feature_len, long sampled_pts_num) {
    memset(pooled_empty_flag, 0, (size_t)boxes_num * sizeof(long long));
    for (long i = 0; i < boxes_num; i++) {
        long cnt = 0;
        for (long j = 0; j < pts_num; j++) {
            const int in = oracle_pt_in_box3d(pts[j * 3], pts[j * 3 + 1], pts[j * 3 + 2], boxes3d[i * 7],
                                              boxes3d[i * 7 + 1], boxes3d[i * 7 + 2], boxes3d[i * 7 + 3],
                                              boxes3d[i * 7 + 4], boxes3d[i * 7 + 5], boxes3d[i * 7 + 6]);
            if (in) {
                if (cnt < sampled_pts_num) {
                    for (int k = 0; k < 3; k++) pooled_pts[(i * sampled_pts_num + cnt) * 3 + k] = pts[j * 3 + k];
                    for (long k = 0; k < feature_len; k++)
                        pooled_features[(i * sampled_pts_num + cnt) * feature_len + k] = pts_feature[j * feature_len + k];
                    cnt++;
                } else
                    break;
            }
        }
        if (cnt == 0) {
            pooled_empty_flag[i] = 1;
        } else if (cnt < sampled_pts_num) {
            for (long j = cnt; j < sampled_pts_num; j++) {
                for (int k = 0; k < 3; k++)
                    pooled_pts[(i * sampled_pts_num + j) * 3 + k] = pooled_pts[(i * sampled_pts_num + (j % cnt)) * 3 + k];
                for (long k = 0; k < feature_len; k++)
                    pooled_features[(i * sampled_pts_num + j) * feature_len + k] =
                        pooled_features[(i * sampled_pts_num + (j % cnt)) * feature_len + k];
            }
        }
    }
}
