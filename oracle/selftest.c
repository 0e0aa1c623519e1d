/* TEST INFRASTRUCTURE: drives every function of the CPU oracle over small random and degenerate inputs.
 * Built and run under the address / undefined-behaviour sanitizers by tests/test_sanitizers.py (GPU sanitizers are
 * not available on the pool, so the CPU restatement -- what every GPU result is compared with -- gets them). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "epnet_oracle.h"

static unsigned long long rng_state = 88172645463325252ull;
static double rnd(void) {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (double)(rng_state >> 11) / 9007199254740992.0;
}
static float *cloud(int n) {
    float *p = malloc(sizeof(float) * 3 * (n > 0 ? n : 1));
    for (int i = 0; i < 3 * n; ++i) p[i] = (float)(rnd() * 20.0 - 10.0);
    return p;
}

static void sa_case(int b, int n, int m, int ns, int c) {
    float *xyz = cloud(b * n), *temp = malloc(sizeof(float) * (b * n + 1));
    int *fidx = malloc(sizeof(int) * (b * m + 1));
    for (int i = 0; i < b * n; ++i) temp[i] = 1e10f;
    oracle_furthest_point_sampling(b, n, m, xyz, temp, fidx);
    float *new_xyz = malloc(sizeof(float) * 3 * (b * m + 1));
    for (int s = 0; s < b; ++s)
        for (int i = 0; i < m; ++i) memcpy(new_xyz + (s * m + i) * 3, xyz + (s * n + fidx[s * m + i]) * 3, 12);
    int *bq = calloc((size_t)b * m * ns + 1, sizeof(int));
    oracle_ball_query(b, n, m, 3.0f, ns, new_xyz, xyz, bq);
    float *feat = malloc(sizeof(float) * ((size_t)b * c * n + 1));
    for (int i = 0; i < b * c * n; ++i) feat[i] = (float)rnd();
    float *grouped = malloc(sizeof(float) * ((size_t)b * c * m * ns + 1));
    oracle_group_points(b, c, n, m, ns, feat, bq, grouped);
    float *gfeat = calloc((size_t)b * c * n + 1, sizeof(float));
    oracle_group_points_grad(b, c, n, m, ns, grouped, bq, gfeat);
    float *gath = malloc(sizeof(float) * ((size_t)b * c * m + 1));
    oracle_gather_points(b, c, n, m, feat, fidx, gath);
    memset(gfeat, 0, sizeof(float) * ((size_t)b * c * n + 1));
    oracle_gather_points_grad(b, c, n, m, gath, fidx, gfeat);
    float *d2 = malloc(sizeof(float) * 3 * (b * n + 1));
    int *nn = malloc(sizeof(int) * 3 * (b * n + 1));
    oracle_three_nn(b, n, m, xyz, new_xyz, d2, nn);
    float *w = malloc(sizeof(float) * 3 * (b * n + 1));
    for (int i = 0; i < 3 * b * n; ++i) w[i] = 1.0f / 3.0f;
    float *interp = malloc(sizeof(float) * ((size_t)b * c * n + 1));
    if (m > 0) {
        oracle_three_interpolate(b, c, m, n, gath, nn, w, interp);
        float *gk = calloc((size_t)b * c * m + 1, sizeof(float));
        oracle_three_interpolate_grad(b, c, n, m, interp, nn, w, gk);
        free(gk);
    }
    free(xyz); free(temp); free(fidx); free(new_xyz); free(bq); free(feat); free(grouped); free(gfeat); free(gath);
    free(d2); free(nn); free(w); free(interp);
}

static void box_case(int n) {
    float *bev = calloc(5 * (size_t)(n + 1), sizeof(float)), *b7 = calloc(7 * (size_t)(n + 1), sizeof(float));
    for (int i = 0; i < n; ++i) {
        const float cx = (float)(rnd() * 10), cz = (float)(rnd() * 10), l = (float)(1 + rnd() * 3), w = (float)(1 + rnd() * 2);
        bev[i * 5 + 0] = cx - l / 2; bev[i * 5 + 1] = cz - w / 2; bev[i * 5 + 2] = cx + l / 2; bev[i * 5 + 3] = cz + w / 2;
        bev[i * 5 + 4] = (float)(rnd() * 6.28 - 3.14);
        b7[i * 7 + 0] = cx; b7[i * 7 + 1] = 1.f; b7[i * 7 + 2] = cz; b7[i * 7 + 3] = 1.5f; b7[i * 7 + 4] = w; b7[i * 7 + 5] = l;
        b7[i * 7 + 6] = bev[i * 5 + 4];
    }
    float *ans = malloc(sizeof(float) * ((size_t)n * n + 1));
    oracle_boxes_overlap_bev(n, bev, n, bev, ans);
    oracle_boxes_iou_bev(n, bev, n, bev, ans);
    long long *keep = malloc(sizeof(long long) * (n + 1));
    const int k1 = oracle_nms(n, 0.3f, bev, keep, 1), k2 = oracle_nms(n, 0.3f, bev, keep, 0);
    if (n > 0 && (k1 < 1 || k2 < 1)) { printf("nms kept nothing\n"); exit(1); }
    /* roipool3d over a small cloud */
    const int pts = 300, c = 5, s = 16;
    float *xyz = cloud(pts), *feat = malloc(sizeof(float) * pts * c);
    for (int i = 0; i < pts * c; ++i) feat[i] = (float)rnd();
    float *pooled = calloc((size_t)(n + 1) * s * (3 + c), sizeof(float));
    int *flag = calloc(n + 1, sizeof(int));
    oracle_roipool3d(1, pts, n, c, s, xyz, b7, feat, pooled, flag);
    long long *pf = calloc((size_t)(n + 1) * pts, sizeof(long long));
    oracle_pts_in_boxes3d(pf, xyz, b7, n, pts);
    free(bev); free(b7); free(ans); free(keep); free(xyz); free(feat); free(pooled); free(flag); free(pf);
}

int main(void) {
    sa_case(2, 300, 40, 8, 3);
    sa_case(1, 1, 1, 1, 1);
    sa_case(1, 37, 20, 4, 2);      /* reference block size 32 */
    sa_case(2, 1030, 257, 16, 2);  /* block size 1024, two points per reference thread */
    sa_case(1, 5, 3, 6, 1);        /* nsample larger than the cloud */
    sa_case(3, 64, 64, 2, 1);      /* every point sampled */
    box_case(0);
    box_case(1);
    box_case(70);                   /* more than one 64-column block of the NMS mask */
    printf("oracle selftest ok\n");
    return 0;
}
