// host.cpp -- the non-kernel part of libepnet_hip.so: ABI bookkeeping, error strings and the two
// roipool3d ops that are HOST-memory ops in the reference itself (roipool3d.cpp:97-195; they run
// inside DataLoader worker processes, lib/datasets/kitti_rcnn_dataset.py:672,767,811,1029,1157).
#include <math.h>
#include <string.h>

#include <string>

#include "common.h"

namespace epnet {

static thread_local std::string g_last_hip_error;

int record_hip_error(hipError_t e, const char *where) {
    g_last_hip_error = std::string(where) + ": " + hipGetErrorString(e);
    return EPNET_ELAUNCH;
}

// pt_in_box3d_cpu, lib/utils/roipool3d/src/roipool3d.cpp:82-95. The box-only terms are hoisted
// out of the point loop by the callers below (same values, computed once per box).
struct BoxTerms {
    float cx, cy, cz, hh, hw, hl, cosa, sina;
};

static inline BoxTerms box_terms(const float *b) {
    BoxTerms t;
    t.cx = b[0];
    t.cz = b[2];
    t.cy = (float)((double)b[1] - (double)b[3] / 2.0);
    t.hh = b[3] * 0.5f;
    t.hw = b[4] * 0.5f;
    t.hl = b[5] * 0.5f;
    t.cosa = (float)cos((double)b[6]);
    t.sina = (float)sin((double)b[6]);
    return t;
}

static inline int pt_in_box(const BoxTerms &t, float x, float y, float z) {
    const float max_dis = 10.0f;
    if ((fabsf(x - t.cx) > max_dis) || (fabsf(y - t.cy) > t.hh) || (fabsf(z - t.cz) > max_dis)) return 0;
    const float x_rot = (x - t.cx) * t.cosa + (z - t.cz) * (-t.sina);
    const float z_rot = (x - t.cx) * t.sina + (z - t.cz) * t.cosa;
    return (x_rot >= -t.hl) & (x_rot <= t.hl) & (z_rot >= -t.hw) & (z_rot <= t.hw);
}

}  // namespace epnet

using namespace epnet;

extern "C" int epnet_abi_version(void) { return EPNET_ABI_VERSION; }

extern "C" const char *epnet_strerror(int code) {
    switch (code) {
        case EPNET_OK: return "ok";
        case EPNET_EINVAL: return "invalid argument (negative size or NULL pointer)";
        case EPNET_ELAUNCH: return "HIP kernel launch failed";
        case EPNET_ENOMEM: return "workspace too small";
        case EPNET_ELIMIT: return "problem size outside the supported range";
        default: return "unknown epnet error code";
    }
}

extern "C" const char *epnet_last_hip_error(void) { return g_last_hip_error.c_str(); }

extern "C" int epnet_pts_in_boxes3d_host(int64_t *pts_flag, const float *pts, const float *boxes3d, int64_t boxes_num,
                                         int64_t pts_num) {
    EPNET_REQUIRE(boxes_num >= 0 && pts_num >= 0);
    if (boxes_num == 0 || pts_num == 0) return EPNET_OK;
    EPNET_REQUIRE(pts_flag && pts && boxes3d);
    for (int64_t i = 0; i < boxes_num; ++i) {
        const BoxTerms t = box_terms(boxes3d + i * 7);
        int64_t *row = pts_flag + i * pts_num;
        for (int64_t j = 0; j < pts_num; ++j) row[j] = pt_in_box(t, pts[j * 3], pts[j * 3 + 1], pts[j * 3 + 2]);
    }
    return EPNET_OK;
}

extern "C" int epnet_roipool3d_host(const float *pts, const float *boxes3d, const float *pts_feature, float *pooled_pts,
                                    float *pooled_features, int64_t *pooled_empty_flag, int64_t boxes_num,
                                    int64_t pts_num, int64_t feature_len, int64_t sampled_pts_num) {
    EPNET_REQUIRE(boxes_num >= 0 && pts_num >= 0 && feature_len >= 0 && sampled_pts_num >= 0);
    if (boxes_num == 0) return EPNET_OK;
    EPNET_REQUIRE(boxes3d && pooled_empty_flag);
    EPNET_REQUIRE(pts_num == 0 || (pts && (pts_feature || feature_len == 0)));
    EPNET_REQUIRE(sampled_pts_num == 0 || (pooled_pts && (pooled_features || feature_len == 0)));
    const int64_t S = sampled_pts_num, C = feature_len;
    for (int64_t i = 0; i < boxes_num; ++i) {
        const BoxTerms t = box_terms(boxes3d + i * 7);
        float *op = pooled_pts + i * S * 3;
        float *of = pooled_features + i * S * C;
        int64_t cnt = 0;
        for (int64_t j = 0; j < pts_num; ++j) {
            if (!pt_in_box(t, pts[j * 3], pts[j * 3 + 1], pts[j * 3 + 2])) continue;
            if (cnt >= S) break;
            memcpy(op + cnt * 3, pts + j * 3, 3 * sizeof(float));
            if (C) memcpy(of + cnt * C, pts_feature + j * C, (size_t)C * sizeof(float));
            ++cnt;
        }
        pooled_empty_flag[i] = (cnt == 0);
        if (cnt > 0)
            for (int64_t j = cnt; j < S; ++j) {  // cyclic duplication, roipool3d.cpp:180-192
                memcpy(op + j * 3, op + (j % cnt) * 3, 3 * sizeof(float));
                if (C) memcpy(of + j * C, of + (j % cnt) * C, (size_t)C * sizeof(float));
            }
    }
    return EPNET_OK;
}
