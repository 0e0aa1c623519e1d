// TEST INFRASTRUCTURE: drives the two host-memory ops of libepnet_hip.so under ASan/UBSan (tests/test_sanitizers.py)
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include "epnet_ops.h"
int main() {
    const long pts = 500, boxes = 9, c = 6, s = 32;
    std::vector<float> xyz(pts * 3), feat(pts * c), b7(boxes * 7);
    srand(3);
    for (auto &v : xyz) v = (float)rand() / RAND_MAX * 20.f - 10.f;
    for (auto &v : feat) v = (float)rand() / RAND_MAX;
    for (long i = 0; i < boxes; ++i) { float *b = &b7[i * 7]; b[0] = (float)rand() / RAND_MAX * 10 - 5; b[1] = 2; b[2] = (float)rand() / RAND_MAX * 10 - 5; b[3] = 4; b[4] = 3; b[5] = 5; b[6] = (float)i; }
    std::vector<int64_t> flag(boxes * pts), empty(boxes);
    std::vector<float> pooled_pts(boxes * s * 3), pooled_feat(boxes * s * c);
    int rc = epnet_pts_in_boxes3d_host(flag.data(), xyz.data(), b7.data(), boxes, pts);
    rc |= epnet_roipool3d_host(xyz.data(), b7.data(), feat.data(), pooled_pts.data(), pooled_feat.data(), empty.data(), boxes, pts, c, s);
    rc |= epnet_roipool3d_host(xyz.data(), b7.data(), feat.data(), pooled_pts.data(), pooled_feat.data(), empty.data(), 0, pts, c, s);
    printf("host ops rc %d, abi %d, %s\n", rc, epnet_abi_version(), epnet_strerror(-3));
    return rc;
}
