// debug harness (not product): big-scene FPS summaries after m rounds
#define EPNET_BIG_DEBUG 1
#include "../epnet_amd/csrc/fps.hip"
#include "../epnet_amd/csrc/ball_query.hip"
#include "../epnet_amd/csrc/host.cpp"
#include <cstdio>
#include <vector>
int main(int argc, char **argv) {
    const int n = 30000, m = atoi(argv[2]);
    std::vector<float> xyz((size_t)n * 3);
    FILE *f = fopen(argv[1], "rb"); if (!f || fread(xyz.data(), 4, xyz.size(), f) != xyz.size()) { printf("no file\n"); return 1; }
    float *dx, *dt; int *di; void *ix;
    size_t nb = epnet_scene_index_bytes(1, n);
    hipMalloc(&dx, xyz.size() * 4); hipMalloc(&dt, n * 4); hipMalloc(&di, m * 4); hipMalloc(&ix, nb);
    hipMemcpy(dx, xyz.data(), xyz.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> t0(n, 1e10f); hipMemcpy(dt, t0.data(), n * 4, hipMemcpyHostToDevice);
    printf("index rc %d\n", epnet_scene_index_build(1, n, dx, ix, nb, 0));
    printf("fps rc %d\n", epnet_furthest_point_sampling_indexed(1, n, m, dx, ix, nb, dt, di, 0));
    hipDeviceSynchronize();
    std::vector<int> idx(m); hipMemcpy(idx.data(), di, m * 4, hipMemcpyDeviceToHost);
    std::vector<float> temp(n); hipMemcpy(temp.data(), dt, n * 4, hipMemcpyDeviceToHost);
    const int np = 32768;
    std::vector<float> sorted((size_t)np * 4); hipMemcpy(sorted.data(), ix, sorted.size() * 4, hipMemcpyDeviceToHost);
    int dbg[4096]; hipMemcpyFromSymbol(dbg, HIP_SYMBOL(epnet::pruned::g_dbg), sizeof(dbg));
    printf("idx:"); for (int i = 0; i < m; ++i) printf(" %d", idx[i]); printf("\n");
    // true per-bucket maxima from temp
    float gmax = -1; int gb = -1;
    for (int b = 0; b < np / 64; ++b) {
        float mx = -1; int kk = -1;
        for (int l = 0; l < 64; ++l) { int k; memcpy(&k, &sorted[((size_t)b * 64 + l) * 4 + 3], 4); if (k >= 0 && temp[k] > mx) { mx = temp[k]; kk = k; } }
        float bm; memcpy(&bm, &dbg[b], 4);
        if (mx > gmax) { gmax = mx; gb = b; }
        if (bm != mx) printf("bucket %d: summary bm %g, true max %g (k=%d)\n", b, bm, mx, kk);
    }
    float bmg; memcpy(&bmg, &dbg[gb], 4);
    printf("global max %g in bucket %d (wave %d lane %d): summary bm %g brank %d publisher %d wbest %g\n", gmax, gb, gb >> 6, gb & 63, bmg, dbg[1024 + gb], dbg[2048 + gb], *(float *)&dbg[3072 + gb]);
    for (int b = 0; b < np / 64; ++b) { float bm; memcpy(&bm, &dbg[b], 4); if (bm == gmax) printf("  bucket %d (wave %d lane %d) bm==max brank %d publisher %d\n", b, b >> 6, b & 63, dbg[1024 + b], dbg[2048 + b]); }
    for (int q = 0; q < 1024; ++q) if (dbg[2048 + q]) printf("  publisher thread %d (wave %d) bm %g brank %d wbest %g\n", q, q >> 6, *(float *)&dbg[q], dbg[1024 + q], *(float *)&dbg[3072 + q]);
    for (int l = 0; l < 64; ++l) { int k; memcpy(&k, &sorted[((size_t)gb * 64 + l) * 4 + 3], 4); if (k >= 0 && temp[k] == gmax) printf("  holder lane %d k=%d rank16=%u\n", l, k, (unsigned)(((__builtin_bitreverse32(k & 1023) >> 22) << 6) | (k >> 10))); }
    return 0;
}
