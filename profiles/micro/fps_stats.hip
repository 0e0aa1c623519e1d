// diagnostic build of the pruned FPS kernel with phase counters (not product)
#define EPNET_FPS_STATS 1
#include "../epnet_amd/csrc/fps.hip"
#include "../epnet_amd/csrc/host.cpp"
#include <cstdio>
#include <vector>
#include <cmath>
#include <random>
int main(int argc, char **argv) {
    const char *path = argc > 1 ? argv[1] : nullptr;
    struct Cfg { int n, m; } cfgs[] = {{16384, 4096}, {4096, 1024}, {8192, 2048}, {2048, 512}};
    for (auto c : cfgs) {
        std::vector<float> xyz((size_t)c.n * 3);
        bool loaded = false;
        if (path) { FILE *f = fopen(path, "rb"); if (f) { loaded = fread(xyz.data(), 4, xyz.size(), f) == xyz.size(); fclose(f); } }
        if (!loaded) { std::mt19937 g(1); std::uniform_real_distribution<float> ux(-40, 40), uy(-1, 3), uz(0, 70.4);
            for (int i = 0; i < c.n; ++i) { xyz[i*3] = ux(g); xyz[i*3+1] = uy(g); xyz[i*3+2] = uz(g); } }
        float *dx; int *di; hipMalloc(&dx, xyz.size() * 4); hipMalloc(&di, c.m * 4);
        hipMemcpy(dx, xyz.data(), xyz.size() * 4, hipMemcpyHostToDevice);
        unsigned long long z[16] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(epnet::pruned::g_stats), z, sizeof(z));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        int rc = epnet_furthest_point_sampling(1, c.n, c.m, dx, nullptr, di, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long st[16];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(epnet::pruned::g_stats), sizeof(st));
        const double it = c.m - 1, waves = c.n > 8192 ? 8 : 4;
        printf("n=%d m=%d rc=%d (%s): %.3f ms (%.3f us/iter). active buckets/iter (all waves) %.2f | cycles/iter/wave: bounds %.0f update %.0f best+resolve %.0f publish+barrier %.0f post %.0f | loop total %.0f cyc/iter\n",
               c.n, c.m, rc, loaded ? "file" : "ubox", ms, ms * 1e3 / it, st[0] / it, st[1] / it / waves, st[2] / it / waves, st[3] / it / waves, st[4] / it / waves, st[5] / it / waves, st[6] / it / waves);
        hipFree(dx); hipFree(di);
    }
    return 0;
}
