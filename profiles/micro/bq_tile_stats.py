"""Work counters of the tiled ball query (diagnostic build: SRC=ball_query bash profiles/micro/build_variants.sh bqtstats
"-DEPNET_BQT_STATS", then EPNET_HIP_LIB=scratch/libs/lib_bqtstats.so python profiles/micro/bq_tile_stats.py [scenes])."""
import ctypes, sys
sys.path.insert(0, '.')
import torch
from epnet_amd import pointnet2_cuda as p2, synth, _lib
dev = 'cuda:0'
b = int(sys.argv[1]) if len(sys.argv) > 1 else 16
names = ["waves (buckets of 64 centres)", "candidate point buckets", "buckets scanned", "groups of 4 points", "groups with a hit", "hits of the larger ball",
         "crowded centres", "-", "cycles: walk", "cycles: lists out", "cycles: crowded balls"]
for n, m, radii, nss in ((16384, 4096, [0.1, 0.5], [16, 32]), (4096, 1024, [0.5, 1.0], [16, 32]), (65536, 16384, [0.5], [64])):
    xyz = synth.scenes("kitti", b, n, seed=3).to(dev)
    index = p2.scene_index(xyz)
    fidx = torch.empty((b, m), dtype=torch.int32, device=dev)
    centres = torch.empty((b, m, 3), device=dev)
    p2.sample_centres_wrapper(b, n, m, xyz, index, fidx, centres)
    ci = p2.scene_index(centres)
    outs = [torch.empty((b, m, ns), dtype=torch.int32, device=dev) for ns in nss]
    lib = _lib.lib()
    out = (ctypes.c_ulonglong * 16)()
    for _ in range(2):
        p2.ball_query_tiled_wrapper(b, n, m, radii, nss, centres, xyz, index, ci, outs)
    torch.cuda.synchronize()
    if hasattr(lib, "epnet_debug_bqt_stats"):
        lib.epnet_debug_bqt_stats(out)
    for which, fn in (("tiled", lambda: p2.ball_query_tiled_wrapper(b, n, m, radii, nss, centres, xyz, index, ci, outs)),
                      ("per centre", lambda: p2.ball_query_multi_wrapper(b, n, m, radii, nss, centres, xyz, index, outs))):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        print("ball query %d x %d r=%s, %d scenes, %s: %.4f ms" % (n, m, radii, b, which, e0.elapsed_time(e1) / 5))
    if hasattr(lib, "epnet_debug_bqt_stats"):
        lib.epnet_debug_bqt_stats(out)
        w = out[0] or 1
        for k, name in enumerate(names):
            print("   %-32s %12d  per wave %.2f" % (name, out[k], out[k] / w))
