"""Per-wave counters of the two-centres-per-wave ball query (diagnostic build: SRC=ball_query bash profiles/micro/build_variants.sh
bqstats "-DEPNET_BQ_STATS", then EPNET_HIP_LIB=scratch/libs/lib_bqstats.so python profiles/micro/bq_stats.py [scenes]): waves whose
balls all fit the 64-entry lists / waves that selected the nsample smallest of a longer list / waves that walked again (bitmap)."""
import ctypes, sys
sys.path.insert(0, '.')
import torch
from epnet_amd import pointnet2_cuda as p2, synth, _lib
dev = 'cuda:0'
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
lib = _lib.lib()
for n, m, radii, nss in ((16384, 4096, [0.1, 0.5], [16, 32]), (65536, 16384, [0.5], [64])):
    xyz = torch.stack([synth.kitti_like_cloud(n, 3 + (i % 16)) for i in range(b)]).to(dev)
    index = p2.scene_index(xyz)
    fidx = torch.empty((b, m), dtype=torch.int32, device=dev)
    centres = torch.empty((b, m, 3), device=dev)
    p2.sample_centres_wrapper(b, n, m, xyz, index, fidx, centres)
    ci = p2.scene_index(centres)
    outs = [torch.empty((b, m, ns), dtype=torch.int32, device=dev) for ns in nss]
    fn = (lambda: p2.ball_query_ordered_wrapper(b, n, m, radii, nss, centres, xyz, index, ci, outs))
    fn(); fn()
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 16)()
    if hasattr(lib, "epnet_debug_bq_stats"):
        lib.epnet_debug_bq_stats(out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record(); torch.cuda.synchronize()
    print("ball query n %d m %d r %s, %d scenes: %.4f ms" % (n, m, radii, b, e0.elapsed_time(e1) / 5))
    if hasattr(lib, "epnet_debug_bq_stats"):
        lib.epnet_debug_bq_stats(out)
        total = sum(out[o + 1] + out[o + 2] for o in (0, 4, 8)) or 1
        for o, name in ((0, "lists only"), (4, "selected"), (8, "walked again")):
            w = out[o] or 1
            print("   %-13s %8d waves (%.4f): walk %9.0f ticks, emission %9.0f ticks, %7.1f rows per wave; share of all wave time %.3f" % (
                name, out[o], out[o] / max(1, out[0] + out[4] + out[8]), out[o + 1] / w, out[o + 2] / w, out[o + 3] / w,
                (out[o + 1] + out[o + 2]) / total))
        print("   longest wave %d ticks" % out[12])
