"""Phase counters of the scene index build (diagnostic build: SRC=ball_query bash profiles/micro/build_variants.sh ixstats
"-DEPNET_IX_STATS", then EPNET_HIP_LIB=scratch/libs/lib_ixstats.so python profiles/micro/ix_stats.py [scenes])."""
import ctypes, sys
sys.path.insert(0, '.')
import torch
from epnet_amd import pointnet2_cuda as p2, synth, _lib
dev = 'cuda:0'
b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
names = ["load points + bounding box", "zero the histogram", "cell codes + histogram", "scan", "scatter through LDS + rows out + bucket boxes"]
for n in (16384, 4096, 1024):
    xyz = synth.scenes("kitti", b, n, seed=3).to(dev)
    index = torch.empty((p2.scene_index_bytes(b, n),), dtype=torch.uint8, device=dev)
    for _ in range(3):
        p2.scene_index_build_wrapper(b, n, xyz, index)
    torch.cuda.synchronize()
    lib = _lib.lib()
    out = (ctypes.c_ulonglong * 8)()
    if hasattr(lib, "epnet_debug_ix_stats"):
        lib.epnet_debug_ix_stats(out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        p2.scene_index_build_wrapper(b, n, xyz, index)
    e1.record(); torch.cuda.synchronize()
    print("scene index n=%d, %d scenes: %.4f ms" % (n, b, e0.elapsed_time(e1) / 5))
    if hasattr(lib, "epnet_debug_ix_stats"):
        lib.epnet_debug_ix_stats(out)
        w = out[7] or 1
        for k, name in enumerate(names):
            print("   %-28s %9.1f ticks per workgroup" % (name, out[k] / w))
