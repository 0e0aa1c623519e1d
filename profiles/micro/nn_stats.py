"""Work counters of the three_nn tile kernel (diagnostic build: SRC=interpolate bash profiles/micro/build_variants.sh nnstats
"-DEPNET_NN_STATS", then EPNET_HIP_LIB=scratch/libs/lib_nnstats.so python profiles/micro/nn_stats.py [scenes])."""
import ctypes, sys
sys.path.insert(0, '.')
import torch
from epnet_amd import pointnet2_cuda as p2, synth, _lib
dev = 'cuda:0'
b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
names = ["waves", "walk iterations", "buckets scanned", "offer4", "offer4 votes passed", "inserts run", "max scans of a wave", "sum scans^2"]
cur = synth.scenes("kitti", b, 16384, seed=3).to(dev)
for m in (4096, 1024, 256, 64):
    n = cur.shape[1]
    kidx = torch.empty((b, m), dtype=torch.int32, device=dev)
    known = torch.empty((b, m, 3), device=dev)
    ui = p2.scene_index(cur)
    p2.sample_centres_wrapper(b, n, m, cur, ui, kidx, known)
    ki = p2.scene_index(known)
    d2 = torch.empty((b, n, 3), device=dev); idx = torch.empty((b, n, 3), dtype=torch.int32, device=dev)
    for _ in range(3):
        p2.three_nn_indexed_wrapper(b, n, m, cur, known, ui, ki, d2, idx)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 16)()
    lib = _lib.lib()
    if hasattr(lib, "epnet_debug_nn_stats"):
        lib.epnet_debug_nn_stats(out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        p2.three_nn_indexed_wrapper(b, n, m, cur, known, ui, ki, d2, idx)
    e1.record(); torch.cuda.synchronize()
    print("three_nn %d x %d, %d scenes: %.4f ms" % (n, m, b, e0.elapsed_time(e1) / 5))
    if hasattr(lib, "epnet_debug_nn_stats"):
        lib.epnet_debug_nn_stats(out)
        w = out[0] or 1
        for k, name in enumerate(names):
            print("   %-22s %12d  per wave %.2f" % (name, out[k], out[k] / w if k not in (6,) else out[k]))
        print("   scans per wave, bins of 4:", [out[8 + k] for k in range(8)])
    cur = known
