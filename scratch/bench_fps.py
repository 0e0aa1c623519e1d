import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from epnet_amd import pointnet2_cuda as ext, synth, _lib
print("lib:", _lib.LIB_PATH, "wide:", os.environ.get("EPNET_FPS_WIDE"))
dev = "cuda"
def timeit(b, n, m, reps=5, indexed=False):
    xyz = synth.scenes("kitti", b, n, seed=3).to(dev)
    temp = torch.empty((b, n), device=dev); idx = torch.empty((b, m), dtype=torch.int32, device=dev)
    index = ext.scene_index(xyz) if indexed else None
    def run():
        if indexed:
            ext.furthest_point_sampling_indexed_wrapper(b, n, m, xyz, index, temp, idx)
        else:
            ext.furthest_point_sampling_wrapper(b, n, m, xyz, temp, idx)
    for _ in range(2):
        temp.fill_(1e10); run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        temp.fill_(1e10)
        e0.record(); run(); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); med = ts[len(ts)//2]
    print("%s B=%3d N=%6d M=%5d  %.3f ms  %.3f us/iter" % ("indexed" if indexed else "plain  ", b, n, m, med, med*1e3/max(1,m-1)), flush=True)
    return idx
for b in (1, 16, 256):
    for n, m in ((16384, 4096), (4096, 1024), (1024, 256), (256, 64)):
        timeit(b, n, m)
        if n > 1024: timeit(b, n, m, indexed=True)
timeit(64, 512, 128); timeit(1024, 512, 128)
