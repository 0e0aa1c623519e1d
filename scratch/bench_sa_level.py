import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from epnet_amd import pointnet2_utils as p2u, synth
dev = "cuda"
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts)//2]
for B in (2, 16):
    xyz = synth.scenes("kitti", B, 4096, seed=3).to(dev); new_xyz = xyz[:, :1024].contiguous()
    feats = torch.randn((B, 96, 4096), device=dev)
    qg = p2u.QueryAndGroup(1.0, 32)
    def fused():
        with torch.no_grad(): return qg(xyz, new_xyz, feats)
    def unfused():
        with torch.no_grad():
            idx = p2u.ball_query(1.0, 32, xyz, new_xyz)
            gx = p2u.grouping_operation(xyz.transpose(1, 2).contiguous(), idx); gx -= new_xyz.transpose(1, 2).unsqueeze(-1)
            gf = p2u.grouping_operation(feats, idx)
            return torch.cat([gx, gf], dim=1)
    print("B=%d level-2 QueryAndGroup (C=96, N=4096, M=1024, ns=32): fused %.4f ms, reference composition %.4f ms" % (B, timeit(fused), timeit(unfused)))
