import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle
from epnet_amd import pointnet2_utils as p2u, synth
for (b, n, m, kind) in [(1, 256, 64, "kitti"), (1, 512, 128, "kitti"), (1, 1024, 256, "kitti"), (1, 2048, 300, "kitti"), (1, 4096, 1024, "kitti"), (1, 8192, 512, "kitti"), (1, 16384, 600, "kitti"), (1,1000,300,"kitti"), (1,100,40,"ubox")]:
    xyz = synth.scenes(kind, b, n, seed=100 + n).numpy()
    got = p2u.furthest_point_sample(torch.from_numpy(xyz).cuda(), m).cpu().numpy()
    ref = oracle.furthest_point_sampling(xyz, m)
    bad = np.argwhere(got != ref)
    print(n, m, "ok" if len(bad) == 0 else ("first mismatch at %s: got %s ref %s" % (bad[0], got[0, bad[0][1]:bad[0][1]+6], ref[0, bad[0][1]:bad[0][1]+6])), flush=True)
