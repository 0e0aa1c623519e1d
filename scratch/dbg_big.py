import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from epnet_amd import pointnet2_cuda as ext, synth
from oracle import oracle
oracle.build()
b, n, m = 2, 30000, 700
xyz = synth.scenes("dup", b, n, seed=300 + n).numpy()
d = torch.from_numpy(xyz).cuda()
index = ext.scene_index(d)
temp = torch.full((b, n), 1e10, device="cuda"); idx = torch.empty((b, m), dtype=torch.int32, device="cuda")
ext.furthest_point_sampling_indexed_wrapper(b, n, m, d, index, temp, idx)
got = idx.cpu().numpy(); want, wtemp = oracle.furthest_point_sampling(xyz, m, return_temp=True)
for s in range(b):
    bad = np.nonzero(got[s] != want[s])[0]
    print("scene", s, "first mismatch at", bad[:5], "got", got[s][bad[:5]], "want", want[s][bad[:5]])
    j = bad[0]
    # distances at that round (float64 replay)
    p = xyz[s]
    t = np.full(n, 1e10, np.float32)
    for r in range(j):
        dd = ((p - p[want[s][r]]) ** 2).astype(np.float32)
        dist = (dd[:, 0] + dd[:, 1]) + dd[:, 2]
        t = np.minimum(t, dist)
    print("  max t", t.max(), "holders", np.nonzero(t == t.max())[0][:10], "count", (t == t.max()).sum())
    print("  distinct points", len(np.unique(p, axis=0)))
print("---- temps")
for mm in (19, 20):
    temp = torch.full((b, n), 1e10, device="cuda"); idx = torch.empty((b, mm), dtype=torch.int32, device="cuda")
    ext.furthest_point_sampling_indexed_wrapper(b, n, mm, d, index, temp, idx)
    want, wtemp = oracle.furthest_point_sampling(xyz, mm, return_temp=True)
    g = temp.cpu().numpy()
    bad = np.nonzero(g[0] != wtemp[0])[0]
    print("m", mm, "idx ok", np.array_equal(idx.cpu().numpy()[0], want[0]), "temp mismatches", len(bad), bad[:10], g[0][bad[:5]], wtemp[0][bad[:5]])
    for k in (7919, 17682, 21416):
        print("   t[%d] = %r want %r  xyz %r" % (k, g[0][k], wtemp[0][k], xyz[0][k]))
