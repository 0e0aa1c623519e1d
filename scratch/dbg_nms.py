import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle
from epnet_amd import iou3d_cuda as ext, kitti_utils
fx = np.load("tests/golden/iou3d.npz")
bev = fx["bev_a"]; scores = fx["scores"]
order = np.argsort(-scores, kind="stable")
sb = np.ascontiguousarray(bev[order])
full_o = oracle.boxes_iou_bev(sb, sb)
d = torch.from_numpy(sb).cuda()
ans = torch.zeros((300,300), device="cuda")
ext.boxes_iou_bev_gpu(d, d, ans)
full_g = ans.cpu().numpy()
diff = np.abs(full_g - full_o)
print("max diff", diff.max(), np.unravel_index(diff.argmax(), diff.shape))
bad = np.argwhere(diff > 1e-5)
print("n bad", len(bad))
for a,b in bad[:10]:
    print(a,b, full_g[a,b], full_o[a,b], sb[a], sb[b])
for thr in (0.1, 0.5):
    flips = np.argwhere((full_g > thr) != (full_o > thr))
    print("thr", thr, "flips", flips[:10].tolist())
    keep, num = ext.nms_device(d, thr)
    n = int(num.item()); kg = keep[:n].cpu().numpy()
    ko = oracle.nms(sb, thr, True)
    print("gpu keep", kg.tolist()); print("ora keep", ko.tolist())
    # sweep using GPU iou matrix on host
    alive = np.ones(300, bool); kk=[]
    for i in range(300):
        if alive[i]:
            kk.append(i); alive[i+1:] &= ~(full_g[i, i+1:] > thr)
    print("host-sweep over gpu iou", kk)
