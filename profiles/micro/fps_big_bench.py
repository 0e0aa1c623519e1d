"""furthest point sampling 65536 -> 16384 over the scene index: python profiles/micro/fps_big_bench.py [scenes ...]
(EPNET_FPS_BIG_WAVES = 4 | 8 | 16 picks the workgroup shape of fps_bigscene_kernel)"""
import os, sys
sys.path.insert(0, '.')
import torch
from epnet_amd import pointnet2_cuda as p2, synth
dev = 'cuda:0'
for b in [int(x) for x in sys.argv[1:]] or [1, 16, 256]:
    xyz = torch.stack([synth.kitti_like_cloud(65536, 3 + (i % 8)) for i in range(b)]).to(dev)
    index = p2.scene_index(xyz)
    idx = torch.empty((b, 16384), dtype=torch.int32, device=dev)
    ctr = torch.empty((b, 16384, 3), device=dev)
    p2.sample_centres_wrapper(b, 65536, 16384, xyz, index, idx, ctr)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        p2.sample_centres_wrapper(b, 65536, 16384, xyz, index, idx, ctr)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print("waves %s scenes %4d: %.3f ms  (%.3f us per round)  checksum %d" % (os.environ.get("EPNET_FPS_BIG_WAVES", "16"), b, ms, ms * 1e3 / 16383, int(idx.long().sum())))
