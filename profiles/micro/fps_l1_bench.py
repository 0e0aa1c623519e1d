"""level-1 furthest point sampling 16384 -> 4096 over the scene index, with the sampling chain's tie watch (prefix_out, cap 1024):
python profiles/micro/fps_l1_bench.py [scenes ...]   (EPNET_HIP_LIB picks the build)"""
import os, sys
sys.path.insert(0, '.')
import torch
from epnet_amd import pointnet2_cuda as p2, synth
dev = 'cuda:0'
for b in [int(x) for x in sys.argv[1:]] or [1, 256]:
    xyz = torch.stack([synth.kitti_like_cloud(16384, 3 + i) for i in range(b)]).to(dev)
    index = p2.scene_index(xyz)
    idx = torch.empty((b, 4096), dtype=torch.int32, device=dev)
    prefix = torch.zeros((b,), dtype=torch.int32, device=dev)
    for with_watch in (True, False):
        fn = (lambda: p2.sample_centres_wrapper(b, 16384, 4096, xyz, index, idx, None, None, prefix, 1024)) if with_watch else \
             (lambda: p2.furthest_point_sampling_indexed_wrapper(b, 16384, 4096, xyz, index, torch.full((b, 16384), 1e10, device=dev), idx))
        fn(); fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print("%s scenes %4d tie watch %d: %.4f ms (%.4f us per round) checksum %d" % (os.environ.get("EPNET_HIP_LIB", "current"), b, with_watch, ms, ms * 1e3 / 4095, int(idx.long().sum())))
