"""ball query in centre order (epnet_ball_query_indexed_multi) against the centres' own spatial order (epnet_ball_query_ordered):
python profiles/micro/bq_order_bench.py [scenes]"""
import sys
sys.path.insert(0, '.')
import torch
from epnet_amd import pointnet2_cuda as p2, synth
dev = 'cuda:0'
b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for n, m, radii, nss in ((16384, 4096, [0.1, 0.5], [16, 32]), (4096, 1024, [0.5, 1.0], [16, 32]), (65536, 16384, [0.5], [64])):
    xyz = torch.stack([synth.kitti_like_cloud(n, 3 + (i % 16)) for i in range(b)]).to(dev)
    index = p2.scene_index(xyz)
    fidx = torch.empty((b, m), dtype=torch.int32, device=dev)
    centres = torch.empty((b, m, 3), device=dev)
    p2.sample_centres_wrapper(b, n, m, xyz, index, fidx, centres)
    ci = p2.scene_index(centres)
    outs = [torch.empty((b, m, ns), dtype=torch.int32, device=dev) for ns in nss]
    outs2 = [torch.empty((b, m, ns), dtype=torch.int32, device=dev) for ns in nss]
    res = {}
    for which, fn in (("centre order", lambda: p2.ball_query_multi_wrapper(b, n, m, radii, nss, centres, xyz, index, outs)),
                      ("spatial order", lambda: p2.ball_query_ordered_wrapper(b, n, m, radii, nss, centres, xyz, index, ci, outs2))):
        fn(); fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        res[which] = e0.elapsed_time(e1) / 5
    same = all(torch.equal(a, c) for a, c in zip(outs, outs2))
    print("ball query %d x %d r=%s, %d scenes: centre order %.4f ms, spatial order %.4f ms, identical %s" % (n, m, radii, b, res["centre order"], res["spatial order"], same))
