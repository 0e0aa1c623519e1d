import sys
sys.path.insert(0, '.')
import torch
from epnet_amd import pointnet2_cuda as p2, synth
dev = 'cuda:0'
b, n, m = 64, 16384, 4096
unknown = synth.scenes("kitti", b, n, seed=3).to(dev)
kidx = torch.empty((b, m), dtype=torch.int32, device=dev)
known = torch.empty((b, m, 3), device=dev)
ui = p2.scene_index(unknown)
p2.sample_centres_wrapper(b, n, m, unknown, ui, kidx, known)
ki = p2.scene_index(known)
d2 = torch.empty((b, n, 3), device=dev); idx = torch.empty((b, n, 3), dtype=torch.int32, device=dev)
for _ in range(4):
    p2.three_nn_indexed_wrapper(b, n, m, unknown, known, ui, ki, d2, idx)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    p2.three_nn_indexed_wrapper(b, n, m, unknown, known, ui, ki, d2, idx)
e1.record(); torch.cuda.synchronize()
print("three_nn ms", e0.elapsed_time(e1) / 5)
