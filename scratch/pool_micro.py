import torch, torch.nn.functional as F, sys
sys.path.insert(0, ".")
from epnet_amd import pointnet2_utils as p2u
dev = "cuda"
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for shape in [(2, 32, 4096, 16), (2, 64, 4096, 32), (2, 128, 1024, 32), (2, 512, 64, 32), (128, 128, 128, 64), (128, 512, 1, 32), (16, 64, 4096, 32)]:
    x = torch.randn(shape, device=dev).relu_().requires_grad_(True)
    g = torch.randn(shape[:3] + (1,), device=dev)
    a = lambda: F.max_pool2d(x, kernel_size=[1, shape[3]])
    b = lambda: p2u.pool_max(x)
    def bw(f):
        def run():
            x.grad = None
            f().backward(g)
        return run
    ta, tb = t(a), t(b)
    print(shape, "fwd stock %.3f hip %.3f ms (%.0f GB/s) | fwd+bwd stock %.3f hip %.3f ms  (input %.1f MB)" % (
        ta, tb, x.numel() * 4 / tb / 1e6, t(bw(a)), t(bw(b)), x.numel() * 4 / 1e6))
