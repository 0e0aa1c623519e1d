import sys, torch, torch.nn.functional as F
sys.path.insert(0, ".")
from epnet_amd.li_fusion import Feature_Gather
d = "cuda"
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (b, c, h, w, n) in [(2, 64, 192, 640, 4096), (2, 128, 96, 320, 1024), (2, 256, 48, 160, 256), (2, 512, 24, 80, 64), (2, 32, 384, 1280, 16384), (16, 64, 192, 640, 4096)]:
    img = torch.randn((b, c, h, w), device=d, requires_grad=True)
    xy = (torch.rand((b, n, 2), device=d) * 2 - 1)
    g = torch.randn((b, c, n), device=d)
    f = lambda: F.grid_sample(img, xy.unsqueeze(1), mode="bilinear", padding_mode="zeros", align_corners=True).squeeze(2)
    h_ = lambda: Feature_Gather(img, xy)
    def fb(fn):
        def run():
            img.grad = None
            fn().backward(g)
        return run
    print((b, c, h, w, n), "fwd stock %.3f hip %.3f | fwd+bwd stock %.3f hip %.3f ms" % (t(f), t(h_), t(fb(f)), t(fb(h_))))
