"""Phase cycle counters of runsum::scatter_kernel for three_interpolate_grad (diagnostic build:
SRC=interpolate bash profiles/micro/build_variants.sh rsstats "-DEPNET_RUNSUM_STATS", then
EPNET_HIP_LIB=scratch/libs/lib_rsstats.so python profiles/micro/runsum_stats.py). s_memtime ticks at 100 MHz."""
import ctypes, sys
sys.path.insert(0, '.')
import torch
from epnet_amd import pointnet2_cuda as p2, synth, _lib
dev = 'cuda:0'
b, c, n, m = 16, 256, 16384, 4096
unknown = synth.scenes("kitti", b, n, seed=3).to(dev)
kidx = torch.empty((b, m), dtype=torch.int32, device=dev)
known = torch.empty((b, m, 3), device=dev)
ui = p2.scene_index(unknown)
p2.sample_centres_wrapper(b, n, m, unknown, ui, kidx, known)
d2 = torch.empty((b, n, 3), device=dev); idx = torch.empty((b, n, 3), dtype=torch.int32, device=dev)
p2.three_nn_wrapper(b, n, m, unknown, known, d2, idx)
inv = 1.0 / (torch.sqrt(d2) + 1e-8)
w = (inv / inv.sum(dim=2, keepdim=True)).contiguous()
go = torch.randn((b, c, n), device=dev)
gp = torch.zeros((b, c, m), device=dev)
for _ in range(3):
    gp.zero_(); p2.three_interpolate_grad_wrapper(b, c, n, m, go, idx, w, gp)
torch.cuda.synchronize()
lib = _lib.lib()
out = (ctypes.c_ulonglong * 16)()
has = hasattr(lib, "epnet_debug_runsum_stats")
if has:
    lib.epnet_debug_runsum_stats(out)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    p2.three_interpolate_grad_wrapper(b, c, n, m, go, idx, w, gp)
e1.record(); torch.cuda.synchronize()
print("three_interpolate_grad ms", e0.elapsed_time(e1) / 5)
if has:
    lib.epnet_debug_runsum_stats(out)
    rows = out[5] or 1
    for k, name in enumerate(["copy rows to LDS + barrier", "entry loop", "barrier after loop", "head / far links + barrier", "output rows"]):
        print("   %-28s first wave %8.1f   last wave %8.1f   ticks of s_memtime per row pass (x %d row passes)"
              % (name, out[k] / rows, out[8 + k] / rows, rows))
    if hasattr(lib, "epnet_debug_pack_stats"):
        pk = (ctypes.c_ulonglong * 8)()
        lib.epnet_debug_pack_stats(pk)
        wgs = pk[4] or 1
        for k, name in enumerate(["count pass", "scans", "placing pass"]):
            print("   pack: %-20s %8.1f ticks per workgroup (x %d)" % (name, pk[k] / wgs, wgs))
