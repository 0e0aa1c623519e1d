import sys, torch
sys.path.insert(0, ".")
from epnet_amd import pointnet2_cuda as ext, pointnet2_utils as p2u
d = "cuda"
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (b, ci, co, n, m, ns) in [(128, 128, 128, 512, 128, 64), (2, 96, 64, 4096, 1024, 32), (2, 256, 128, 1024, 256, 32), (128, 128, 128, 128, 32, 64)]:
    g = torch.Generator().manual_seed(0)
    xyz = (torch.rand((b, n, 3), generator=g) * 4).to(d); new_xyz = xyz[:, :m].contiguous()
    feats = torch.randn((b, ci, n), generator=g).to(d)
    idx = torch.randint(0, n, (b, m, ns), generator=g, dtype=torch.int32).to(d)
    w = (torch.randn((co, 3 + ci), generator=g) * 0.1).to(d)
    gy = torch.randn((b, co, m, ns), generator=g).to(d)
    z = torch.matmul(w[:, 3:], feats).contiguous()
    out = torch.empty((b, co, m, ns), device=d)
    grouped = torch.empty((b, 3 + ci, m, ns), device=d)
    gz = torch.zeros((b, co, n), device=d)
    dxyz = torch.empty((b, 3, m, ns), device=d)
    ggrouped = torch.empty((b, 3 + ci, m, ns), device=d); gfe = torch.zeros((b, ci, n), device=d)
    print((b, ci, co, n, m, ns))
    print("  fold  fwd: matmul %.3f  group_linear %.3f" % (t(lambda: torch.matmul(w[:, 3:], feats)), t(lambda: ext.group_linear_wrapper(b, co, n, m, ns, xyz, new_xyz, z, idx, w[:, :3].contiguous(), None, out))))
    print("  plain fwd: group_concat %.3f  conv %.3f" % (t(lambda: ext.group_concat_wrapper(b, ci, n, m, ns, xyz, new_xyz, feats, idx, grouped, True)), t(lambda: torch.nn.functional.conv2d(grouped, w[:, :, None, None]))))
    print("  fold  bwd: group_grad %.3f  dxyz %.3f  einsum %.3f  matmul-bwd(dW %.3f dF %.3f)" % (
        t(lambda: (gz.zero_(), ext.group_points_grad_wrapper(b, co, n, m, ns, gy, idx, gz))),
        t(lambda: ext.group_concat_wrapper(b, 0, n, m, ns, xyz, new_xyz, None, idx, dxyz, True)),
        t(lambda: torch.einsum("bcp,bkp->ck", gy.view(b, co, -1), dxyz.view(b, 3, -1))),
        t(lambda: torch.einsum("bcn,bkn->ck", gz, feats)), t(lambda: torch.matmul(w[:, 3:].t(), gz))))
    print("  plain bwd: conv dW %.3f  conv dX %.3f  group_concat_grad %.3f" % (
        t(lambda: torch.einsum("bcp,bkp->ck", gy.view(b, co, -1), grouped.view(b, 3 + ci, -1))),
        t(lambda: torch.matmul(w.t(), gy.view(b, co, -1))),
        t(lambda: (gfe.zero_(), ext.group_concat_grad_wrapper(b, ci, n, m, ns, ggrouped, idx, gfe, True)))))
