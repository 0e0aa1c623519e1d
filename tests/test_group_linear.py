"""The first shared-MLP layer folded into the grouping (epnet_group_linear):
W . [xyz[idx] - centre ; F[:, idx]] = W_xyz . (xyz[idx] - centre) + (W_f . F)[:, idx]. Checked against the oracle's
restatement (exact), against the reference's composition -- grouped tensor, then a 1x1 convolution (pointnet2_utils.py:
250-257, pointnet2_modules.py:61) -- to dense-product rounding, and through the SA module with the folding on and off
(outputs, parameter gradients, feature gradients). The reference's own SA-module fixture runs through the folded path in
tests/test_surface_cpu.py (CPU, oracle stand-ins) and tests/test_gpu_parity.py (GPU)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

CASES = [(2, 64, 4096, 1024, 32, False), (3, 16, 1024, 128, 16, True), (1, 8, 700, 37, 8, False), (2, 5, 300, 20, 6, True),
         (1, 40, 2048, 64, 64, True), (2, 128, 512, 128, 64, False), (1, 3, 64, 8, 4, False), (1, 96, 20000, 256, 16, False)]


def inputs(b, c, n, m, ns, bias, seed=0):
    g = torch.Generator().manual_seed(seed + c + n)
    xyz = torch.rand((b, n, 3), generator=g) * torch.tensor([80.0, 4.0, 70.0]) - torch.tensor([40.0, 1.0, 0.0])
    new_xyz = xyz[:, :m].contiguous()
    feats = torch.randn((b, c, n), generator=g)
    idx = torch.randint(0, n, (b, m, ns), generator=g, dtype=torch.int32)
    w = torch.randn((c, 3 + c), generator=g) * 0.3
    bv = torch.randn((c,), generator=g) if bias else None
    return xyz, new_xyz, feats, idx, w, bv


@pytest.mark.gpu
@pytest.mark.parametrize("b,c,n,m,ns,bias", CASES)
def test_group_linear_equals_oracle_and_the_composition(hiplib, oracle, b, c, n, m, ns, bias):
    from epnet_amd import pointnet2_cuda as ext, pointnet2_utils as p2u
    xyz, new_xyz, feats, idx, w, bv = inputs(b, c, n, m, ns, bias)
    d = "cuda"
    z = torch.matmul(w[:, 3:].to(d), feats.to(d)).contiguous()
    got = p2u.group_linear(xyz.to(d), new_xyz.to(d), z, idx.to(d), w[:, :3].to(d), None if bv is None else bv.to(d))
    want = oracle.group_linear(xyz.numpy(), new_xyz.numpy(), z.cpu().numpy(), idx.numpy(), w[:, :3].numpy(), None if bv is None else bv.numpy())
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    # the reference's order of operations: build [dxyz ; grouped features], convolve
    grouped = torch.empty((b, 3 + c, m, ns), device=d)
    ext.group_concat_wrapper(b, c, n, m, ns, xyz.to(d), new_xyz.to(d), feats.to(d), idx.to(d), grouped, True)
    ref = F.conv2d(grouped, w.to(d)[:, :, None, None], None if bv is None else bv.to(d))
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 2e-6 * scale + 1e-6, (float((got - ref).abs().max()), scale)


@pytest.mark.gpu
@pytest.mark.parametrize("bn", [True, False])
def test_sa_module_folded_equals_unfolded(hiplib, bn):
    """PointnetSAModuleMSG with the first layer folded into the grouping vs the reference's composition (grouped tensor,
    then the whole MLP): outputs, parameter gradients and feature gradients"""
    from epnet_amd import pointnet2_modules as p2m, synth
    torch.manual_seed(1)
    sa = p2m.PointnetSAModuleMSG(npoint=512, radii=[0.5, 1.0], nsamples=[16, 32], mlps=[[24, 32, 48], [24, 32, 64]], bn=bn).cuda()
    xyz = synth.scenes("kitti", 2, 4096, seed=5).cuda()
    f0 = torch.randn((2, 24, 4096), generator=torch.Generator().manual_seed(6)).cuda()
    results = []
    for fold in (True, False):
        sa.fold_first_layer = fold
        sa.zero_grad(set_to_none=True)
        feats = f0.clone().requires_grad_(True)
        new_xyz, out, idx = sa(xyz, feats)
        (out * torch.linspace(0.5, 1.5, out.shape[1], device="cuda")[None, :, None]).sum().backward()
        results.append((new_xyz, out.detach(), idx, [p.grad.clone() for p in sa.parameters()], feats.grad.clone()))
    a, b = results
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])
    torch.testing.assert_close(a[1], b[1], rtol=1e-4, atol=1e-5)
    for ga, gb in zip(a[3], b[3]):
        torch.testing.assert_close(ga, gb, rtol=2e-3, atol=2e-4 * float(gb.abs().max()) + 1e-6)
    torch.testing.assert_close(a[4], b[4], rtol=2e-3, atol=2e-4 * float(b[4].abs().max()) + 1e-6)


@pytest.mark.gpu
def test_folding_applies_only_where_it_is_the_same_function(hiplib):
    from epnet_amd import pointnet2_modules as p2m, pointnet2_utils as p2u, pytorch_utils as ptu
    sa = p2m.PointnetSAModuleMSG(npoint=16, radii=[1.0], nsamples=[8], mlps=[[4, 8]]).cuda()
    feats = torch.randn((1, 4, 64)).cuda()
    assert p2m._foldable(sa.groupers[0], sa.mlps[0], feats)
    assert not p2m._foldable(sa.groupers[0], sa.mlps[0], None)                                   # xyz-only level (level 1 of the RPN)
    assert not p2m._foldable(p2u.GroupAll(), sa.mlps[0], feats)                                  # no neighbour lists
    assert not p2m._foldable(p2u.QueryAndGroup(1.0, 8, use_xyz=False), sa.mlps[0], feats)
    assert not p2m._foldable(sa.groupers[0], ptu.SharedMLP([7, 8], bn=True, preact=True), feats)   # bn / act before the convolution
    assert not p2m._foldable(sa.groupers[0], sa.mlps[0], torch.randn((1, 5, 64)).cuda())         # channel count does not match


@pytest.mark.gpu
@pytest.mark.parametrize("b,c,n,m,ns,bias", [(2, 16, 1024, 128, 16, True), (1, 8, 700, 37, 6, False), (3, 40, 512, 64, 64, True), (1, 9, 300, 7, 3, False)])
def test_group_linear_gradients_equal_the_composition(hiplib, b, c, n, m, ns, bias):
    """d/dz, d/dw_xyz, d/dbias of group_linear against autograd through [grouped xyz - centre ; grouped z] with an
    identity-on-z 1x1 convolution (float64 on the host as the yardstick)"""
    from epnet_amd import pointnet2_utils as p2u
    xyz, new_xyz, _, idx, w, bv = inputs(b, c, n, m, ns, bias, seed=3)
    g = torch.Generator().manual_seed(11)
    z = torch.randn((b, c, n), generator=g)
    gy = torch.randn((b, c, m, ns), generator=g)
    d = "cuda"
    zc, wc = z.to(d).requires_grad_(True), w[:, :3].contiguous().to(d).requires_grad_(True)
    bc = bv.to(d).requires_grad_(True) if bias else None
    out = p2u.group_linear(xyz.to(d), new_xyz.to(d), zc, idx.to(d), wc, bc)
    grads = torch.autograd.grad(out, [zc, wc] + ([bc] if bias else []), gy.to(d))
    # yardstick in float64
    z64, w64 = z.double().requires_grad_(True), w[:, :3].double().requires_grad_(True)
    b64 = bv.double().requires_grad_(True) if bias else None
    li = idx.long()
    dxyz = torch.stack([xyz[i][li[i]] for i in range(b)]).double() - new_xyz.double()[:, :, None, :]        # (b,m,ns,3)
    zg = torch.stack([z64[i][:, li[i]] for i in range(b)])                                                   # (b,c,m,ns)
    ref = zg + torch.einsum("ck,bmsk->bcms", w64, dxyz) + (b64[None, :, None, None] if bias else 0)
    want = torch.autograd.grad(ref, [z64, w64] + ([b64] if bias else []), gy.double())
    torch.testing.assert_close(out.detach().cpu().double(), ref.detach(), rtol=1e-5, atol=1e-5)
    for got, exp in zip(grads, want):
        torch.testing.assert_close(got.cpu().double(), exp, rtol=1e-4, atol=1e-4 * float(exp.abs().max()) + 1e-6)


@pytest.mark.gpu
def test_sampling_pyramid_gives_the_modules_their_own_results(hiplib):
    """pointnet2_utils.sample_pyramid (all levels' FPS on a side stream, up front) + presampled SA modules == the same
    modules sampling for themselves: identical indices, centres and features"""
    from epnet_amd import pointnet2_modules as p2m, pointnet2_utils as p2u, synth
    torch.manual_seed(2)
    sas = [p2m.PointnetSAModuleMSG(npoint=1024, radii=[0.5, 1.0], nsamples=[16, 32], mlps=[[0, 16, 32], [0, 16, 32]]).cuda().eval(),
           p2m.PointnetSAModuleMSG(npoint=256, radii=[1.0, 2.0], nsamples=[16, 32], mlps=[[64, 32, 64], [64, 32, 64]]).cuda().eval(),
           p2m.PointnetSAModuleMSG(npoint=64, radii=[2.0, 4.0], nsamples=[16, 32], mlps=[[128, 64, 64], [128, 64, 64]]).cuda().eval()]
    xyz = synth.scenes("kitti", 2, 8192, seed=8).cuda()

    def run(use_pyramid):
        pyr = p2u.sample_pyramid(xyz, [sa.npoint for sa in sas]) if use_pyramid else [None] * 3
        cur, feats, outs = xyz, None, []
        with torch.no_grad():
            for sa, pre in zip(sas, pyr):
                cur, feats, idx = sa(cur, feats, presampled=pre)
                outs.append((cur, feats, idx))
        torch.cuda.synchronize()
        return outs
    a, b = run(True), run(False)
    for (x1, f1, i1), (x2, f2, i2) in zip(a, b):
        assert torch.equal(i1, i2) and torch.equal(x1, x2) and torch.equal(f1, f2)
