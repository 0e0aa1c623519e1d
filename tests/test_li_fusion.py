"""SURVEY.md section 8(f) row N4 -- the point-to-pixel sampler of LI-Fusion (epnet_feature_gather, csrc/sample.hip) against
the op the reference calls, torch.nn.functional.grid_sample (lib/net/pointnet2_msg.py:107-120): stock PyTorch, so the
reference's arithmetic itself is the yardstick, on the CPU and on the GPU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

SHAPES = [(2, 64, 48, 160, 4096), (2, 128, 24, 80, 1024), (1, 256, 12, 40, 256), (2, 512, 6, 20, 64), (1, 32, 96, 320, 16384), (3, 5, 7, 9, 33), (1, 1, 1, 1, 4)]


def make(b, c, h, w, n, seed=0):
    g = torch.Generator().manual_seed(seed + c + n)
    fmap = torch.randn((b, c, h, w), generator=g)
    xy = torch.rand((b, n, 2), generator=g) * 2.4 - 1.2          # some points fall outside the image: zero padding
    xy[:, :4] = torch.tensor([[-1.0, -1.0], [1.0, 1.0], [1.0, -1.0], [0.0, 0.0]])[:min(4, n)] if n >= 4 else xy[:, :4]
    return fmap, xy


@pytest.mark.gpu
@pytest.mark.parametrize("b,c,h,w,n", SHAPES)
@pytest.mark.parametrize("align", [True, False])
def test_feature_gather_equals_grid_sample(hiplib, b, c, h, w, n, align):
    from epnet_amd.li_fusion import Feature_Gather
    fmap, xy = make(b, c, h, w, n)
    want = F.grid_sample(fmap, xy.unsqueeze(1), mode="bilinear", padding_mode="zeros", align_corners=align).squeeze(2)   # CPU
    fm = fmap.cuda().requires_grad_(True)
    got = Feature_Gather(fm, xy.cuda(), align_corners=align)
    assert tuple(got.shape) == (b, c, n)
    # align_corners=True (the reference's behaviour) agrees to 1e-5. With False the pixel coordinate is ((x + 1) * W - 1) / 2:
    # one rounding of a value ~W, i.e. up to W * 6e-8 pixels between a fused and an unfused evaluation (torch's vectorised
    # CPU kernel fuses the multiply-add), times the image gradient -- the comparison is only as tight as that
    tol = 1e-5 if align else 1e-4
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.numpy(), rtol=tol, atol=tol)
    g = torch.randn((b, c, n), generator=torch.Generator().manual_seed(5))
    got_g, = torch.autograd.grad(got, fm, g.cuda())
    f64 = fmap.double().requires_grad_(True)
    ref = F.grid_sample(f64, xy.double().unsqueeze(1), mode="bilinear", padding_mode="zeros", align_corners=align).squeeze(2)
    want_g, = torch.autograd.grad(ref, f64, g.double())
    np.testing.assert_allclose(got_g.cpu().numpy(), want_g.numpy(), rtol=1e-4, atol=1e-4 if align else 1e-3)


@pytest.mark.gpu
def test_feature_gather_far_outside_and_non_finite_coordinates(hiplib):
    """points projected far outside the image, at infinity or NaN sample nothing: zeros (grid_sample's zero padding),
    and they scatter nothing in the backward"""
    from epnet_amd.li_fusion import Feature_Gather
    fmap = torch.randn((1, 3, 5, 7)).cuda().requires_grad_(True)
    xy = torch.tensor([[[0.0, 0.0], [50.0, 0.0], [0.0, -1e30], [float("inf"), 0.0], [float("nan"), 0.2], [0.3, float("nan")], [-1.0, 1.0]]]).cuda()
    out = Feature_Gather(fmap, xy)
    assert torch.equal(out[0, :, 1:6], torch.zeros((3, 5), device="cuda"))
    want = F.grid_sample(fmap.detach(), xy[:, [0, 6]].unsqueeze(1), mode="bilinear", padding_mode="zeros", align_corners=True).squeeze(2)
    torch.testing.assert_close(out[:, :, [0, 6]].detach(), want, rtol=1e-5, atol=1e-6)
    out[:, :, 1:6].sum().backward()
    assert float(fmap.grad.abs().sum()) == 0.0


@pytest.mark.gpu
def test_feature_gather_with_fps_indices(hiplib):
    """the reference's two steps -- torch.gather of xy over the FPS indices, then the sampler (:214-218) -- in one call"""
    from epnet_amd.li_fusion import Feature_Gather
    fmap, xy = make(2, 64, 48, 160, 4096, seed=3)
    idx = torch.stack([torch.randperm(4096, generator=torch.Generator().manual_seed(i))[:1024] for i in range(2)]).int()
    li_index = idx.long().unsqueeze(-1).repeat(1, 1, 2)
    xy_sel = torch.gather(xy, 1, li_index)
    want = F.grid_sample(fmap, xy_sel.unsqueeze(1), mode="bilinear", padding_mode="zeros", align_corners=True).squeeze(2)
    got, got_xy = Feature_Gather(fmap.cuda(), xy.cuda(), idx.cuda())
    assert torch.equal(got_xy.cpu(), xy_sel)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=1e-5, atol=1e-5)
    stock = F.grid_sample(fmap.cuda(), xy_sel.cuda().unsqueeze(1), mode="bilinear", padding_mode="zeros", align_corners=True).squeeze(2)
    np.testing.assert_allclose(got.cpu().numpy(), stock.cpu().numpy(), rtol=1e-5, atol=1e-5)


def test_feature_gather_host_logic_cpu(monkeypatch):
    """shapes, the folded xy gather, the returned coordinates and the autograd wiring of Feature_Gather, with the kernel
    replaced by the stock op it reproduces (tests/oracle_ext.py)"""
    import oracle_ext
    from epnet_amd import pointnet2_cuda
    p2, _, _ = oracle_ext.make_modules()
    for name in ("feature_gather_wrapper", "feature_gather_grad_wrapper"):
        monkeypatch.setattr(pointnet2_cuda, name, getattr(p2, name))
    from epnet_amd.li_fusion import Feature_Gather
    fmap, xy = make(2, 6, 12, 20, 50)
    fm = fmap.clone().requires_grad_(True)
    idx = torch.stack([torch.randperm(50, generator=torch.Generator().manual_seed(i))[:16] for i in range(2)]).int()
    out, sel = Feature_Gather(fm, xy, idx)
    want_xy = torch.gather(xy, 1, idx.long().unsqueeze(-1).repeat(1, 1, 2))
    assert torch.equal(sel, want_xy) and not sel.requires_grad and tuple(out.shape) == (2, 6, 16)
    ref_in = fmap.clone().requires_grad_(True)
    ref = F.grid_sample(ref_in, want_xy.unsqueeze(1), mode="bilinear", padding_mode="zeros", align_corners=True).squeeze(2)
    assert torch.equal(out, ref)
    g = torch.randn(out.shape, generator=torch.Generator().manual_seed(2))
    out.backward(g)
    ref.backward(g)
    torch.testing.assert_close(fm.grad, ref_in.grad)
    plain = Feature_Gather(fmap, xy)
    assert tuple(plain.shape) == (2, 6, 50)

