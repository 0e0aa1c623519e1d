"""The neighbourhood max-pool of an SA level (epnet_pool_max / epnet_pool_max_grad, csrc/pool.hip) against the op the
reference calls, F.max_pool2d(kernel_size=[1, nsample]) (pointnet2_lib/pointnet2/pointnet2_modules.py:61-68) -- stock
PyTorch, so the reference itself runs here: the oracle's numpy restatement is pinned against it on the CPU, the HIP
kernels against both on the GPU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

SHAPES = [(2, 64, 128, 32), (1, 32, 100, 16), (3, 8, 33, 64), (2, 5, 7, 4), (1, 4, 9, 128), (1, 3, 5, 256), (2, 16, 8, 20), (1, 7, 3, 1),
          (1, 2, 3, 300), (4, 6, 1, 8), (1, 1, 1, 70)]


def data(shape, seed, ties):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(shape, generator=g)
    if ties:   # what the pool really sees: ReLU output (runs of zeros) and repeated columns (ball-query padding)
        x = torch.relu(x)
        x[..., shape[-1] // 2:] = x[..., :1]
    return x


@pytest.mark.parametrize("shape", SHAPES)
def test_oracle_pool_equals_the_stock_op(oracle, shape):
    for ties in (False, True):
        x = data(shape, 3, ties)
        want = F.max_pool2d(x, kernel_size=[1, shape[-1]])
        v, a = oracle.pool_max(x.numpy())
        np.testing.assert_array_equal(v, want.numpy())
        assert np.array_equal(np.take_along_axis(x.numpy(), a[..., None].astype(np.int64), -1), v)
        # first position of the maximum == the index the stock op records
        _, idx = F.max_pool2d(x, kernel_size=[1, shape[-1]], return_indices=True)
        np.testing.assert_array_equal(a, idx.squeeze(-1).numpy() % shape[-1])


@pytest.mark.gpu
@pytest.mark.parametrize("shape", SHAPES)
def test_pool_forward_and_backward_equal_the_stock_op(hiplib, oracle, shape):
    from epnet_amd import pointnet2_utils as p2u
    for ties in (False, True):
        x = data(shape, 5, ties).cuda().requires_grad_(True)
        got = p2u.pool_max(x)
        want = F.max_pool2d(x, kernel_size=[1, shape[-1]])
        assert torch.equal(got, want)
        g = torch.randn(want.shape, generator=torch.Generator().manual_seed(9)).cuda()
        gx, = torch.autograd.grad(got, x, g)
        wx, = torch.autograd.grad(want, x, g)
        assert torch.equal(gx, wx)     # both route the gradient to the first position of the maximum
        np.testing.assert_array_equal(gx.cpu().numpy(), oracle.pool_max_grad(g.cpu().numpy().squeeze(-1), oracle.pool_max(x.detach().cpu().numpy())[1], shape[-1]))


@pytest.mark.gpu
def test_pool_without_grad_and_nan(hiplib):
    from epnet_amd import pointnet2_utils as p2u
    x = torch.randn((2, 3, 4, 32)).cuda()
    assert torch.equal(p2u.pool_max(x), F.max_pool2d(x, kernel_size=[1, 32]))
    x[0, 0, 0, 5] = float("nan")
    out = p2u.pool_max(x)
    assert torch.isnan(out[0, 0, 0, 0]) and torch.isfinite(out[0, 0, 1:]).all()   # NaN propagates, as in the stock op
    big = torch.randn((128, 128, 128, 64), generator=torch.Generator().manual_seed(1)).cuda()   # the RCNN stage's shape
    assert torch.equal(p2u.pool_max(big), F.max_pool2d(big, kernel_size=[1, 64]))


@pytest.mark.gpu
def test_sa_module_uses_the_pool_kernel(hiplib, monkeypatch):
    """the SA module's forward and backward are unchanged by the kernel: same outputs and parameter gradients as with
    the stock pooling op"""
    from epnet_amd import pointnet2_modules as p2m, pointnet2_utils as p2u, synth
    torch.manual_seed(0)
    sa = p2m.PointnetSAModuleMSG(npoint=256, radii=[0.5, 1.0], nsamples=[16, 32], mlps=[[8, 16, 32], [8, 16, 32]]).cuda()
    xyz = synth.scenes("kitti", 2, 2048, seed=3).cuda()
    feats = torch.randn((2, 8, 2048), generator=torch.Generator().manual_seed(4)).cuda().requires_grad_(True)
    calls = []
    real = p2u.pool_max
    monkeypatch.setattr(p2u, "pool_max", lambda x: (calls.append(tuple(x.shape)), real(x))[1])
    _, out, _ = sa(xyz, feats)
    out.sum().backward()
    assert calls == [(2, 32, 256, 16), (2, 32, 256, 32)]
    grads = [p.grad.clone() for p in sa.parameters()] + [feats.grad.clone()]
    for p in sa.parameters():
        p.grad = None
    feats.grad = None
    monkeypatch.setattr(p2u, "pool_max", lambda x: F.max_pool2d(x, kernel_size=[1, x.size(3)]))
    _, out2, _ = sa(xyz, feats)
    out2.sum().backward()
    assert torch.equal(out, out2)
    # the pooled gradients are identical (test above); the dense layers' weight-gradient kernels accumulate with atomics,
    # so two backward passes of the same module agree only to rounding
    for a, b in zip(grads, [p.grad for p in sa.parameters()] + [feats.grad]):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)
