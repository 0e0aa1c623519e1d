"""LI-Fusion's point-to-pixel sampler (reference: lib/net/pointnet2_msg.py:107-120 ``Feature_Gather`` and the xy gather of
:214-217) -- SURVEY.md section 8(f) row N4, the consumer of the FPS indices on the image side.

``Feature_Gather(feature_map, xy)`` has the reference's name, arguments and result: feature_map (B,C,H,W), xy (B,N,2)
normalised to [-1,1] -> (B,C,N), the values of ``torch.nn.functional.grid_sample(feature_map, xy.unsqueeze(1))``
(bilinear, zero padding). Optional extras: ``idx`` (B,N) int32 = the FPS indices of the SA level, which folds the
reference's ``torch.gather(l_xy_cor[i], 1, li_index)`` into the same kernel and returns the picked coordinates as well;
``align_corners`` (default True: the reference was written for torch <= 1.2, whose grid_sample had no such switch and
behaved that way). Differentiable w.r.t. the feature map (the pixel coordinates are data, as in the reference).
"""
import torch
from torch.autograd import Function

from . import pointnet2_cuda as _ext


class _FeatureGather(Function):
    @staticmethod
    def forward(ctx, feature_map, xy, idx, align_corners):
        feature_map, xy = feature_map.contiguous(), xy.contiguous()
        b, c, h, w = feature_map.shape
        n_src = xy.shape[1]
        n = idx.shape[1] if idx is not None else n_src
        out = torch.empty((b, c, n), dtype=torch.float32, device=feature_map.device)
        xy_sel = torch.empty((b, n, 2), dtype=torch.float32, device=feature_map.device) if idx is not None else None
        _ext.feature_gather_wrapper(b, c, h, w, n_src, n, align_corners, feature_map, xy, idx, out, xy_sel)
        ctx.save_for_backward(xy if idx is None else xy_sel)
        ctx.dims = (b, c, h, w, n, align_corners)
        ctx.mark_non_differentiable(*([xy_sel] if xy_sel is not None else []))
        return (out, xy_sel) if idx is not None else out

    @staticmethod
    def backward(ctx, grad_out, *unused):
        (xy,) = ctx.saved_tensors
        b, c, h, w, n, align_corners = ctx.dims
        grad_map = None
        if ctx.needs_input_grad[0]:
            grad_map = torch.zeros((b, c, h, w), dtype=torch.float32, device=grad_out.device)
            _ext.feature_gather_grad_wrapper(b, c, h, w, n, align_corners, grad_out.detach().contiguous(), xy, grad_map)
        return grad_map, None, None, None


def Feature_Gather(feature_map, xy, idx=None, align_corners=True):
    """feature_map (B,C,H,W), xy (B,N,2) in [-1,1] -> (B,C,N); with idx (B,M) int32: samples at xy[b, idx[b, m]] and
    returns ((B,C,M), picked xy (B,M,2))"""
    return _FeatureGather.apply(feature_map, xy, idx, bool(align_corners))
