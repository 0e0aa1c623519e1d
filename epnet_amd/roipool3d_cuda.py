"""Stand-in for the reference's ``roipool3d_cuda`` extension module (lib/utils/roipool3d/src/roipool3d.cpp:198-203)."""
import torch

from . import _lib
from ._tensor import dev_ptr, host_ptr, need, on_device_of, writes

_F = torch.float32


@writes("pooled_features", "pooled_empty_flag")
def forward(xyz, boxes3d, pts_feature, pooled_features, pooled_empty_flag):
    """roipool3d_gpu, roipool3d.cpp:48-79 (S is read from pooled_features.size(2), :64)"""
    px, pb, pf = dev_ptr(xyz, "xyz", _F), dev_ptr(boxes3d, "boxes3d", _F), dev_ptr(pts_feature, "pts_feature", _F)
    po, pe = dev_ptr(pooled_features, "pooled_features", _F), dev_ptr(pooled_empty_flag, "pooled_empty_flag", torch.int32)
    b, n = xyz.size(0), xyz.size(1)
    m, c, s_num = boxes3d.size(1), pts_feature.size(2), pooled_features.size(2)
    need(xyz, b * n * 3, "xyz"); need(boxes3d, b * m * 7, "boxes3d"); need(pts_feature, b * n * c, "pts_feature")
    need(pooled_features, b * m * s_num * (3 + c), "pooled_features"); need(pooled_empty_flag, b * m, "pooled_empty_flag")
    with on_device_of(xyz) as s:
        _lib.check(_lib.lib().epnet_roipool3d(b, n, m, c, s_num, px, pb, pf, po, pe, None, 0, s), "roipool3d")
    return 1


# forward_slow (roipool3d.cpp:15-44) computes the same result with one thread per box; same entry here
forward_slow = forward


@writes("pts_flag")
def pts_in_boxes3d_cpu(pts_flag, pts, boxes3d):
    """roipool3d.cpp:97-125 -- a HOST op in the reference as well"""
    pf, pp, pb = host_ptr(pts_flag, "pts_flag", torch.int64), host_ptr(pts, "pts", _F), host_ptr(boxes3d, "boxes3d", _F)
    m, n = boxes3d.size(0), pts.size(0)
    need(pts_flag, m * n, "pts_flag"); need(pts, n * 3, "pts"); need(boxes3d, m * 7, "boxes3d")
    _lib.check(_lib.lib().epnet_pts_in_boxes3d_host(pf, pp, pb, m, n), "pts_in_boxes3d_cpu")
    return 1


@writes("pooled_pts", "pooled_features", "pooled_empty_flag")
def roipool3d_cpu(pts, boxes3d, pts_feature, pooled_pts, pooled_features, pooled_empty_flag):
    """roipool3d.cpp:127-195 -- a HOST op in the reference as well"""
    pp, pb, pf = host_ptr(pts, "pts", _F), host_ptr(boxes3d, "boxes3d", _F), host_ptr(pts_feature, "pts_feature", _F)
    op, of = host_ptr(pooled_pts, "pooled_pts", _F), host_ptr(pooled_features, "pooled_features", _F)
    oe = host_ptr(pooled_empty_flag, "pooled_empty_flag", torch.int64)
    m, n, c, s = boxes3d.size(0), pts.size(0), pts_feature.size(1), pooled_pts.size(1)
    need(pts, n * 3, "pts"); need(boxes3d, m * 7, "boxes3d"); need(pts_feature, n * c, "pts_feature")
    need(pooled_pts, m * s * 3, "pooled_pts"); need(pooled_features, m * s * c, "pooled_features"); need(pooled_empty_flag, m, "pooled_empty_flag")
    _lib.check(_lib.lib().epnet_roipool3d_host(pp, pb, pf, op, of, oe, m, n, c, s), "roipool3d_cpu")
    return 1
