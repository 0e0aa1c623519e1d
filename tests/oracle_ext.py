"""TEST INFRASTRUCTURE: extension stand-ins backed by the CPU oracle, operating on CPU torch tensors.

Used (a) by tests/golden/make_golden.py to run the REFERENCE's Python surface on the CPU and (b) by
tests/test_surface_cpu.py to run epnet_amd's OWN Python surface on the CPU runner, so that the
composition logic (not the kernels) is checked against the fixtures without a GPU. Never imported by
the product.
"""
import sys
import types

import numpy as np
import torch

from oracle import oracle


def _np(t):
    return t.detach().cpu().numpy()


def make_modules():
    """returns (pointnet2_cuda, iou3d_cuda, roipool3d_cuda) look-alikes"""
    def wr(t, arr):
        t.copy_(torch.from_numpy(np.ascontiguousarray(arr)).view_as(t))

    p2 = types.ModuleType("pointnet2_cuda")

    def ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
        wr(idx, oracle.ball_query(radius, nsample, _np(xyz), _np(new_xyz)))
        return 1

    def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
        wr(out, oracle.group_points(_np(points), _np(idx)))
        return 1

    def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
        wr(grad_points, _np(grad_points) + oracle.group_points_grad(_np(grad_out), _np(idx), n))
        return 1

    def gather_points_wrapper(b, c, n, npoints, points, idx, out):
        wr(out, oracle.gather_points(_np(points), _np(idx)))
        return 1

    def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
        wr(grad_points, _np(grad_points) + oracle.gather_points_grad(_np(grad_out), _np(idx), n))
        return 1

    def furthest_point_sampling_wrapper(b, n, m, points, temp, idx):
        wr(idx, oracle.furthest_point_sampling(_np(points), m))
        return 1

    def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
        d, i = oracle.three_nn(_np(unknown), _np(known))
        wr(dist2, d)
        wr(idx, i)

    def three_interpolate_wrapper(b, c, m, n, points, idx, weight, out):
        wr(out, oracle.three_interpolate(_np(points), _np(idx), _np(weight)))

    def three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points):
        wr(grad_points, _np(grad_points) + oracle.three_interpolate_grad(_np(grad_out), _np(idx), _np(weight), m))

    def group_concat_wrapper(b, c, n, npoints, nsample, xyz, new_xyz, features, idx, out, use_xyz):
        parts = []
        if use_xyz:
            g = oracle.group_points(np.ascontiguousarray(_np(xyz).transpose(0, 2, 1)), _np(idx))
            parts.append(g - _np(new_xyz).transpose(0, 2, 1)[..., None])
        if c:
            parts.append(oracle.group_points(_np(features), _np(idx)))
        wr(out, np.concatenate(parts, axis=1))
        return 1

    def group_concat_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_features, use_xyz):
        g = _np(grad_out)[:, (3 if use_xyz else 0):]
        wr(grad_features, _np(grad_features) + oracle.group_points_grad(np.ascontiguousarray(g), _np(idx), n))
        return 1

    for f in (group_concat_wrapper, group_concat_grad_wrapper, ball_query_wrapper, group_points_wrapper, group_points_grad_wrapper, gather_points_wrapper,
              gather_points_grad_wrapper, furthest_point_sampling_wrapper, three_nn_wrapper,
              three_interpolate_wrapper, three_interpolate_grad_wrapper):
        setattr(p2, f.__name__, f)

    def pool_max_wrapper(rows, nsample, x, out, arg):
        v, a = oracle.pool_max(_np(x).reshape(rows, nsample))
        wr(out, v)
        if arg is not None:
            wr(arg, a)
        return 1

    def pool_max_grad_wrapper(rows, nsample, grad_out, arg, grad_x):
        wr(grad_x, oracle.pool_max_grad(_np(grad_out).reshape(rows), _np(arg).reshape(rows), nsample))
        return 1

    def group_linear_wrapper(b, c, n, npoints, nsample, xyz, new_xyz, z, idx, w_xyz, bias, out):
        wr(out, oracle.group_linear(_np(xyz), _np(new_xyz), _np(z), _np(idx), _np(w_xyz), None if bias is None else _np(bias)))
        return 1

    def group_linear_grad_w_wrapper(b, c, n, npoints, nsample, grad_out, xyz, new_xyz, idx, grad_w):
        d = _np(xyz)[np.arange(b)[:, None, None], _np(idx).astype(np.int64)] - _np(new_xyz)[:, :, None, :]      # (b,m,ns,3)
        wr(grad_w, _np(grad_w) + np.einsum("bcms,bmsk->ck", _np(grad_out).astype(np.float64), d.astype(np.float64)).astype(np.float32))
        return 1

    def feature_gather_wrapper(b, c, h, w, n_src, n, align_corners, feature_map, xy, idx, out, xy_out):
        # the op the reference calls (lib/net/pointnet2_msg.py:107-120) is stock torch: it is its own CPU yardstick
        import torch.nn.functional as F
        sel = xy if idx is None else torch.gather(xy, 1, idx.long().unsqueeze(-1).repeat(1, 1, 2))
        out.copy_(F.grid_sample(feature_map, sel.unsqueeze(1), mode="bilinear", padding_mode="zeros", align_corners=bool(align_corners)).squeeze(2))
        if xy_out is not None:
            xy_out.copy_(sel)
        return 1

    def feature_gather_grad_wrapper(b, c, h, w, n, align_corners, grad_out, xy, grad_feature_map):
        import torch.nn.functional as F
        with torch.enable_grad():   # called from inside an autograd backward, where recording is off
            fm = torch.zeros((b, c, h, w), requires_grad=True)
            F.grid_sample(fm, xy.unsqueeze(1), mode="bilinear", padding_mode="zeros", align_corners=bool(align_corners)).squeeze(2).backward(grad_out)
        grad_feature_map.add_(fm.grad)
        return 1

    for f in (pool_max_wrapper, pool_max_grad_wrapper, group_linear_wrapper, group_linear_grad_w_wrapper, feature_gather_wrapper,
              feature_gather_grad_wrapper):
        setattr(p2, f.__name__, f)

    iou = types.ModuleType("iou3d_cuda")

    def boxes_overlap_bev_gpu(a, b, ans):
        wr(ans, oracle.boxes_overlap_bev(_np(a), _np(b)))
        return 1

    def boxes_iou_bev_gpu(a, b, ans):
        wr(ans, oracle.boxes_iou_bev(_np(a), _np(b)))
        return 1

    def nms_gpu(boxes, keep, thresh):
        k = oracle.nms(_np(boxes), thresh, True)
        keep[:len(k)] = torch.from_numpy(k)
        return len(k)

    def nms_normal_gpu(boxes, keep, thresh):
        k = oracle.nms(_np(boxes), thresh, False)
        keep[:len(k)] = torch.from_numpy(k)
        return len(k)

    _iou3d_np = oracle.boxes_iou3d  # reference composition (iou3d_utils.py:21-53) on top of the oracle's BEV overlap

    def boxes_iou3d_fused_gpu(a, b, ans):
        wr(ans, _iou3d_np(_np(a), _np(b)))
        return 1

    def boxes_iou3d_pairs_gpu(a, b, ans):
        wr(ans, np.diagonal(_iou3d_np(_np(a), _np(b))).copy())
        return 1

    def aug_roi_by_noise_gpu(rois, gts, iou_src, keep_draw, noise, pos_thresh, iou_out, tries=None):
        r, i = oracle.aug_roi_by_noise(_np(rois), _np(gts), _np(iou_src), None if keep_draw is None else _np(keep_draw),
                                       None if noise is None else _np(noise), pos_thresh, None if tries is None else _np(tries))
        wr(rois, r)
        wr(iou_out, i)
        return 1

    def rpn_proposals_gpu(proposals, scores, order, distance_based, pre_nms_top_n, post_nms_top_n, nms_thresh, rotated,
                          ret_bbox3d, ret_scores, ret_count=None):
        rb, rs, cnt = oracle.rpn_proposals(_np(proposals), _np(scores), _np(order), distance_based, pre_nms_top_n, post_nms_top_n,
                                           nms_thresh, rotated)
        wr(ret_bbox3d, rb)
        wr(ret_scores, rs)
        if ret_count is not None:
            wr(ret_count, cnt)
        return 1

    def nms_device(boxes, thresh):  # epnet_amd.iou3d_cuda's all-device form
        k = oracle.nms(_np(boxes), thresh, True)
        return torch.from_numpy(k), torch.tensor([len(k)], dtype=torch.int32)

    def nms_normal_device(boxes, thresh):
        k = oracle.nms(_np(boxes), thresh, False)
        return torch.from_numpy(k), torch.tensor([len(k)], dtype=torch.int32)

    for f in (boxes_overlap_bev_gpu, boxes_iou_bev_gpu, nms_gpu, nms_normal_gpu, nms_device, nms_normal_device,
              boxes_iou3d_fused_gpu, boxes_iou3d_pairs_gpu, aug_roi_by_noise_gpu, rpn_proposals_gpu):
        setattr(iou, f.__name__, f)

    rp = types.ModuleType("roipool3d_cuda")

    def forward(xyz, boxes3d, pts_feature, pooled_features, pooled_empty_flag):
        pooled, flag = oracle.roipool3d(_np(xyz), _np(boxes3d), _np(pts_feature), pooled_features.size(2))
        wr(pooled_features, pooled)
        wr(pooled_empty_flag, flag)
        return 1

    rp.forward = forward
    return p2, iou, rp


def install_as_top_level():
    p2, iou, rp = make_modules()
    sys.modules["pointnet2_cuda"] = p2
    sys.modules["iou3d_cuda"] = iou
    sys.modules["roipool3d_cuda"] = rp


