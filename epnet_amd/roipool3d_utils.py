"""ROI point pooling operator surface (reference: lib/utils/roipool3d/roipool3d_utils.py:7-108)."""
import numpy as np
import torch

from . import kitti_utils
from . import roipool3d_cuda


def roipool3d_gpu(pts, pts_feature, boxes3d, pool_extra_width, sampled_pt_num=512):
    """pts (B,N,3), pts_feature (B,N,C), boxes3d (B,M,7) -> (pooled_features (B,M,S,3+C),
    pooled_empty_flag (B,M) int32). Boxes are enlarged by pool_extra_width first; each box takes its
    first S in-box points in index order, repeated cyclically when fewer; empty boxes stay zero."""
    batch_size, boxes_num, feature_len = pts.shape[0], boxes3d.shape[1], pts_feature.shape[2]
    pooled_boxes3d = kitti_utils.enlarge_box3d(boxes3d.view(-1, 7), pool_extra_width).view(batch_size, -1, 7)
    pooled_features = torch.zeros((batch_size, boxes_num, sampled_pt_num, 3 + feature_len), dtype=torch.float32,
                                  device=pts.device)
    pooled_empty_flag = torch.zeros((batch_size, boxes_num), dtype=torch.int32, device=pts.device)
    roipool3d_cuda.forward(pts.contiguous(), pooled_boxes3d.contiguous(), pts_feature.contiguous(), pooled_features,
                           pooled_empty_flag)
    return pooled_features, pooled_empty_flag


def pts_in_boxes3d_cpu(pts, boxes3d):
    """pts (N,3), boxes3d (M,7), both CPU -> list of M boolean masks (N,)"""
    if pts.is_cuda:
        raise NotImplementedError
    pts = pts.float().contiguous()
    boxes3d = boxes3d.float().contiguous()
    pts_flag = torch.zeros((boxes3d.size(0), pts.size(0)), dtype=torch.int64)
    roipool3d_cuda.pts_in_boxes3d_cpu(pts_flag, pts, boxes3d)
    return [pts_flag[k] > 0 for k in range(boxes3d.shape[0])]


def roipool_pc_cpu(pts, pts_feature, boxes3d, sampled_pt_num):
    """CPU pooling: -> pooled_pts (M,S,3), pooled_features (M,S,C), pooled_empty_flag (M) int64"""
    pts = pts.cpu().float().contiguous()
    pts_feature = pts_feature.cpu().float().contiguous()
    boxes3d = boxes3d.cpu().float().contiguous()
    assert pts.shape[0] == pts_feature.shape[0] and pts.shape[1] == 3, '%s %s' % (pts.shape, pts_feature.shape)
    num_boxes = boxes3d.shape[0]
    pooled_pts = torch.zeros((num_boxes, sampled_pt_num, 3), dtype=torch.float32)
    pooled_features = torch.zeros((num_boxes, sampled_pt_num, pts_feature.shape[1]), dtype=torch.float32)
    pooled_empty_flag = torch.zeros((num_boxes,), dtype=torch.int64)
    roipool3d_cuda.roipool3d_cpu(pts, boxes3d, pts_feature, pooled_pts, pooled_features, pooled_empty_flag)
    return pooled_pts, pooled_features, pooled_empty_flag


def _rotate_xz(points, angle):
    """rotate_pc_along_y of the reference (lib/utils/kitti_utils.py:29-42) for one (S, 3+C) array"""
    c, s = np.cos(angle), np.sin(angle)
    points[:, [0, 2]] = points[:, [0, 2]] @ np.array([[c, s], [-s, c]])
    return points


def roipool3d_cpu(boxes3d, pts, pts_feature, pts_extra_input, pool_extra_width, sampled_pt_num=512,
                  canonical_transform=True):
    """numpy front end used by the dataset code: pools [extra_input | feature] per enlarged box and,
    with canonical_transform, moves every ROI to its own frame (centre at origin, heading along x)."""
    pooled_boxes3d = kitti_utils.enlarge_box3d(boxes3d, pool_extra_width)
    all_feature = np.concatenate((pts_extra_input, pts_feature), axis=1)
    pooled_pts, pooled_features, pooled_empty_flag = roipool_pc_cpu(
        torch.from_numpy(pts), torch.from_numpy(all_feature), torch.from_numpy(pooled_boxes3d), sampled_pt_num)

    n_extra = pts_extra_input.shape[1]
    sampled_pts_input = torch.cat((pooled_pts, pooled_features[:, :, 0:n_extra]), dim=2).numpy()
    sampled_pts_feature = pooled_features[:, :, n_extra:].numpy()
    if not canonical_transform:
        return sampled_pts_input, sampled_pts_feature, pooled_empty_flag.numpy()

    roi_ry = boxes3d[:, 6] % (2 * np.pi)
    sampled_pts_input[:, :, 0:3] = sampled_pts_input[:, :, 0:3] - boxes3d[:, np.newaxis, 0:3]
    for k in range(sampled_pts_input.shape[0]):
        sampled_pts_input[k] = _rotate_xz(sampled_pts_input[k], roi_ry[k])
    return sampled_pts_input, sampled_pts_feature
