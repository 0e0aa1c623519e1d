"""The three box helpers the operator surface depends on (reference: lib/utils/kitti_utils.py:45-63,
137-150, 153-163). The rest of that file (label / hull utilities, shapely IoU) is out of scope."""
import numpy as np
import torch


def rotate_pc_along_y_torch(pc, rot_angle):
    """pc (N,S,3+C), rot_angle (N): rotates the (x,z) columns in place by the per-row angle"""
    c, s = torch.cos(rot_angle).view(-1, 1, 1), torch.sin(rot_angle).view(-1, 1, 1)
    rot_t = torch.cat([torch.cat([c, s], dim=2), torch.cat([-s, c], dim=2)], dim=1)  # R^T, (N,2,2)
    pc[:, :, [0, 2]] = torch.matmul(pc[:, :, [0, 2]], rot_t)
    return pc


def boxes3d_to_bev_torch(boxes3d):
    """(N,7) [x,y,z,h,w,l,ry] -> (N,5) [x1,y1,x2,y2,ry] with x along l, y along w"""
    half_l, half_w = boxes3d[:, 5] / 2, boxes3d[:, 4] / 2
    cu, cv = boxes3d[:, 0], boxes3d[:, 2]
    return torch.stack([cu - half_l, cv - half_w, cu + half_l, cv + half_w, boxes3d[:, 6]], dim=1)


def enlarge_box3d(boxes3d, extra_width):
    """(N,7): h,w,l += 2*extra_width; y (bottom centre) += extra_width. numpy or torch."""
    large = boxes3d.copy() if isinstance(boxes3d, np.ndarray) else boxes3d.clone()
    large[:, 3:6] += extra_width * 2
    large[:, 1] += extra_width
    return large
