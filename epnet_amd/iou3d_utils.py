"""Rotated-box IoU / NMS operator surface (reference: lib/utils/iou3d/iou3d_utils.py:6-87).

Same four public functions and return conventions. The NMS variants stay on the device end to end:
mask and greedy sweep are kernels, and the kept-index gather reads the device-side keep list; the
only host round trip is the 4-byte count that sizes the (data dependent) result.
"""
import torch

from . import iou3d_cuda
from . import kitti_utils


def boxes_iou_bev(boxes_a, boxes_b):
    """boxes_a (M,5), boxes_b (N,5) [x1,y1,x2,y2,ry] -> rotated BEV IoU (M,N)"""
    ans_iou = torch.zeros((boxes_a.shape[0], boxes_b.shape[0]), dtype=torch.float32, device=boxes_a.device)
    iou3d_cuda.boxes_iou_bev_gpu(boxes_a.contiguous(), boxes_b.contiguous(), ans_iou)
    return ans_iou


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """boxes_a (N,7), boxes_b (M,7) [x,y,z,h,w,l,ry] (y = bottom centre, camera coords) -> 3-D IoU (N,M):
    rotated BEV overlap x height overlap over the union volume (clamped at 1e-7). One fused launch; the
    reference composes boxes3d_to_bev_torch, the overlap kernel and ~10 elementwise torch ops (:21-53)."""
    ans = torch.empty((boxes_a.shape[0], boxes_b.shape[0]), dtype=torch.float32, device=boxes_a.device)
    iou3d_cuda.boxes_iou3d_fused_gpu(boxes_a.float().contiguous(), boxes_b.float().contiguous(), ans)
    return ans


def boxes_iou3d_pairs_gpu(boxes_a, boxes_b):
    """boxes_a (K,7), boxes_b (K,7) -> (K,) 3-D IoU of pair i = (a_i, b_i) in one launch -- the batched form of
    the single-pair calls in lib/rpn/proposal_target_layer.py:239 (not part of the reference's surface)"""
    ans = torch.empty((boxes_a.shape[0],), dtype=torch.float32, device=boxes_a.device)
    iou3d_cuda.boxes_iou3d_pairs_gpu(boxes_a.float().contiguous(), boxes_b.float().contiguous(), ans)
    return ans


def boxes_iou3d_composed(boxes_a, boxes_b):
    """the reference's composition (BEV overlap kernel + torch height/volume math), kept for cross-checks"""
    bev_a = kitti_utils.boxes3d_to_bev_torch(boxes_a)
    bev_b = kitti_utils.boxes3d_to_bev_torch(boxes_b)
    overlaps_bev = torch.zeros((boxes_a.shape[0], boxes_b.shape[0]), dtype=torch.float32, device=boxes_a.device)
    iou3d_cuda.boxes_overlap_bev_gpu(bev_a.contiguous(), bev_b.contiguous(), overlaps_bev)
    # y points down: a box spans [y - h, y]
    a_top, a_bottom = (boxes_a[:, 1] - boxes_a[:, 3]).view(-1, 1), boxes_a[:, 1].view(-1, 1)
    b_top, b_bottom = (boxes_b[:, 1] - boxes_b[:, 3]).view(1, -1), boxes_b[:, 1].view(1, -1)
    overlaps_h = torch.clamp(torch.min(a_bottom, b_bottom) - torch.max(a_top, b_top), min=0)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(1, -1)
    return overlaps_3d / torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-7)


def _nms(device_fn, boxes, scores, thresh):
    order = scores.sort(0, descending=True)[1]
    keep, num = device_fn(boxes[order].contiguous(), thresh)
    return order[keep[:int(num.item())]].contiguous()


def nms_gpu(boxes, scores, thresh):
    """rotated NMS: boxes (N,5), scores (N) -> int64 indices (into the input) of the kept boxes,
    highest score first"""
    return _nms(iou3d_cuda.nms_device, boxes, scores, thresh)


def nms_normal_gpu(boxes, scores, thresh):
    """axis-aligned NMS on the [x1,y1,x2,y2] part (ry ignored)"""
    return _nms(iou3d_cuda.nms_normal_device, boxes, scores, thresh)
