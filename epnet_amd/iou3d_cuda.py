"""Stand-in for the reference's ``iou3d_cuda`` extension module (lib/utils/iou3d/src/iou3d.cpp:174-179).

``nms_gpu`` / ``nms_normal_gpu`` keep the reference contract -- boxes (N,5) on the GPU sorted by
descending score, ``keep`` a CPU int64 tensor that receives the kept positions, return value = their
count (iou3d.cpp:73-120) -- but mask AND greedy sweep run on the device; only the kept positions
(N*8 B instead of the reference's N*ceil(N/64)*8 B mask) cross PCIe. ``nms_device`` /
``nms_normal_device`` expose the all-device form for callers that do not need a host list.
"""
import torch

from . import _lib
from ._tensor import dev_ptr, host_ptr, need, on_device_of

_F = torch.float32


def _pairwise(fn_name, boxes_a, boxes_b, ans):
    pa, pb, po = dev_ptr(boxes_a, "boxes_a", _F), dev_ptr(boxes_b, "boxes_b", _F), dev_ptr(ans, "ans", _F)
    na, nb = boxes_a.size(0), boxes_b.size(0)
    need(boxes_a, na * 5, "boxes_a"); need(boxes_b, nb * 5, "boxes_b"); need(ans, na * nb, "ans")
    with on_device_of(boxes_a) as s:
        _lib.check(getattr(_lib.lib(), fn_name)(na, pa, nb, pb, po, s), fn_name)
    return 1


def boxes_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
    """iou3d.cpp:31-50"""
    return _pairwise("epnet_boxes_overlap_bev", boxes_a, boxes_b, ans_overlap)


def boxes_iou_bev_gpu(boxes_a, boxes_b, ans_iou):
    """iou3d.cpp:52-71"""
    return _pairwise("epnet_boxes_iou_bev", boxes_a, boxes_b, ans_iou)


def boxes_iou3d_fused_gpu(boxes_a, boxes_b, ans_iou3d):
    """(N,7) x (M,7) -> (N,M) 3-D IoU in one launch (not in the reference extension; see epnet_ops.h)"""
    pa, pb, po = dev_ptr(boxes_a, "boxes_a", _F), dev_ptr(boxes_b, "boxes_b", _F), dev_ptr(ans_iou3d, "ans", _F)
    na, nb = boxes_a.size(0), boxes_b.size(0)
    need(boxes_a, na * 7, "boxes_a"); need(boxes_b, nb * 7, "boxes_b"); need(ans_iou3d, na * nb, "ans")
    with on_device_of(boxes_a) as s:
        _lib.check(_lib.lib().epnet_boxes_iou3d(na, pa, nb, pb, po, s), "boxes_iou3d")
    return 1


def boxes_iou3d_pairs_gpu(boxes_a, boxes_b, ans_iou3d):
    """(K,7), (K,7) -> (K,) 3-D IoU of corresponding pairs in one launch"""
    pa, pb, po = dev_ptr(boxes_a, "boxes_a", _F), dev_ptr(boxes_b, "boxes_b", _F), dev_ptr(ans_iou3d, "ans", _F)
    k = boxes_a.size(0)
    need(boxes_a, k * 7, "boxes_a"); need(boxes_b, k * 7, "boxes_b"); need(ans_iou3d, k, "ans")
    with on_device_of(boxes_a) as s:
        _lib.check(_lib.lib().epnet_boxes_iou3d_pairs(k, pa, pb, po, s), "boxes_iou3d_pairs")
    return 1


def _nms_device(fn_name, boxes, thresh):
    """returns (keep_dev int64 (N,), num_keep_dev int32 (1,)), both on the boxes' device, no sync"""
    pb = dev_ptr(boxes, "boxes", _F)
    n = boxes.size(0)
    need(boxes, n * 5, "boxes")
    l = _lib.lib()
    ws_bytes = l.epnet_nms_workspace_bytes(n)
    ws = torch.empty((max(ws_bytes, 8),), dtype=torch.uint8, device=boxes.device)
    keep = torch.empty((max(n, 1),), dtype=torch.int64, device=boxes.device)
    num = torch.empty((1,), dtype=torch.int32, device=boxes.device)
    with on_device_of(boxes) as s:
        _lib.check(getattr(l, fn_name)(pb, n, float(thresh), ws.data_ptr(), ws.numel(), keep.data_ptr(),
                                       num.data_ptr(), s), fn_name)
    return keep, num


def nms_device(boxes, thresh):
    return _nms_device("epnet_nms", boxes, thresh)


def nms_normal_device(boxes, thresh):
    return _nms_device("epnet_nms_normal", boxes, thresh)


def _nms_host_contract(fn_name, boxes, keep, thresh):
    host_ptr(keep, "keep", torch.int64)
    need(keep, boxes.size(0), "keep")
    keep_dev, num = _nms_device(fn_name, boxes, thresh)
    n = int(num.item())  # the one unavoidable sync: the output length is data dependent
    keep[:n].copy_(keep_dev[:n])
    return n


def nms_gpu(boxes, keep, nms_overlap_thresh):
    """iou3d.cpp:73-120"""
    return _nms_host_contract("epnet_nms", boxes, keep, nms_overlap_thresh)


def nms_normal_gpu(boxes, keep, nms_overlap_thresh):
    """iou3d.cpp:123-170"""
    return _nms_host_contract("epnet_nms_normal", boxes, keep, nms_overlap_thresh)
