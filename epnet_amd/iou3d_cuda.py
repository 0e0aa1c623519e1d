"""Stand-in for the reference's ``iou3d_cuda`` extension module (lib/utils/iou3d/src/iou3d.cpp:174-179).

``nms_gpu`` / ``nms_normal_gpu`` keep the reference contract -- boxes (N,5) on the GPU sorted by
descending score, ``keep`` a CPU int64 tensor that receives the kept positions, return value = their
count (iou3d.cpp:73-120) -- but mask AND greedy sweep run on the device; only the kept positions
(N*8 B instead of the reference's N*ceil(N/64)*8 B mask) cross PCIe. ``nms_device`` /
``nms_normal_device`` expose the all-device form for callers that do not need a host list.
"""
import torch

from . import _lib
from ._tensor import dev_ptr, host_ptr, need, on_device_of, writes

_F = torch.float32


@writes("ans")
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


@writes("ans_iou3d")
def boxes_iou3d_fused_gpu(boxes_a, boxes_b, ans_iou3d):
    """(N,7) x (M,7) -> (N,M) 3-D IoU in one launch (not in the reference extension; see epnet_ops.h)"""
    pa, pb, po = dev_ptr(boxes_a, "boxes_a", _F), dev_ptr(boxes_b, "boxes_b", _F), dev_ptr(ans_iou3d, "ans", _F)
    na, nb = boxes_a.size(0), boxes_b.size(0)
    need(boxes_a, na * 7, "boxes_a"); need(boxes_b, nb * 7, "boxes_b"); need(ans_iou3d, na * nb, "ans")
    with on_device_of(boxes_a) as s:
        _lib.check(_lib.lib().epnet_boxes_iou3d(na, pa, nb, pb, po, s), "boxes_iou3d")
    return 1


@writes("ans_iou3d")
def boxes_iou3d_pairs_gpu(boxes_a, boxes_b, ans_iou3d):
    """(K,7), (K,7) -> (K,) 3-D IoU of corresponding pairs in one launch"""
    pa, pb, po = dev_ptr(boxes_a, "boxes_a", _F), dev_ptr(boxes_b, "boxes_b", _F), dev_ptr(ans_iou3d, "ans", _F)
    k = boxes_a.size(0)
    need(boxes_a, k * 7, "boxes_a"); need(boxes_b, k * 7, "boxes_b"); need(ans_iou3d, k, "ans")
    with on_device_of(boxes_a) as s:
        _lib.check(_lib.lib().epnet_boxes_iou3d_pairs(k, pa, pb, po, s), "boxes_iou3d_pairs")
    return 1


@writes("roi_boxes3d", "iou_of_rois")
def aug_roi_by_noise_gpu(roi_boxes3d, gt_boxes3d, iou3d_src, keep_draw, noise, pos_thresh, iou_of_rois, tries=None):
    """the ROI augmentation loop of lib/rpn/proposal_target_layer.py:220-247 for all K ROIs in one launch; roi_boxes3d
    (K,7) is updated in place; keep_draw (K,T) uint8, noise (K,T,7), tries (K) int32 per-ROI try limits or None (not in
    the reference extension; see epnet_ops.h)"""
    k = roi_boxes3d.size(0)
    t = keep_draw.size(1) if keep_draw is not None else 0
    pr, pg = dev_ptr(roi_boxes3d, "roi_boxes3d", _F), dev_ptr(gt_boxes3d, "gt_boxes3d", _F)
    ps, po = dev_ptr(iou3d_src, "iou3d_src", _F), dev_ptr(iou_of_rois, "iou_of_rois", _F)
    need(roi_boxes3d, k * 7, "roi_boxes3d"); need(gt_boxes3d, k * 7, "gt_boxes3d"); need(iou3d_src, k, "iou3d_src")
    need(iou_of_rois, k, "iou_of_rois")
    pk = pn = pt = None
    if tries is not None:
        pt = dev_ptr(tries, "tries", torch.int32)
        need(tries, k, "tries")
    if t:
        pk, pn = dev_ptr(keep_draw, "keep_draw", torch.uint8), dev_ptr(noise, "noise", _F)
        need(keep_draw, k * t, "keep_draw"); need(noise, k * t * 7, "noise")
    with on_device_of(roi_boxes3d) as s:
        _lib.check(_lib.lib().epnet_aug_roi_by_noise(k, t, float(pos_thresh), pr, pg, ps, pt, pk, pn, po, s), "aug_roi_by_noise")
    return 1


@writes("ret_bbox3d", "ret_scores", "ret_count")
def rpn_proposals_gpu(proposals, scores, order, distance_based, pre_nms_top_n, post_nms_top_n, nms_thresh, rotated,
                      ret_bbox3d, ret_scores, ret_count=None):
    """lib/rpn/proposal_layer.py:34-55 for the whole batch, no host sync: proposals (B,N,7), scores (B,N), order (B,N)
    int64 (descending scores) -> ret_bbox3d (B,post,7), ret_scores (B,post) (not in the reference extension; see
    epnet_ops.h)"""
    b, n = scores.size(0), scores.size(1)
    pp, ps, po = dev_ptr(proposals, "proposals", _F), dev_ptr(scores, "scores", _F), dev_ptr(order, "order", torch.int64)
    pb, pr = dev_ptr(ret_bbox3d, "ret_bbox3d", _F), dev_ptr(ret_scores, "ret_scores", _F)
    need(proposals, b * n * 7, "proposals"); need(order, b * n, "order")
    need(ret_bbox3d, b * post_nms_top_n * 7, "ret_bbox3d"); need(ret_scores, b * post_nms_top_n, "ret_scores")
    pc = None
    if ret_count is not None:
        pc = dev_ptr(ret_count, "ret_count", torch.int32)
        need(ret_count, b, "ret_count")
    l = _lib.lib()
    ws_bytes = l.epnet_rpn_proposals_workspace_bytes(b, int(bool(distance_based)), pre_nms_top_n, post_nms_top_n)
    ws = torch.empty((max(ws_bytes, 16),), dtype=torch.uint8, device=scores.device)
    with on_device_of(scores) as s:
        _lib.check(l.epnet_rpn_proposals(b, n, pp, ps, po, int(bool(distance_based)), pre_nms_top_n, post_nms_top_n,
                                         float(nms_thresh), int(bool(rotated)), ws.data_ptr(), ws.numel(), pb, pr, pc, s),
                   "rpn_proposals")
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
