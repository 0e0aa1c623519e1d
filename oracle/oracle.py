"""TEST INFRASTRUCTURE -- numpy front-end of the CPU oracle (oracle/epnet_oracle.c).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module. It restates the reference's device kernels on the CPU (citations in epnet_oracle.c); it
is the checker, never the thing shipped or measured as the product.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libepnet_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "epnet_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libepnet_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_box_overlap.restype = ctypes.c_float
        _lib.oracle_iou_bev.restype = ctypes.c_float
        _lib.oracle_iou_normal.restype = ctypes.c_float
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


_F = ctypes.c_float
_L = ctypes.c_long


def opt_n_threads(n):
    return int(lib().oracle_opt_n_threads(int(n)))


def furthest_point_sampling(xyz, npoint, return_temp=False, temp=None):
    """temp: optional (b, n) running distances to start from (the extension's in/out buffer, sampling.cpp:36-46);
    default 1e10 everywhere, as pointnet2_utils.py:26 fills it"""
    xyz = _f32(xyz)
    b, n, _ = xyz.shape
    temp = np.full((b, n), 1e10, dtype=np.float32) if temp is None else _f32(temp).copy()
    idx = np.zeros((b, npoint), dtype=np.int32)
    lib().oracle_furthest_point_sampling(b, n, npoint, _p(xyz), _p(temp), _p(idx))
    return (idx, temp) if return_temp else idx


def gather_points(points, idx):
    points, idx = _f32(points), _i32(idx)
    b, c, n = points.shape
    m = idx.shape[1]
    out = np.empty((b, c, m), dtype=np.float32)
    lib().oracle_gather_points(b, c, n, m, _p(points), _p(idx), _p(out))
    return out


def gather_points_grad(grad_out, idx, n):
    grad_out, idx = _f32(grad_out), _i32(idx)
    b, c, m = grad_out.shape
    g = np.zeros((b, c, n), dtype=np.float32)
    lib().oracle_gather_points_grad(b, c, n, m, _p(grad_out), _p(idx), _p(g))
    return g


def ball_query(radius, nsample, xyz, new_xyz):
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = np.zeros((b, m, nsample), dtype=np.int32)
    lib().oracle_ball_query(b, n, m, _F(radius), nsample, _p(new_xyz), _p(xyz), _p(idx))
    return idx


def group_points(points, idx):
    points, idx = _f32(points), _i32(idx)
    b, c, n = points.shape
    _, m, ns = idx.shape
    out = np.empty((b, c, m, ns), dtype=np.float32)
    lib().oracle_group_points(b, c, n, m, ns, _p(points), _p(idx), _p(out))
    return out


def group_points_grad(grad_out, idx, n):
    grad_out, idx = _f32(grad_out), _i32(idx)
    b, c, m, ns = grad_out.shape
    g = np.zeros((b, c, n), dtype=np.float32)
    lib().oracle_group_points_grad(b, c, n, m, ns, _p(grad_out), _p(idx), _p(g))
    return g


def three_nn(unknown, known):
    """returns (dist2, idx): SQUARED distances, as the extension does (the Python surface sqrt's them)."""
    unknown, known = _f32(unknown), _f32(known)
    b, n, _ = unknown.shape
    m = known.shape[1]
    dist2 = np.empty((b, n, 3), dtype=np.float32)
    idx = np.empty((b, n, 3), dtype=np.int32)
    lib().oracle_three_nn(b, n, m, _p(unknown), _p(known), _p(dist2), _p(idx))
    return dist2, idx


def three_interpolate(points, idx, weight):
    points, idx, weight = _f32(points), _i32(idx), _f32(weight)
    b, c, m = points.shape
    n = idx.shape[1]
    out = np.empty((b, c, n), dtype=np.float32)
    lib().oracle_three_interpolate(b, c, m, n, _p(points), _p(idx), _p(weight), _p(out))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    grad_out, idx, weight = _f32(grad_out), _i32(idx), _f32(weight)
    b, c, n = grad_out.shape
    g = np.zeros((b, c, m), dtype=np.float32)
    lib().oracle_three_interpolate_grad(b, c, n, m, _p(grad_out), _p(idx), _p(weight), _p(g))
    return g


def boxes_overlap_bev(boxes_a, boxes_b):
    boxes_a, boxes_b = _f32(boxes_a), _f32(boxes_b)
    out = np.empty((boxes_a.shape[0], boxes_b.shape[0]), dtype=np.float32)
    lib().oracle_boxes_overlap_bev(boxes_a.shape[0], _p(boxes_a), boxes_b.shape[0], _p(boxes_b), _p(out))
    return out


def boxes_iou_bev(boxes_a, boxes_b):
    boxes_a, boxes_b = _f32(boxes_a), _f32(boxes_b)
    out = np.empty((boxes_a.shape[0], boxes_b.shape[0]), dtype=np.float32)
    lib().oracle_boxes_iou_bev(boxes_a.shape[0], _p(boxes_a), boxes_b.shape[0], _p(boxes_b), _p(out))
    return out


def nms_mask(boxes, thresh, rotated):
    boxes = _f32(boxes)
    n = boxes.shape[0]
    mask = np.zeros((n, (n + 63) // 64), dtype=np.uint64)
    lib().oracle_nms_mask(n, _F(thresh), _p(boxes), _p(mask), int(bool(rotated)))
    return mask


def nms(boxes, thresh, rotated):
    """boxes already sorted by descending score; returns kept positions (int64), like the ext."""
    boxes = _f32(boxes)
    n = boxes.shape[0]
    keep = np.zeros((max(n, 1),), dtype=np.int64)
    k = lib().oracle_nms(n, _F(thresh), _p(boxes), _p(keep), int(bool(rotated)))
    return keep[:k].copy()


def roipool3d(xyz, boxes3d, pts_feature, sampled_pts_num):
    xyz, boxes3d, pts_feature = _f32(xyz), _f32(boxes3d), _f32(pts_feature)
    b, n, _ = xyz.shape
    m = boxes3d.shape[1]
    c = pts_feature.shape[2]
    pooled = np.zeros((b, m, sampled_pts_num, 3 + c), dtype=np.float32)
    flag = np.zeros((b, m), dtype=np.int32)
    lib().oracle_roipool3d(b, n, m, c, sampled_pts_num, _p(xyz), _p(boxes3d), _p(pts_feature), _p(pooled), _p(flag))
    return pooled, flag


def pts_in_boxes3d(pts, boxes3d):
    pts, boxes3d = _f32(pts), _f32(boxes3d)
    m, n = boxes3d.shape[0], pts.shape[0]
    flag = np.empty((m, n), dtype=np.int64)
    lib().oracle_pts_in_boxes3d(_p(flag), _p(pts), _p(boxes3d), _L(m), _L(n))
    return flag


def roipool3d_cpu(pts, boxes3d, pts_feature, sampled_pts_num):
    pts, boxes3d, pts_feature = _f32(pts), _f32(boxes3d), _f32(pts_feature)
    m, n, c = boxes3d.shape[0], pts.shape[0], pts_feature.shape[1]
    pooled_pts = np.zeros((m, sampled_pts_num, 3), dtype=np.float32)
    pooled_feat = np.zeros((m, sampled_pts_num, c), dtype=np.float32)
    flag = np.zeros((m,), dtype=np.int64)
    lib().oracle_roipool3d_cpu(_p(pts), _p(boxes3d), _p(pts_feature), _p(pooled_pts), _p(pooled_feat), _p(flag),
                               _L(m), _L(n), _L(c), _L(sampled_pts_num))
    return pooled_pts, pooled_feat, flag


def boxes_iou3d(boxes_a, boxes_b):
    """boxes_iou3d_gpu, lib/utils/iou3d/iou3d_utils.py:21-53: (N,7),(M,7) [x,y,z,h,w,l,ry] -> (N,M). The reference's fp32
    torch composition (boxes3d_to_bev_torch kitti_utils.py:137-150, overlap kernel, height overlap, volumes, clamp,
    divide) restated with numpy float32 operations in the same order."""
    a, b = _f32(boxes_a), _f32(boxes_b)

    def bev(x):
        return np.stack([x[:, 0] - x[:, 5] / 2, x[:, 2] - x[:, 4] / 2, x[:, 0] + x[:, 5] / 2, x[:, 2] + x[:, 4] / 2, x[:, 6]],
                        1).astype(np.float32)
    ov = boxes_overlap_bev(bev(a), bev(b))
    a_top, a_bot = (a[:, 1] - a[:, 3])[:, None], a[:, 1][:, None]
    b_top, b_bot = (b[:, 1] - b[:, 3])[None, :], b[:, 1][None, :]
    oh = np.maximum(np.minimum(a_bot, b_bot) - np.maximum(a_top, b_top), np.float32(0))
    o3 = (ov * oh).astype(np.float32)
    va, vb = (a[:, 3] * a[:, 4] * a[:, 5])[:, None], (b[:, 3] * b[:, 4] * b[:, 5])[None, :]
    return (o3 / np.maximum(va + vb - o3, np.float32(1e-7))).astype(np.float32)


def aug_roi_by_noise(roi_boxes3d, gt_boxes3d, iou3d_src, keep_draw, noise, pos_thresh, tries=None):
    """aug_roi_by_noise_torch, lib/rpn/proposal_target_layer.py:220-247, with the random draws of try `cnt` of ROI `k`
    taken from keep_draw[k, cnt] (the coin `np.random.rand() < 0.2`, :232) and noise[k, cnt] = pos_shift[3],
    hwl_scale[3], angle_rot (random_aug_box3d :250-275; the box is [xyz + shift, hwl * scale, ry + rot], :259,274).
    tries[k] bounds the tries of ROI k (aug_times of the call: ROI_FG_AUG_TIMES for foreground, 1 for background).
    Returns (augmented rois (K,7), iou_of_rois (K))."""
    rois, gts, src = _f32(roi_boxes3d).copy(), _f32(gt_boxes3d), _f32(iou3d_src)
    noise = _f32(noise)
    k_total = rois.shape[0]
    aug_times = 0 if keep_draw is None else keep_draw.shape[1]
    out_iou = np.zeros((k_total,), np.float32)
    thresh = np.float32(pos_thresh)
    for k in range(k_total):
        n_try = aug_times if tries is None else min(max(int(tries[k]), 0), aug_times)
        temp_iou, cnt, keep = np.float32(0), 0, True
        roi = rois[k].copy()
        aug = roi
        while temp_iou < thresh and cnt < n_try:                                    # :231
            if keep_draw[k, cnt]:
                aug, keep = roi, True                                               # :233-234
            else:
                nz = noise[k, cnt]
                aug = np.concatenate([roi[0:3] + nz[0:3], roi[3:6] * nz[3:6], roi[6:7] + nz[6:7]]).astype(np.float32)
                keep = False
            temp_iou = boxes_iou3d(aug[None], gts[k][None])[0, 0]                   # :238-240
            cnt += 1
        rois[k] = aug                                                               # :242
        out_iou[k] = src[k] if (cnt == 0 or keep) else temp_iou                     # :243-246
    return rois, out_iou


def rpn_proposals(proposals, scores, order, distance_based, pre_nms_top_n, post_nms_top_n, nms_thresh, rotated):
    """ProposalLayer.forward after the decoding, lib/rpn/proposal_layer.py:34-55 with distance_based_proposal :58-119 and
    score_based_proposal :121-142, scene by scene in numpy on top of the oracle's NMS. proposals (B,N,7), scores (B,N),
    order (B,N) = positions by descending score. Returns (ret_bbox3d (B,post,7), ret_scores (B,post), kept per scene)."""
    proposals, scores = _f32(proposals), _f32(scores)
    b = scores.shape[0]
    ret_b = np.zeros((b, post_nms_top_n, 7), np.float32)
    ret_s = np.zeros((b, post_nms_top_n), np.float32)
    count = np.zeros((b,), np.int32)

    def bev_of(x):  # kitti_utils.boxes3d_to_bev_torch :137-150
        return np.stack([x[:, 0] - x[:, 5] / 2, x[:, 2] - x[:, 4] / 2, x[:, 0] + x[:, 5] / 2, x[:, 2] + x[:, 4] / 2, x[:, 6]],
                        1).astype(np.float32).reshape(-1, 5)

    for k in range(b):
        s_ord, p_ord = scores[k][order[k]], proposals[k][order[k]]
        s_list, p_list = [], []
        if distance_based:
            ranges = [0, 40.0, 80.0]                                                                     # :65
            pre = [0, int(pre_nms_top_n * 0.7), pre_nms_top_n - int(pre_nms_top_n * 0.7)]                # :66-67
            post = [0, int(post_nms_top_n * 0.7), post_nms_top_n - int(post_nms_top_n * 0.7)]            # :68-69
            dist = p_ord[:, 2]
            first_mask = (dist > ranges[0]) & (dist <= ranges[1])
            for i in (1, 2):
                m = (dist > ranges[i - 1]) & (dist <= ranges[i])
                if m.sum() != 0:
                    cur_s, cur_p = s_ord[m][:pre[i]], p_ord[m][:pre[i]]                                  # :82-91
                else:
                    cur_s = s_ord[first_mask][pre[i - 1]:][:pre[i]]                                      # :92-100
                    cur_p = p_ord[first_mask][pre[i - 1]:][:pre[i]]
                keep = nms(bev_of(cur_p), nms_thresh, rotated)[:post[i]]                                  # :102-112
                s_list.append(cur_s[keep])
                p_list.append(cur_p[keep])
        else:
            cur_s, cur_p = s_ord[:pre_nms_top_n], p_ord[:pre_nms_top_n]                                  # :133-134
            keep = nms(bev_of(cur_p), nms_thresh, True)[:post_nms_top_n]                                  # :136-140
            s_list.append(cur_s[keep])
            p_list.append(cur_p[keep])
        s_all, p_all = np.concatenate(s_list), np.concatenate(p_list).reshape(-1, 7)
        ret_b[k, :len(s_all)] = p_all                                                                     # :52-54
        ret_s[k, :len(s_all)] = s_all
        count[k] = len(s_all)
    return ret_b, ret_s, count


def pool_max(x):
    """F.max_pool2d(x, kernel_size=[1, nsample]) of pointnet2_lib/pointnet2/pointnet2_modules.py:61-68 as numpy: maximum over
    the last axis (keepdim) and its first position (the stock kernel scans in order with a strict `>`)."""
    x = _f32(x)
    return x.max(axis=-1, keepdims=True), x.argmax(axis=-1).astype(np.int32)


def pool_max_grad(grad_out, arg, nsample):
    g = np.zeros(arg.shape + (nsample,), np.float32)
    np.put_along_axis(g, arg[..., None].astype(np.int64), _f32(grad_out).reshape(arg.shape + (1,)), axis=-1)
    return g


def group_linear(xyz, new_xyz, z, idx, w_xyz, bias=None):
    """epnet_group_linear: out[b,co,m,s] = z[b,co,idx[b,m,s]] + (w[co,0]*dx + w[co,1]*dy + w[co,2]*dz) (+ bias[co]) with
    (dx,dy,dz) = xyz[b,idx[b,m,s]] - new_xyz[b,m], fp32 products and sums in this order. It is the first 1x1 convolution
    of an SA level (pointnet2_modules.py:61) applied to [grouped xyz - centre ; grouped features] (pointnet2_utils.py:250-257)
    with the feature part of the product, z = W_f . features, taken before the gather."""
    xyz, new_xyz, z, w = _f32(xyz), _f32(new_xyz), _f32(z), _f32(w_xyz)
    b, c, n = z.shape
    m, ns = idx.shape[1], idx.shape[2]
    out = np.empty((b, c, m, ns), np.float32)
    for bi in range(b):
        d = (xyz[bi][idx[bi]] - new_xyz[bi][:, None, :]).astype(np.float32)          # (m, ns, 3)
        t = (w[:, None, None, 0] * d[None, :, :, 0]).astype(np.float32)
        t = (t + (w[:, None, None, 1] * d[None, :, :, 1]).astype(np.float32)).astype(np.float32)
        t = (t + (w[:, None, None, 2] * d[None, :, :, 2]).astype(np.float32)).astype(np.float32)
        v = (z[bi][:, idx[bi]] + t).astype(np.float32)
        if bias is not None:
            v = (v + _f32(bias)[:, None, None]).astype(np.float32)
        out[bi] = v
    return out

