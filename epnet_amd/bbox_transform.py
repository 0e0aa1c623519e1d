"""Box decoding from the bin-based regression head (reference: lib/utils/bbox_transform.py:25-262, decode_bbox_target)
-- the step in front of the proposal layer's NMS.

Same signature and results as the reference function; the two switches it reads from the global ``cfg``
(``TRAIN/TEST.BBOX_AVG_BY_BIN`` :44-46, ``TRAIN/TEST.RY_WITH_BIN`` :133-134) are keyword arguments here, defaulting to
the values of tools/cfgs/LI_Fusion_with_attention_use_ce_loss.yaml (:180-181, :191-192). Stock tensor expressions (dense
elementwise work over (N, C) rows) written so that nothing synchronises with the host (no boolean-mask indexing, no
host-built index tensors): the proposal layer records into a HIP graph.
"""
import numpy as np
import torch
import torch.nn.functional as F


def rotate_pc_along_y_torch(pc, rot_angle):
    """pc (N, 3+C), rot_angle (N): rotates (x, z) of every row by its own angle, in place (:6-22)"""
    cosa, sina = torch.cos(rot_angle), torch.sin(rot_angle)
    x, z = pc[:, 0].clone(), pc[:, 2].clone()
    # [x z] @ [[c, -s], [s, c]]^T as the reference's 1x2 by 2x2 matmul sums it: x*c + z*(-s), x*s + z*c
    pc[:, 0] = x * cosa + z * (-sina)
    pc[:, 2] = x * sina + z * cosa
    return pc


def _hard_bin(reg, bins_at, res_at, nb, bin_size, scope, with_residual):
    """arg-max bin centre (+ the residual predicted for that bin): bbox_transform.py:48-72, one axis"""
    which = torch.argmax(reg[:, bins_at:bins_at + nb], dim=1)
    pos = which.float() * bin_size + bin_size / 2 - scope
    if with_residual:
        res_norm = torch.gather(reg[:, res_at:res_at + nb], dim=1, index=which.unsqueeze(dim=1)).squeeze(dim=1)
        pos += res_norm * bin_size
    return pos


def _soft_bin(reg, bins_at, res_at, nb, centres, bin_size):
    """soft-max weighted mean of (bin centre + that bin's residual): bbox_transform.py:73-106, one axis"""
    prob = F.softmax(reg[:, bins_at:bins_at + nb], 1)
    absolute = centres + reg[:, res_at:res_at + nb] * bin_size
    return (absolute * prob).sum(dim=1)


def decode_bbox_target(roi_box3d, pred_reg, loc_scope, loc_bin_size, num_head_bin, anchor_size, get_xz_fine=True,
                       get_y_by_bin=False, loc_y_scope=0.5, loc_y_bin_size=0.25, get_ry_fine=False, bbox_avg_by_bin=True,
                       ry_with_bin=False):
    """roi_box3d (N,3|7) anchors, pred_reg (N,C) -> boxes (N,7) [x, y, z, h, w, l, ry] (:25-262).
    Channel layout of pred_reg: [x bins | z bins | x residuals | z residuals]? [y offset | y bins, y residuals]
    [ry bins | ry residuals] [h, w, l residuals]"""
    anchor_size = anchor_size.to(roi_box3d.device)
    nb = int(loc_scope / loc_bin_size) * 2          # bins per axis
    nb_y = int(loc_y_scope / loc_y_bin_size) * 2

    # ---- x, z
    if bbox_avg_by_bin:
        assert get_xz_fine, 'now only support bin format!'
        centres = torch.arange(nb, device=pred_reg.device).float() * loc_bin_size + loc_bin_size / 2 - loc_scope
        pos_x = _soft_bin(pred_reg, 0, nb * 2, nb, centres, loc_bin_size)
        pos_z = _soft_bin(pred_reg, nb, nb * 3, nb, centres, loc_bin_size)
    else:
        pos_x = _hard_bin(pred_reg, 0, nb * 2, nb, loc_bin_size, loc_scope, get_xz_fine)
        pos_z = _hard_bin(pred_reg, nb, nb * 3, nb, loc_bin_size, loc_scope, get_xz_fine)
    cursor = nb * 4 if get_xz_fine else nb * 2

    # ---- y (:108-124)
    if get_y_by_bin:
        pos_y = _hard_bin(pred_reg, cursor, cursor + nb_y, nb_y, loc_y_bin_size, loc_y_scope, True) + roi_box3d[:, 1]
        cursor += nb_y * 2
    else:
        pos_y = roi_box3d[:, 1] + pred_reg[:, cursor]
        cursor += 1

    # ---- heading (:126-152)
    bin_logits = pred_reg[:, cursor:cursor + num_head_bin]
    bin_res = pred_reg[:, cursor + num_head_bin:cursor + num_head_bin * 2]
    cursor += num_head_bin * 2
    if ry_with_bin:
        # (:146-238) every bin proposes a heading; the bins fall on two sides (fine: left / right of the ROI's own heading,
        # coarse: the first / second half turn); the answer is the probability-weighted mean over the likelier side -- a mean
        # over both sides would average headings that point in opposite directions
        prob = F.softmax(bin_logits, dim=1)
        k = torch.arange(num_head_bin, device=pred_reg.device).float()
        if get_ry_fine:
            per_bin = (np.pi / 2) / num_head_bin
            each = (k * per_bin + per_bin / 2) + bin_res * (per_bin / 2) - np.pi / 4
            right = each >= 0
        else:
            per_bin = (2 * np.pi) / num_head_bin
            each = (k * per_bin + bin_res * (per_bin / 2)) % (2 * np.pi)
            right = each <= np.pi
        zero = torch.zeros_like(prob)
        p_right, p_left = torch.where(right, prob, zero), torch.where(right, zero, prob)
        mass_right, mass_left = p_right.sum(dim=1, keepdim=True) + 1e-7, p_left.sum(dim=1, keepdim=True) + 1e-7
        ry_right = (torch.where(right, each, zero) * (p_right / mass_right)).sum(dim=1)
        ry_left = (torch.where(right, zero, each) * (p_left / mass_left)).sum(dim=1)
        use_right = (mass_right >= mass_left).squeeze(1)
        ry = ry_right * use_right.float() + ry_left * (~use_right).float()
        if not get_ry_fine:
            ry = torch.where(ry > np.pi, ry - 2 * np.pi, ry)
    else:
        head = torch.argmax(bin_logits, dim=1)
        head_res = torch.gather(bin_res, dim=1, index=head.unsqueeze(dim=1)).squeeze(dim=1)
        if get_ry_fine:      # a quarter turn split into bins around the ROI's own heading
            per_bin = (np.pi / 2) / num_head_bin
            ry = (head.float() * per_bin + per_bin / 2) + head_res * (per_bin / 2) - np.pi / 4
        else:                # the full turn, wrapped into (-pi, pi]
            per_bin = (2 * np.pi) / num_head_bin
            ry = (head.float() * per_bin + head_res * (per_bin / 2)) % (2 * np.pi)
            ry = torch.where(ry > np.pi, ry - 2 * np.pi, ry)   # `ry[ry > np.pi] -= 2 * np.pi` without the mask's host sync

    # ---- size (:243-248) and back to scene coordinates (:250-262)
    assert cursor + 3 == pred_reg.shape[1]
    hwl = pred_reg[:, cursor:cursor + 3] * anchor_size + anchor_size
    box = torch.cat((pos_x.view(-1, 1), pos_y.view(-1, 1), pos_z.view(-1, 1), hwl, ry.view(-1, 1)), dim=1)
    if roi_box3d.shape[1] == 7:
        roi_ry = roi_box3d[:, 6]
        box = rotate_pc_along_y_torch(box, -roi_ry)
        box[:, 6] += roi_ry
    box[:, 0] += roi_box3d[:, 0]
    box[:, 2] += roi_box3d[:, 2]
    return box
