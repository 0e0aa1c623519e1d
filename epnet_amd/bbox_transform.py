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


def decode_bbox_target(roi_box3d, pred_reg, loc_scope, loc_bin_size, num_head_bin, anchor_size, get_xz_fine=True,
                       get_y_by_bin=False, loc_y_scope=0.5, loc_y_bin_size=0.25, get_ry_fine=False, bbox_avg_by_bin=True,
                       ry_with_bin=False):
    """roi_box3d (N,3|7) anchors, pred_reg (N,C) -> boxes (N,7) [x, y, z, h, w, l, ry] (:25-262)"""
    anchor_size = anchor_size.to(roi_box3d.device)
    per_loc_bin_num = int(loc_scope / loc_bin_size) * 2
    loc_y_bin_num = int(loc_y_scope / loc_y_bin_size) * 2
    nb = per_loc_bin_num

    if not bbox_avg_by_bin:                                   # :48-72: arg-max bin + its residual
        start_offset = nb * 2
        x_bin = torch.argmax(pred_reg[:, 0:nb], dim=1)
        z_bin = torch.argmax(pred_reg[:, nb:nb * 2], dim=1)
        pos_x = x_bin.float() * loc_bin_size + loc_bin_size / 2 - loc_scope
        pos_z = z_bin.float() * loc_bin_size + loc_bin_size / 2 - loc_scope
        if get_xz_fine:
            start_offset = nb * 4
            x_res_norm = torch.gather(pred_reg[:, nb * 2:nb * 3], dim=1, index=x_bin.unsqueeze(dim=1)).squeeze(dim=1)
            z_res_norm = torch.gather(pred_reg[:, nb * 3:nb * 4], dim=1, index=z_bin.unsqueeze(dim=1)).squeeze(dim=1)
            pos_x += x_res_norm * loc_bin_size
            pos_z += z_res_norm * loc_bin_size
    else:                                                     # :73-106: soft-max weighted mean over the bins
        start_offset = nb * 2
        pred_x_bin = F.softmax(pred_reg[:, 0:nb], 1)
        pred_z_bin = F.softmax(pred_reg[:, nb:nb * 2], 1)
        xz_bin_center = torch.arange(nb, device=pred_reg.device).float() * loc_bin_size + loc_bin_size / 2 - loc_scope
        pred_x_abs = pred_z_abs = xz_bin_center
        assert get_xz_fine, 'now only support bin format!'
        start_offset = nb * 4
        pred_x_abs = pred_x_abs + pred_reg[:, nb * 2:nb * 3] * loc_bin_size
        pred_z_abs = pred_z_abs + pred_reg[:, nb * 3:nb * 4] * loc_bin_size
        pos_x = (pred_x_abs * pred_x_bin).sum(dim=1)
        pos_z = (pred_z_abs * pred_z_bin).sum(dim=1)

    if get_y_by_bin:                                          # :109-119
        y_bin_l, y_bin_r = start_offset, start_offset + loc_y_bin_num
        y_res_l, y_res_r = y_bin_r, y_bin_r + loc_y_bin_num
        start_offset = y_res_r
        y_bin = torch.argmax(pred_reg[:, y_bin_l:y_bin_r], dim=1)
        y_res_norm = torch.gather(pred_reg[:, y_res_l:y_res_r], dim=1, index=y_bin.unsqueeze(dim=1)).squeeze(dim=1)
        pos_y = y_bin.float() * loc_y_bin_size + loc_y_bin_size / 2 - loc_y_scope + y_res_norm * loc_y_bin_size
        pos_y = pos_y + roi_box3d[:, 1]
    else:                                                     # :120-124
        pos_y = roi_box3d[:, 1] + pred_reg[:, start_offset]
        start_offset += 1

    ry_bin_l, ry_bin_r = start_offset, start_offset + num_head_bin
    ry_res_l, ry_res_r = ry_bin_r, ry_bin_r + num_head_bin
    if not ry_with_bin:                                       # :135-152
        ry_bin = torch.argmax(pred_reg[:, ry_bin_l:ry_bin_r], dim=1)
        ry_res_norm = torch.gather(pred_reg[:, ry_res_l:ry_res_r], dim=1, index=ry_bin.unsqueeze(dim=1)).squeeze(dim=1)
        if get_ry_fine:
            angle_per_class = (np.pi / 2) / num_head_bin
            ry_res = ry_res_norm * (angle_per_class / 2)
            ry = (ry_bin.float() * angle_per_class + angle_per_class / 2) + ry_res - np.pi / 4
        else:
            angle_per_class = (2 * np.pi) / num_head_bin
            ry_res = ry_res_norm * (angle_per_class / 2)
            ry = (ry_bin.float() * angle_per_class + ry_res) % (2 * np.pi)
            ry = torch.where(ry > np.pi, ry - 2 * np.pi, ry)   # `ry[ry > np.pi] -= 2 * np.pi` without the mask's host sync
    else:
        raise NotImplementedError("RY_WITH_BIN (bbox_transform.py:146-238) is off in every shipped config (lib/config.py:199,209)")

    size_res_l, size_res_r = ry_res_r, ry_res_r + 3           # :243-248
    assert size_res_r == pred_reg.shape[1]
    size_res_norm = pred_reg[:, size_res_l:size_res_r]
    hwl = size_res_norm * anchor_size + anchor_size

    roi_center = roi_box3d[:, 0:3]                            # :250-262
    shift_ret_box3d = torch.cat((pos_x.view(-1, 1), pos_y.view(-1, 1), pos_z.view(-1, 1), hwl, ry.view(-1, 1)), dim=1)
    ret_box3d = shift_ret_box3d
    if roi_box3d.shape[1] == 7:
        roi_ry = roi_box3d[:, 6]
        ret_box3d = rotate_pc_along_y_torch(shift_ret_box3d, -roi_ry)
        ret_box3d[:, 6] += roi_ry
    ret_box3d[:, 0] += roi_center[:, 0]
    ret_box3d[:, 2] += roi_center[:, 2]
    return ret_box3d
