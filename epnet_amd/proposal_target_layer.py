"""RCNN target assignment for `rcnn_online` training (reference: lib/rpn/proposal_target_layer.py:10-349) --
SURVEY.md section 8(f) row N1: the caller of the iou3d / roipool3d ops, with its host loops taken off the step.

Same class name, same ``forward(input_dict) -> output_dict`` keys, shapes and value formulas as the reference. What
differs is where the work runs:

* ``aug_roi_by_noise_torch`` (:220-247) walks every sampled ROI on the host -- up to 10 tries per ROI, each a host
  coin, five tiny torch kernels for the noisy box, a single-pair ``boxes_iou3d_gpu`` and a device->host read of the
  IoU for the loop condition: up to 640 round trips per scene. Here the draws of all tries of all ROIs of the BATCH
  are tables drawn on the device and ``epnet_aug_roi_by_noise`` (csrc/iou3d.hip) runs the same loop for every ROI in
  one launch (``aug_roi_by_noise_batched``); foreground and background ROIs differ only in their try limit.
* ``sample_rois_for_rcnn`` (:85-189): the ROI x ground-truth IoU of every scene is one fused launch (no torch glue),
  the three ``nonzero`` host syncs per scene become ONE copy of the (B, M) overlap maxima per batch, and the index
  selection then runs on the host with the reference's own host random calls in the reference's order
  (``np.random.permutation`` for foreground, ``torch.randint`` on the CPU generator for background, :133,197-213).
* the canonical transformation (:51-62) and ``data_augmentation`` (:292-349) are batched tensor expressions instead
  of per-scene Python loops (the reference's loop recomputes ``ry`` of every row on every iteration from values that
  do not change once a row is rotated, so the batched form yields the same numbers).

Random streams: the reference interleaves ``np.random`` coins and device ``torch.rand`` calls inside its ROI loop, so
its stream positions depend on the IoUs of earlier tries; a batched form cannot consume a stream that way. The draws
have the same distributions (coin p = 0.2; the three ``REG_AUG_METHOD`` noise models) and are addressed [roi][try];
``tests/`` pins the loop against the reference's own method fed from the same tables.
"""
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

from . import iou3d_cuda, iou3d_utils, kitti_utils, roipool3d_utils

# pos_range, hwl_range, angle_range of random_aug_box3d's 'multiple' method (:263-267; the 4th column, mean_iou, is unused)
_RANGE_CONFIG = ((0.2, 0.1, np.pi / 12), (0.3, 0.15, np.pi / 12), (0.5, 0.15, np.pi / 9), (0.8, 0.15, np.pi / 6),
                 (1.0, 0.15, np.pi / 3))


def default_cfg():
    """the keys this layer reads, with the values of tools/cfgs/LI_Fusion_with_attention_use_ce_loss.yaml (:6-9, :85-137);
    any object with the same attributes works -- e.g. the reference's ``lib.config.cfg``"""
    rcnn = SimpleNamespace(USE_INTENSITY=False, USE_DEPTH=True, USE_RGB=False, POOL_EXTRA_WIDTH=0.2, NUM_POINTS=512,
                           ROI_FG_AUG_TIMES=10, REG_AUG_METHOD="multiple", CLS_FG_THRESH=0.6, CLS_BG_THRESH=0.45,
                           CLS_BG_THRESH_LO=0.05, REG_FG_THRESH=0.55, FG_RATIO=0.5, ROI_PER_IMAGE=64, HARD_BG_RATIO=0.8)
    return SimpleNamespace(AUG_DATA=True, AUG_ROT_RANGE=18, RCNN=rcnn)


def draw_aug_raw(num_rois, aug_times, method="multiple", device=None, generator=None):
    """the uniform draws behind every (roi, try), on `device`, no host sync: coin (K,T) for `np.random.rand() < 0.2`
    (:232), which (K,T) int64 = the row of range_config picked by 'multiple' (:268), u (K,T,7) = the three torch.rand
    calls of random_aug_box3d in their order (3 position, 3 size, 1 angle draws)"""
    k, t = int(num_rois), int(aug_times)
    coin = torch.rand((k, t), device=device, generator=generator)
    which = None
    if method == "multiple":
        which = torch.randint(0, len(_RANGE_CONFIG), (k, t), device=device, generator=generator)
    u = torch.rand((k, t, 7), device=device, generator=generator)
    return coin, which, u


def aug_tables_from_raw(coin, which, u, method="multiple"):
    """keep_draw (K,T) uint8 and noise (K,T,7) = pos_shift[3], hwl_scale[3], angle_rot with the arithmetic of
    random_aug_box3d (:250-275)"""
    keep = (coin < 0.2).to(torch.uint8)
    if method == "single":        # :255-260
        pos_shift = u[..., 0:3] - 0.5
        hwl_scale = (u[..., 3:6] - 0.5) / (0.5 / 0.15) + 1.0
        angle_rot = (u[..., 6:7] - 0.5) / (0.5 / (np.pi / 12))
    elif method == "multiple":    # :261-275
        rng = torch.tensor(_RANGE_CONFIG, dtype=torch.float32, device=u.device)[which]        # (K,T,3)
        pos_shift = ((u[..., 0:3] - 0.5) / 0.5) * rng[..., 0:1]
        hwl_scale = ((u[..., 3:6] - 0.5) / 0.5) * rng[..., 1:2] + 1.0
        angle_rot = ((u[..., 6:7] - 0.5) / 0.5) * rng[..., 2:3]
    else:
        # 'normal' (:276-289) ADDS gaussian noise to h, w, l and cannot run in the reference either (`torch.rand()`
        # without a size, :283)
        raise NotImplementedError("REG_AUG_METHOD %r" % (method,))
    return keep.contiguous(), torch.cat([pos_shift, hwl_scale, angle_rot], dim=-1).float().contiguous()


def draw_aug_tables(num_rois, aug_times, method="multiple", device=None, generator=None):
    """random draws for every (roi, try): keep_draw (K,T) uint8 and noise (K,T,7) for epnet_aug_roi_by_noise"""
    return aug_tables_from_raw(*draw_aug_raw(num_rois, aug_times, method, device, generator), method=method)


def aug_roi_by_noise_batched(roi_boxes3d, gt_boxes3d, iou3d_src, pos_thresh, keep_draw, noise, tries=None):
    """aug_roi_by_noise_torch (:220-247) for all K ROIs in one launch. roi_boxes3d (K,7) is updated in place as in the
    reference (:242); returns (roi_boxes3d, iou_of_rois (K))."""
    iou_of_rois = torch.empty((roi_boxes3d.shape[0],), dtype=torch.float32, device=roi_boxes3d.device)
    iou3d_cuda.aug_roi_by_noise_gpu(roi_boxes3d, gt_boxes3d.float().contiguous(), iou3d_src.float().contiguous(), keep_draw,
                                    noise, pos_thresh, iou_of_rois, tries)
    return roi_boxes3d, iou_of_rois


def rotate_along_y(xz_rows, angle):
    """rows (..., S, 3+) rotated in place about y by angle (...): the reference's rotate_pc_along_y_torch
    (lib/utils/kitti_utils.py:45-63) for any number of leading dimensions"""
    flat = xz_rows.reshape(-1, xz_rows.shape[-2], xz_rows.shape[-1])
    kitti_utils.rotate_pc_along_y_torch(flat, angle.reshape(-1))
    if flat.data_ptr() != xz_rows.data_ptr():  # reshape had to copy (non-contiguous rows): write the result back
        xz_rows.copy_(flat.view(xz_rows.shape))
    return xz_rows


def _ambient_cfg():
    """the reference's global config when its module is loaded in this process (drop-in use under lib/net/*), else the
    yaml-valued defaults above"""
    import sys
    ref = sys.modules.get("lib.config")
    return ref.cfg if ref is not None and hasattr(ref, "cfg") else default_cfg()


class ProposalTargetLayer(nn.Module):
    def __init__(self, cfg=None, generator=None):
        super().__init__()
        self.cfg = cfg if cfg is not None else _ambient_cfg()
        self.generator = generator  # device generator for the augmentation draws (None: the default one)

    # ------------------------------------------------------------------------------------------------------ forward
    def forward(self, input_dict):
        cfg = self.cfg
        roi_boxes3d, gt_boxes3d = input_dict['roi_boxes3d'], input_dict['gt_boxes3d']
        batch_rois, batch_gt_of_rois, batch_roi_iou = self.sample_rois_for_rcnn(roi_boxes3d, gt_boxes3d)

        rpn_xyz, rpn_features = input_dict['rpn_xyz'], input_dict['rpn_features']
        if cfg.RCNN.USE_INTENSITY:
            extra = [input_dict['rpn_intensity'].unsqueeze(dim=2), input_dict['seg_mask'].unsqueeze(dim=2)]
        else:
            extra = [input_dict['seg_mask'].unsqueeze(dim=2)]
        if cfg.RCNN.USE_DEPTH:
            extra.append((input_dict['pts_depth'] / 70.0 - 0.5).unsqueeze(dim=2))
        if cfg.RCNN.USE_RGB:
            extra.append(input_dict['pts_rgb'])
        pts_feature = torch.cat(extra + [rpn_features], dim=2)

        pooled_features, pooled_empty_flag = roipool3d_utils.roipool3d_gpu(
            rpn_xyz, pts_feature, batch_rois, cfg.RCNN.POOL_EXTRA_WIDTH, sampled_pt_num=cfg.RCNN.NUM_POINTS)
        sampled_pts, sampled_features = pooled_features[:, :, :, 0:3], pooled_features[:, :, :, 3:]
        mask_score = pooled_features[:, :, :, 3].sum(-1) / cfg.RCNN.NUM_POINTS

        if cfg.AUG_DATA:
            sampled_pts, batch_rois, batch_gt_of_rois = self.data_augmentation(sampled_pts, batch_rois, batch_gt_of_rois)

        # canonical transformation (:51-62), all scenes at once
        roi_ry = batch_rois[:, :, 6] % (2 * np.pi)
        roi_center = batch_rois[:, :, 0:3]
        sampled_pts = sampled_pts - roi_center.unsqueeze(dim=2)
        batch_gt_of_rois[:, :, 0:3] = batch_gt_of_rois[:, :, 0:3] - roi_center
        batch_gt_of_rois[:, :, 6] = batch_gt_of_rois[:, :, 6] - roi_ry
        rotate_along_y(sampled_pts, batch_rois[:, :, 6])
        rotate_along_y(batch_gt_of_rois.unsqueeze(dim=2), roi_ry)

        valid_mask = (pooled_empty_flag == 0)
        reg_valid_mask = ((batch_roi_iou > cfg.RCNN.REG_FG_THRESH) & valid_mask).long()
        batch_cls_label = (batch_roi_iou > cfg.RCNN.CLS_FG_THRESH).long()
        invalid_mask = (batch_roi_iou > cfg.RCNN.CLS_BG_THRESH) & (batch_roi_iou < cfg.RCNN.CLS_FG_THRESH)
        batch_cls_label[valid_mask == 0] = -1
        batch_cls_label[invalid_mask > 0] = -1

        return {'sampled_pts': sampled_pts.reshape(-1, cfg.RCNN.NUM_POINTS, 3),
                'pts_feature': sampled_features.reshape(-1, cfg.RCNN.NUM_POINTS, sampled_features.shape[3]),
                'cls_label': batch_cls_label.view(-1),
                'mask_score': mask_score.view(-1),
                'reg_valid_mask': reg_valid_mask.view(-1),
                'gt_of_rois': batch_gt_of_rois.view(-1, 7),
                'gt_iou': batch_roi_iou.view(-1),
                'roi_boxes3d': batch_rois.view(-1, 7)}

    # ----------------------------------------------------------------------------------------------------- sampling
    def sample_rois_for_rcnn(self, roi_boxes3d, gt_boxes3d):
        """roi_boxes3d (B,M,7), gt_boxes3d (B,N,7+) zero-padded -> batch_rois (B,R,7), batch_gt_of_rois (B,R,7),
        batch_roi_iou (B,R), R = ROI_PER_IMAGE (:85-189)"""
        cfg = self.cfg.RCNN
        batch_size, num_roi = roi_boxes3d.size(0), roi_boxes3d.size(1)
        per_image = cfg.ROI_PER_IMAGE
        fg_rois_per_image = int(np.round(cfg.FG_RATIO * per_image))
        fg_thresh = min(cfg.REG_FG_THRESH, cfg.CLS_FG_THRESH)
        device = roi_boxes3d.device
        rois = roi_boxes3d.float().contiguous()
        gts = gt_boxes3d[:, :, 0:7].float().contiguous()

        # trailing all-zero ground-truth rows are padding (:105-108): one host copy for the whole batch
        row_sum = gt_boxes3d.sum(dim=2)
        num_gt = []
        for sums in (row_sum != 0).cpu().numpy():
            nz = np.nonzero(sums)[0]
            assert nz.size > 0, "a scene without ground-truth boxes (the reference's loop would index below 0, :106-107)"
            num_gt.append(int(nz[-1]) + 1)

        max_overlaps = torch.empty((batch_size, num_roi), dtype=torch.float32, device=device)
        gt_assignment = torch.empty((batch_size, num_roi), dtype=torch.int64, device=device)
        for b in range(batch_size):
            iou3d = iou3d_utils.boxes_iou3d_gpu(rois[b], gts[b, :num_gt[b]].contiguous())   # :111
            best = iou3d.max(dim=1).values                                                  # :113
            # index of the FIRST ground truth reaching the maximum (torch.max leaves ties open on the GPU)
            cols = torch.arange(num_gt[b], device=device).unsqueeze(0).expand_as(iou3d)
            first = torch.where(iou3d == best.unsqueeze(1), cols, torch.full_like(cols, num_gt[b])).min(dim=1).values
            max_overlaps[b], gt_assignment[b] = best, first.clamp(max=num_gt[b] - 1)
        overlaps_host = max_overlaps.cpu().numpy()                                     # the batch's one sync

        src_inds = np.empty((batch_size, per_image), dtype=np.int64)
        tries = np.empty((batch_size, per_image), dtype=np.int32)
        bg_aug_times = 1 if cfg.ROI_FG_AUG_TIMES > 0 else 0                            # :174
        for b in range(batch_size):
            ov = overlaps_host[b]
            fg_inds = np.nonzero(ov >= fg_thresh)[0]                                   # :117
            easy_bg_inds = np.nonzero(ov < cfg.CLS_BG_THRESH_LO)[0]                    # :122
            hard_bg_inds = np.nonzero((ov < cfg.CLS_BG_THRESH) & (ov >= cfg.CLS_BG_THRESH_LO))[0]
            fg_num, bg_num = fg_inds.size, hard_bg_inds.size + easy_bg_inds.size
            if fg_num > 0 and bg_num > 0:                                              # :129-138
                fg_this = min(fg_rois_per_image, fg_num)
                rand_num = np.random.permutation(fg_num)
                fg_inds = fg_inds[rand_num[:fg_this]]
                bg_inds = self.sample_bg_inds(hard_bg_inds, easy_bg_inds, per_image - fg_this)
            elif fg_num > 0 and bg_num == 0:                                           # :140-146
                rand_num = np.floor(np.random.rand(per_image) * fg_num).astype(np.int64)
                fg_inds = fg_inds[rand_num]
                fg_this, bg_inds = per_image, np.empty((0,), np.int64)
            elif bg_num > 0 and fg_num == 0:                                           # :147-151
                bg_inds = self.sample_bg_inds(hard_bg_inds, easy_bg_inds, per_image)
                fg_this, fg_inds = 0, np.empty((0,), np.int64)
            else:
                raise NotImplementedError("no foreground and no background ROI (the reference stops in pdb here, :153-156)")
            src_inds[b] = np.concatenate([fg_inds, bg_inds])
            tries[b, :fg_this] = cfg.ROI_FG_AUG_TIMES
            tries[b, fg_this:] = bg_aug_times

        inds = torch.from_numpy(src_inds).to(device)
        tries_dev = torch.from_numpy(tries).to(device)
        batch_rois = torch.gather(rois, 1, inds.unsqueeze(2).expand(-1, -1, 7)).contiguous()        # cur_roi[fg/bg_inds]
        assigned = torch.gather(gt_assignment, 1, inds)
        batch_gt_of_rois = torch.gather(gts, 1, assigned.unsqueeze(2).expand(-1, -1, 7)).contiguous()  # cur_gt[gt_assignment[..]]
        iou3d_src = torch.gather(max_overlaps, 1, inds).contiguous()

        # augment the rois by noise (:158-179): every ROI of the batch in one launch
        total, aug_times = batch_size * per_image, int(cfg.ROI_FG_AUG_TIMES)
        flat_rois = batch_rois.view(total, 7)
        if aug_times > 0:
            keep_draw, noise = draw_aug_tables(total, aug_times, cfg.REG_AUG_METHOD, device, self.generator)
            _, batch_roi_iou = aug_roi_by_noise_batched(flat_rois, batch_gt_of_rois.view(total, 7), iou3d_src.view(total),
                                                        fg_thresh, keep_draw, noise, tries_dev.view(total))
            batch_roi_iou = batch_roi_iou.view(batch_size, per_image)
        else:
            batch_roi_iou = iou3d_src
        return batch_rois, batch_gt_of_rois, batch_roi_iou

    @staticmethod
    def _randint(high, size):
        return torch.randint(low=0, high=int(high), size=(int(size),)).numpy()  # CPU default generator, as :197,201

    def sample_bg_inds(self, hard_bg_inds, easy_bg_inds, bg_rois_per_this_image):
        """:191-218 on host index arrays, same random calls in the same order"""
        if hard_bg_inds.size > 0 and easy_bg_inds.size > 0:
            hard_num = int(bg_rois_per_this_image * self.cfg.RCNN.HARD_BG_RATIO)
            easy_num = bg_rois_per_this_image - hard_num
            hard = hard_bg_inds[self._randint(hard_bg_inds.size, hard_num)]
            easy = easy_bg_inds[self._randint(easy_bg_inds.size, easy_num)]
            return np.concatenate([hard, easy])
        if hard_bg_inds.size > 0:
            return hard_bg_inds[self._randint(hard_bg_inds.size, bg_rois_per_this_image)]
        if easy_bg_inds.size > 0:
            return easy_bg_inds[self._randint(easy_bg_inds.size, bg_rois_per_this_image)]
        raise NotImplementedError

    # ------------------------------------------------------------------------------------------------- augmentation
    def data_augmentation(self, pts, rois, gt_of_rois, draws=None):
        """pts (B,M,S,3), rois (B,M,7), gt_of_rois (B,M,7) -> rotated / scaled / flipped copies (:292-349).
        draws = (rot_u, scale_u, flip_u), three (B,M) uniform[0,1) tensors (None: drawn here, in that order)"""
        batch_size, boxes_num = pts.shape[0], pts.shape[1]
        if draws is None:
            draws = [torch.rand((batch_size, boxes_num), device=pts.device, generator=self.generator) for _ in range(3)]
        rot_u, scale_u, flip_u = draws
        # `- 0.5 / 0.5` is the reference's expression (:302): the draw minus ONE, not a centred draw
        angles = (rot_u - 0.5 / 0.5) * (np.pi / self.cfg.AUG_ROT_RANGE)

        def alpha_of(boxes):   # :305-311
            beta = torch.atan2(boxes[:, :, 2], boxes[:, :, 0])
            return -torch.sign(beta) * np.pi / 2 + beta + boxes[:, :, 6]
        gt_alpha, roi_alpha = alpha_of(gt_of_rois), alpha_of(rois)

        pts = rotate_along_y(pts.contiguous(), angles)
        rotate_along_y(gt_of_rois.unsqueeze(dim=2), angles)
        rotate_along_y(rois.unsqueeze(dim=2), angles)
        for boxes, alpha in ((gt_of_rois, gt_alpha), (rois, roi_alpha)):     # :319-327
            beta = torch.atan2(boxes[:, :, 2], boxes[:, :, 0])
            boxes[:, :, 6] = torch.sign(beta) * np.pi / 2 + alpha - beta

        scales = 1 + ((scale_u - 0.5) / 0.5) * 0.05                          # :329-332
        pts = pts * scales.unsqueeze(dim=2).unsqueeze(dim=3)
        gt_of_rois[:, :, 0:6] = gt_of_rois[:, :, 0:6] * scales.unsqueeze(dim=2)
        rois[:, :, 0:6] = rois[:, :, 0:6] * scales.unsqueeze(dim=2)

        flip_flag = torch.sign(flip_u - 0.5)                                  # :335-347
        pts[:, :, :, 0] = pts[:, :, :, 0] * flip_flag.unsqueeze(dim=2)
        for boxes in (gt_of_rois, rois):
            boxes[:, :, 0] = boxes[:, :, 0] * flip_flag
            src_ry = boxes[:, :, 6]
            boxes[:, :, 6] = (flip_flag == 1).float() * src_ry + (flip_flag == -1).float() * (torch.sign(src_ry) * np.pi - src_ry)
        return pts, rois, gt_of_rois
