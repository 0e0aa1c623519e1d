"""Regenerates tests/golden/proposal_target.npz and proposal_layer.npz. Runs ONLY in the build container (needs
/root/reference); the fixtures are plain data.

What runs is the REFERENCE'S OWN Python, imported unmodified from /root/reference:
``lib/rpn/proposal_target_layer.py`` (ProposalTargetLayer), ``lib/rpn/proposal_layer.py`` (ProposalLayer),
``lib/utils/bbox_transform.py`` (decode_bbox_target), over ``lib/config.py``. As in make_golden.py the three CUDA
extensions are stand-ins backed by the CPU oracle (the reference has no CPU kernels for them), tensors stay on the
CPU (``Tensor.cuda`` = identity) and ``easydict`` -- a third-party package this image lacks, 20 lines of attribute
access over a dict -- is a process-local stand-in.

Randomness: the reference's ROI augmentation loop draws from ``np.random.rand`` / ``torch.rand`` / ``torch.randint``
as it goes. To pin the loop against given draws, those three functions are replaced WHILE THE REFERENCE METHOD RUNS by
readers of a table addressed [roi][try] (the reference code itself is untouched); the fixture stores the table, the
inputs and what the reference returned.
"""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True  # importing the reference must not leave __pycache__ files in /root/reference

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

from epnet_amd import synth  # noqa: E402
from epnet_amd import proposal_target_layer as mine  # noqa: E402  (only its draw helpers: tables of uniform numbers)
import oracle_ext  # noqa: E402

REF = "/root/reference"


def _np(t):
    return t.detach().cpu().numpy()


def install_easydict():
    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in dict(d or {}, **kw).items():
                setattr(self, k, v)

        def __setattr__(self, k, v):
            if isinstance(v, dict) and not isinstance(v, EasyDict):
                v = EasyDict(v)
            self[k] = v

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)
    mod = types.ModuleType("easydict")
    mod.EasyDict = EasyDict
    sys.modules["easydict"] = mod


def import_reference():
    torch.cuda.FloatTensor = torch.FloatTensor
    torch.cuda.IntTensor = torch.IntTensor
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.Tensor.get_device = lambda self: self.device  # `anchor_size.to(roi_box3d.get_device())`, bbox_transform.py:41: -1 on the CPU
    install_easydict()
    oracle_ext.install_as_top_level()
    sys.path.insert(0, REF)
    from lib.config import cfg
    import lib.rpn.proposal_target_layer as ptl
    import lib.rpn.proposal_layer as pl
    import lib.utils.bbox_transform as bt
    return cfg, ptl, pl, bt


class TableRandom:
    """feeds the reference's aug loop (proposal_target_layer.py:229-241, 250-274) from tables: the coin of try `cnt`
    from coin[k, cnt], the range_config row from which[k, cnt], the three torch.rand calls from u[k, cnt, 0:3 / 3:6 / 6:7]"""

    def __init__(self, coin, which, u):
        self.coin, self.which, self.u = coin, which, u
        self.k, self.cnt, self.col = 0, -1, 0
        self.saved = None

    def start_roi(self, k):
        self.k, self.cnt, self.col = k, -1, 0

    def np_rand(self, *shape):
        assert not shape
        self.cnt += 1
        self.col = 0
        return float(self.coin[self.k, self.cnt])

    def torch_rand(self, *size, **kw):
        n = int(size[0])
        out = self.u[self.k, self.cnt, self.col:self.col + n].clone()
        self.col += n
        return out

    def torch_randint(self, low=0, high=None, size=None, **kw):
        return self.which[self.k, self.cnt].view(1)

    def __enter__(self):
        self.saved = (np.random.rand, torch.rand, torch.randint)
        np.random.rand, torch.rand, torch.randint = self.np_rand, self.torch_rand, self.torch_randint
        return self

    def __exit__(self, *exc):
        np.random.rand, torch.rand, torch.randint = self.saved
        return False


class QueueRandom:
    """torch.rand returns the queued tensors one after another (data_augmentation's three (B,M) draws, :302,329,335)"""

    def __init__(self, tensors):
        self.queue = list(tensors)
        self.saved = None

    def __enter__(self):
        self.saved = torch.rand
        torch.rand = lambda *a, **k: self.queue.pop(0).clone()
        return self

    def __exit__(self, *exc):
        torch.rand = self.saved
        return False


def scene_boxes(num_roi, num_gt, seed):
    """proposals scattered around ground-truth boxes so that all of fg / hard bg / easy bg occur"""
    g = torch.Generator().manual_seed(seed)
    gt, _ = synth.proposal_boxes(num_gt, seed=seed, num_objects=num_gt)
    src = gt[torch.randint(0, num_gt, (num_roi,), generator=g)]
    spread = torch.rand((num_roi, 1), generator=g) ** 2
    noise = (torch.rand((num_roi, 7), generator=g) - 0.5) * torch.tensor([3.0, 0.6, 3.0, 0.5, 0.5, 1.2, 1.5]) * spread
    rois = (src + noise).float()
    rois[:, 3:6] = rois[:, 3:6].clamp(min=0.5)
    return rois.contiguous(), gt.float().contiguous()


def proposal_layer_fixture(cfg, pl, bt):
    """the reference's ProposalLayer (lib/rpn/proposal_layer.py) and decode_bbox_target (lib/utils/bbox_transform.py) on
    small scenes: distance-based with axis-aligned NMS (TRAIN), distance-based with rotated NMS (TEST), score-based; one
    scene has no point beyond 40 m (the far bin falls back to the near bin's next boxes, :92-100)"""
    out = {}
    g = torch.Generator().manual_seed(81)
    b, n = 2, 1024
    cfg.RPN.LOC_XZ_FINE, cfg.RPN.LOC_SCOPE, cfg.RPN.LOC_BIN_SIZE, cfg.RPN.NUM_HEAD_BIN = True, 3.0, 0.5, 12
    cfg.CLS_MEAN_SIZE = np.array([[1.52563191462, 1.62856739989, 3.88311640418]], dtype=np.float32)
    xyz = synth.scenes("kitti", b, n, seed=82)
    xyz[1, :, 2] = xyz[1, :, 2] * 0.5 + 0.5          # scene 1: everything within 40 m
    channels = 12 * 4 + 1 + 12 * 2 + 3
    rpn_reg = (torch.randn((b, n, channels), generator=g) * 0.6).half().float()
    rpn_scores = torch.randn((b, n), generator=g)
    out.update({"xyz": _np(xyz), "rpn_reg_f16": _np(rpn_reg.half()), "rpn_scores": _np(rpn_scores)})
    for avg in (True, False):
        cfg.TRAIN.BBOX_AVG_BY_BIN = cfg.TEST.BBOX_AVG_BY_BIN = avg
        dec = bt.decode_bbox_target(xyz.view(-1, 3).clone(), rpn_reg.view(-1, channels).clone(), anchor_size=torch.from_numpy(cfg.CLS_MEAN_SIZE[0]),
                                    loc_scope=3.0, loc_bin_size=0.5, num_head_bin=12, get_xz_fine=True, get_y_by_bin=False, get_ry_fine=False)
        out["decode_rpn_avg%d" % avg] = _np(dec)
    # the RCNN-stage call shape (tools/eval_rcnn.py:568-575): 7-column anchors, rotated back by the ROI's ry, y by offset, fine ry
    cfg.TRAIN.BBOX_AVG_BY_BIN = cfg.TEST.BBOX_AVG_BY_BIN = False
    rois = scene_boxes(256, 6, 83)[0]
    reg_ch = 6 * 4 + 1 + 9 * 2 + 3
    rcnn_reg = (torch.randn((256, reg_ch), generator=g) * 0.6).half().float()
    dec = bt.decode_bbox_target(rois.clone(), rcnn_reg.clone(), anchor_size=torch.from_numpy(cfg.CLS_MEAN_SIZE[0]), loc_scope=1.5,
                                loc_bin_size=0.5, num_head_bin=9, get_xz_fine=True, get_y_by_bin=False, loc_y_scope=0.5,
                                loc_y_bin_size=0.25, get_ry_fine=True)
    out.update({"rcnn_rois": _np(rois), "rcnn_reg_f16": _np(rcnn_reg.half()), "decode_rcnn": _np(dec)})
    cfg.TRAIN.BBOX_AVG_BY_BIN = cfg.TEST.BBOX_AVG_BY_BIN = True

    cases = (("train_normal", "TRAIN", True, "normal", 600, 64, 0.85), ("test_rotate", "TEST", True, "rotate", 300, 40, 0.5),
             ("train_tight", "TRAIN", True, "normal", 900, 128, 0.3), ("score_based", "TEST", False, "normal", 500, 48, 0.6), ("padded_normal", "TRAIN", True, "normal", 900, 800, 0.05),
             ("padded_rotate", "TEST", True, "rotate", 1000, 900, 0.02), ("padded_score", "TRAIN", False, "normal", 700, 650, 0.03))
    for tag, mode, dist_based, nms_type, pre, post, thresh in cases:
        cfg.TEST.RPN_DISTANCE_BASED_PROPOSE = dist_based
        cfg.RPN.NMS_TYPE = nms_type
        cfg[mode].RPN_PRE_NMS_TOP_N, cfg[mode].RPN_POST_NMS_TOP_N, cfg[mode].RPN_NMS_THRESH = pre, post, thresh
        layer = pl.ProposalLayer(mode=mode)
        boxes, scores = layer(rpn_scores.clone(), rpn_reg.clone(), xyz.clone())
        out["%s__bbox3d" % tag], out["%s__scores" % tag] = _np(boxes), _np(scores)
        out["%s__cfg" % tag] = np.array([mode == "TRAIN", dist_based, nms_type == "rotate", pre, post, thresh], dtype=np.float64)
        print(tag, "kept per scene", [int((boxes[i].abs().sum(1) > 0).sum()) for i in range(b)], "of", post)
    np.savez_compressed(os.path.join(HERE, "proposal_layer.npz"), **out)
    print("wrote proposal_layer.npz", os.path.getsize(os.path.join(HERE, "proposal_layer.npz")), "bytes")


def main():
    assert os.path.isdir(REF), "needs the reference checkout"
    cfg, ptl, pl, bt = import_reference()
    out = {}
    layer = ptl.ProposalTargetLayer()

    # the yaml's values for the keys the layer reads (tools/cfgs/LI_Fusion_with_attention_use_ce_loss.yaml:85-137)
    cfg.RCNN.REG_AUG_METHOD = 'multiple'
    cfg.RCNN.ROI_FG_AUG_TIMES = 10
    cfg.RCNN.CLS_FG_THRESH, cfg.RCNN.CLS_BG_THRESH, cfg.RCNN.CLS_BG_THRESH_LO, cfg.RCNN.REG_FG_THRESH = 0.6, 0.45, 0.05, 0.55
    cfg.RCNN.FG_RATIO, cfg.RCNN.HARD_BG_RATIO = 0.5, 0.8
    cfg.RCNN.USE_INTENSITY, cfg.RCNN.USE_DEPTH, cfg.RCNN.USE_RGB = False, True, False
    cfg.RCNN.POOL_EXTRA_WIDTH = 0.2
    cfg.AUG_DATA, cfg.AUG_ROT_RANGE = True, 18

    # ---- (A) aug_roi_by_noise_torch, ROI by ROI, fed from tables ('multiple' and 'single', 10 tries and 1 try)
    for tag, method, aug_times, seed in (("multiple", "multiple", 10, 31), ("single", "single", 10, 32), ("onetry", "multiple", 1, 33)):
        cfg.RCNN.REG_AUG_METHOD = method
        rois, gts = scene_boxes(96, 6, seed)
        g = torch.Generator().manual_seed(seed)
        assigned = gts[torch.randint(0, gts.shape[0], (rois.shape[0],), generator=g)].contiguous()
        iou_src = torch.rand((rois.shape[0],), generator=g)
        coin, which, u = mine.draw_aug_raw(rois.shape[0], aug_times, method, None, g)
        if which is None:
            which = torch.zeros((rois.shape[0], aug_times), dtype=torch.int64)
        out_rois, out_iou = rois.clone(), torch.zeros(rois.shape[0])
        with TableRandom(coin, which, u) as feed:
            for k in range(rois.shape[0]):
                feed.start_roi(k)
                r, i = layer.aug_roi_by_noise_torch(rois[k:k + 1].clone(), assigned[k:k + 1], iou_src[k:k + 1], aug_times=aug_times)
                out_rois[k], out_iou[k] = r[0], i[0]
        pre = "aug_%s__" % tag
        out.update({pre + "rois": _np(rois), pre + "gts": _np(assigned), pre + "iou_src": _np(iou_src), pre + "coin": _np(coin),
                    pre + "which": _np(which), pre + "u": _np(u), pre + "out_rois": _np(out_rois), pre + "out_iou": _np(out_iou)})
        print(tag, "changed", int((out_rois != rois).any(1).sum()), "of", rois.shape[0], "mean iou", float(out_iou.mean()))
    cfg.RCNN.REG_AUG_METHOD = 'multiple'

    # ---- (B) data_augmentation with given draws
    g = torch.Generator().manual_seed(41)
    b, m, s = 2, 12, 16
    pts = (torch.rand((b, m, s, 3), generator=g) - 0.5) * 6
    rois = torch.stack([scene_boxes(m, 4, 42 + i)[0] for i in range(b)])
    gt_of = torch.stack([scene_boxes(m, 4, 52 + i)[0] for i in range(b)])
    draws = [torch.rand((b, m), generator=g) for _ in range(3)]
    with QueueRandom(draws):
        o_pts, o_rois, o_gt = layer.data_augmentation(pts.clone(), rois.clone(), gt_of.clone())
    out.update({"da__pts": _np(pts), "da__rois": _np(rois), "da__gt_of_rois": _np(gt_of), "da__draws": _np(torch.stack(draws)),
                "da__out_pts": _np(o_pts), "da__out_rois": _np(o_rois), "da__out_gt_of_rois": _np(o_gt)})

    # ---- (C) sample_rois_for_rcnn with the augmentation switched to "keep every ROI" (identity), seeded host streams
    cfg.RCNN.ROI_PER_IMAGE = 16
    b, m, n_gt_pad = 3, 128, 8
    roi_list, gt_list = [], []
    for i, n_gt in enumerate((5, 2, 7)):
        r, gt = scene_boxes(m, n_gt, 60 + i)
        gt8 = torch.zeros((n_gt_pad, 7))   # (the docstring of :88 says 8 columns; the code only runs with 7, :186)
        gt8[:n_gt] = gt
        roi_list.append(r)
        gt_list.append(gt8)
    roi_boxes3d, gt_boxes3d = torch.stack(roi_list), torch.stack(gt_list)
    identity = lambda roi, gt, iou_src, aug_times=10: (roi, iou_src)   # noqa: E731
    layer.aug_roi_by_noise_torch = identity
    np.random.seed(7)
    torch.manual_seed(7)
    s_rois, s_gt, s_iou = layer.sample_rois_for_rcnn(roi_boxes3d.clone(), gt_boxes3d.clone())
    del layer.aug_roi_by_noise_torch
    out.update({"smp__roi_boxes3d": _np(roi_boxes3d), "smp__gt_boxes3d": _np(gt_boxes3d), "smp__out_rois": _np(s_rois),
                "smp__out_gt_of_rois": _np(s_gt), "smp__out_iou": _np(s_iou)})
    print("sampling: fg per scene", [(int((s_iou[i] >= 0.55).sum())) for i in range(b)])

    # ---- (D) forward after the sampling: pooling, augmentation, canonical transform, labels
    cfg.RCNN.NUM_POINTS = 32
    g = torch.Generator().manual_seed(71)
    n_pts, c_feat = 2048, 5
    rpn_xyz = synth.scenes("kitti", b, n_pts, seed=72)
    # make sure some ROIs hold points: centre a few ROIs on points of the cloud
    fwd_rois = s_rois.clone()
    for i in range(b):
        pick = torch.randint(0, n_pts, (8,), generator=g)
        fwd_rois[i, :8, 0:3] = rpn_xyz[i, pick] + torch.tensor([0.0, 0.8, 0.0])
    fwd_iou = torch.rand((b, 16), generator=g)
    inputs = {"roi_boxes3d": roi_boxes3d, "gt_boxes3d": gt_boxes3d, "rpn_xyz": rpn_xyz,
              "rpn_features": torch.randn((b, n_pts, c_feat), generator=g), "seg_mask": (torch.rand((b, n_pts), generator=g) > 0.5).float(),
              "pts_depth": torch.rand((b, n_pts), generator=g) * 70}
    layer.sample_rois_for_rcnn = lambda r, gt: (fwd_rois.clone(), s_gt.clone(), fwd_iou.clone())
    draws = [torch.rand((b, 16), generator=g) for _ in range(3)]
    with QueueRandom(draws):
        res = layer.forward({k: v.clone() for k, v in inputs.items()})
    for k, v in inputs.items():
        out["fwd__in_" + k] = _np(v)
    out.update({"fwd__sampled_rois": _np(fwd_rois), "fwd__sampled_gt": _np(s_gt), "fwd__sampled_iou": _np(fwd_iou),
                "fwd__draws": _np(torch.stack(draws))})
    for k, v in res.items():
        out["fwd__out_" + k] = _np(v)
    print("forward: non-empty ROIs", int((res["cls_label"] >= 0).sum()), "of", res["cls_label"].numel())

    proposal_layer_fixture(cfg, pl, bt)

    np.savez_compressed(os.path.join(HERE, "proposal_target.npz"), **out)
    print("wrote proposal_target.npz", os.path.getsize(os.path.join(HERE, "proposal_target.npz")), "bytes")


if __name__ == "__main__":
    main()
