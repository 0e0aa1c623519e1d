"""Regenerates tests/golden/ry_with_bin.npz. Runs ONLY in the build container (needs /root/reference); the fixture is plain data.

What runs is the REFERENCE'S OWN ``lib/utils/bbox_transform.py`` (decode_bbox_target), imported unmodified as in
make_golden_rcnn.py, with ``cfg.TRAIN.RY_WITH_BIN = cfg.TEST.RY_WITH_BIN = True`` (:146-238: the heading as the
probability-weighted mean of the bins on the likelier side) -- the RPN call shape (coarse heading, 3-column anchors) and the
RCNN call shape (fine heading, 7-column ROIs, rotated back)."""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import make_golden_rcnn as base  # noqa: E402


def main():
    cfg, _ptl, _pl, bt = base.import_reference()
    cfg.TRAIN.RY_WITH_BIN = cfg.TEST.RY_WITH_BIN = True
    cfg.TRAIN.BBOX_AVG_BY_BIN = cfg.TEST.BBOX_AVG_BY_BIN = False
    cfg.CLS_MEAN_SIZE = np.array([[1.52563191462, 1.62856739989, 3.88311640418]], dtype=np.float32)
    anchor = torch.from_numpy(cfg.CLS_MEAN_SIZE[0])
    g = torch.Generator().manual_seed(91)
    out = {}
    n = 700
    xyz = base.synth.scenes("kitti", 1, n, seed=92)[0]
    ch = 12 * 4 + 1 + 12 * 2 + 3
    reg = torch.randn((n, ch), generator=g) * 0.8
    reg[:40, 12 * 4 + 1:12 * 4 + 1 + 12] *= 6.0            # some sharply peaked heading distributions
    reg[40:60, 12 * 4 + 1:12 * 4 + 1 + 12] = 0.0           # and some exactly flat ones (both sides equally likely: the right side wins)
    reg = reg.half().float()                               # (the fixture stores halves: what the reference sees is exactly that)
    dec = bt.decode_bbox_target(xyz.clone(), reg.clone(), anchor_size=anchor, loc_scope=3.0, loc_bin_size=0.5, num_head_bin=12,
                                get_xz_fine=True, get_y_by_bin=False, get_ry_fine=False)
    out.update({"rpn_xyz": base._np(xyz), "rpn_reg_f16": base._np(reg.half()), "rpn_decoded": base._np(dec)})
    rois = base.scene_boxes(300, 6, 93)[0]
    ch2 = 6 * 4 + 1 + 9 * 2 + 3
    reg2 = torch.randn((300, ch2), generator=g) * 0.8
    reg2[:30, 6 * 4 + 1:6 * 4 + 1 + 9] *= 6.0
    reg2[30:45, 6 * 4 + 1:6 * 4 + 1 + 9] = 0.0
    reg2 = reg2.half().float()
    dec2 = bt.decode_bbox_target(rois.clone(), reg2.clone(), anchor_size=anchor, loc_scope=1.5, loc_bin_size=0.5, num_head_bin=9,
                                 get_xz_fine=True, get_y_by_bin=False, loc_y_scope=0.5, loc_y_bin_size=0.25, get_ry_fine=True)
    out.update({"rcnn_rois": base._np(rois), "rcnn_reg_f16": base._np(reg2.half()), "rcnn_decoded": base._np(dec2)})
    path = os.path.join(HERE, "ry_with_bin.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; heading range", float(dec[:, 6].min()), float(dec[:, 6].max()),
          float(dec2[:, 6].min()), float(dec2[:, 6].max()))


if __name__ == "__main__":
    main()
