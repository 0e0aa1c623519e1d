"""RPN proposal layer (reference: lib/rpn/proposal_layer.py:9-142) -- SURVEY.md section 8(f) row N2: the caller of
the NMS ops, with its per-scene host work moved onto the device.

Same class name, constructor argument and ``forward(rpn_scores, rpn_reg, xyz) -> (ret_bbox3d, ret_scores)`` as the
reference. The reference decodes the boxes, sorts the scores and then walks the scenes in Python: boolean-mask
indexing per distance bin, top-K slices, an NMS whose 5 MB suppression mask goes to the host for a greedy sweep under
the GIL, more slicing, ``torch.cat`` and a copy into the zero-padded result -- about ten host synchronisations per
scene. Here the decoding and the sort stay stock tensor ops and everything after them is ``epnet_rpn_proposals``
(csrc/iou3d.hip): bin compaction in score order, one batched NMS (mask + sweep) over all (scene, bin) groups with the
group sizes read from device memory, and the gather into the padded outputs. Nothing is read back to the host, so the
layer can sit inside a captured HIP graph.
"""
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

from . import iou3d_cuda
from .bbox_transform import decode_bbox_target


def default_cfg():
    """the keys this layer reads, values of tools/cfgs/LI_Fusion_with_attention_use_ce_loss.yaml (:19, :46-49, :76,
    :171-174, :180-181, :185-188); any object with the same attributes works (e.g. the reference's lib.config.cfg)"""
    rpn = SimpleNamespace(LOC_SCOPE=3.0, LOC_BIN_SIZE=0.5, NUM_HEAD_BIN=12, LOC_XZ_FINE=True, NMS_TYPE="normal")
    train = SimpleNamespace(RPN_PRE_NMS_TOP_N=9000, RPN_POST_NMS_TOP_N=512, RPN_NMS_THRESH=0.85, RPN_DISTANCE_BASED_PROPOSE=True,
                            BBOX_AVG_BY_BIN=True, RY_WITH_BIN=False)
    test = SimpleNamespace(RPN_PRE_NMS_TOP_N=9000, RPN_POST_NMS_TOP_N=100, RPN_NMS_THRESH=0.8, RPN_DISTANCE_BASED_PROPOSE=True,
                           BBOX_AVG_BY_BIN=True, RY_WITH_BIN=False)
    return SimpleNamespace(CLS_MEAN_SIZE=np.array([[1.52563191462, 1.62856739989, 3.88311640418]], dtype=np.float32),
                           RPN=rpn, TRAIN=train, TEST=test)


def _ambient_cfg():
    """the reference's global config when its module is loaded in this process (drop-in use under lib/net/*), else the
    yaml-valued defaults above"""
    import sys
    ref = sys.modules.get("lib.config")
    return ref.cfg if ref is not None and hasattr(ref, "cfg") else default_cfg()


class ProposalLayer(nn.Module):
    def __init__(self, mode='TRAIN', cfg=None):
        super().__init__()
        self.mode = mode
        self.cfg = cfg if cfg is not None else _ambient_cfg()
        self.register_buffer("MEAN_SIZE", torch.from_numpy(np.asarray(self.cfg.CLS_MEAN_SIZE[0], dtype=np.float32)), persistent=False)

    def _mode_cfg(self):
        return self.cfg[self.mode] if isinstance(self.cfg, dict) else getattr(self.cfg, self.mode)

    def decode(self, rpn_reg, xyz):
        """(B,N,C), (B,N,3) -> decoded boxes (B,N,7), y moved from the box centre to the bottom centre (:23-32)"""
        cfg, batch_size = self.cfg, xyz.shape[0]
        proposals = decode_bbox_target(xyz.reshape(-1, 3), rpn_reg.reshape(-1, rpn_reg.shape[-1]), anchor_size=self.MEAN_SIZE,
                                       loc_scope=cfg.RPN.LOC_SCOPE, loc_bin_size=cfg.RPN.LOC_BIN_SIZE,
                                       num_head_bin=cfg.RPN.NUM_HEAD_BIN, get_xz_fine=cfg.RPN.LOC_XZ_FINE, get_y_by_bin=False,
                                       get_ry_fine=False, bbox_avg_by_bin=cfg.TRAIN.BBOX_AVG_BY_BIN,
                                       ry_with_bin=cfg.TEST.RY_WITH_BIN)
        proposals[:, 1] += proposals[:, 3] / 2
        return proposals.view(batch_size, -1, 7)

    def forward(self, rpn_scores, rpn_reg, xyz):
        """rpn_scores (B,N), rpn_reg (B,N,C), xyz (B,N,3) -> ret_bbox3d (B,M,7), ret_scores (B,M), M = RPN_POST_NMS_TOP_N,
        zero rows behind the kept proposals"""
        proposals = self.decode(rpn_reg, xyz).float().contiguous()
        return self.propose(rpn_scores.float().contiguous(), proposals)

    def propose(self, scores, proposals, ret_count=None):
        """everything after the decoding (:34-55): scores (B,N), proposals (B,N,7) -> (ret_bbox3d, ret_scores)"""
        mode = self._mode_cfg()
        _, sorted_idxs = torch.sort(scores, dim=1, descending=True)                       # :35
        batch_size, post = scores.size(0), mode.RPN_POST_NMS_TOP_N
        ret_bbox3d = torch.empty((batch_size, post, 7), dtype=torch.float32, device=scores.device)
        ret_scores = torch.empty((batch_size, post), dtype=torch.float32, device=scores.device)
        distance_based = bool(self.cfg.TEST.RPN_DISTANCE_BASED_PROPOSE)                   # :45 reads cfg.TEST's switch in both modes
        if distance_based:
            if self.cfg.RPN.NMS_TYPE not in ('rotate', 'normal'):
                raise NotImplementedError                                                 # :104-109
            rotated = self.cfg.RPN.NMS_TYPE == 'rotate'
        else:
            rotated = True                                                                # score_based_proposal: nms_gpu (:137)
        iou3d_cuda.rpn_proposals_gpu(proposals, scores, sorted_idxs.contiguous(), distance_based, mode.RPN_PRE_NMS_TOP_N, post,
                                     mode.RPN_NMS_THRESH, rotated, ret_bbox3d, ret_scores, ret_count)
        return ret_bbox3d, ret_scores
