"""Set-abstraction (SA) and feature-propagation (FP) modules over the gfx950 operators.

Same public classes, constructor keywords, return values and ``state_dict`` key names as the
reference's pointnet2_lib/pointnet2/pointnet2_modules.py (``_PointnetSAModuleBase``:10,
``PointnetSAModuleMSG``:75, ``PointnetSAModule``:112, ``PointnetFPModule``:133), so lib/net's
``Pointnet2MSG`` / ``RCNNNet`` build on it unchanged and reference checkpoints load. The geometry
ops (FPS, gather, ball query, grouping, three_nn, three_interpolate) and the neighbourhood max-pool run on
the HIP kernels; the shared MLPs stay on stock PyTorch-ROCm, the first 1x1 convolution of a level being applied to
the N points before the grouping instead of to the npoint * nsample grouped columns after it.
"""
import os
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pointnet2_utils
from . import pytorch_utils as pt_utils


def _foldable(grouper, mlp, features) -> bool:
    """can the first layer of `mlp` be folded into the grouping? A ball-query grouper that prepends xyz, features present,
    and a first unit that starts with a plain 1x1 convolution over [xyz(3) ; features(C)]"""
    if not isinstance(grouper, pointnet2_utils.QueryAndGroup) or not grouper.use_xyz or features is None:
        return False
    if features.dtype != torch.float32 or len(mlp) == 0:
        return False
    first = mlp[0]
    children = list(first.children())
    conv = getattr(first, "conv", None)
    if not isinstance(conv, nn.Conv2d) or not children or children[0] is not conv:   # pre-activated units start with bn / act
        return False
    return (conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0) and conv.groups == 1
            and conv.in_channels == features.shape[1] + 3)


class _PointnetSAModuleBase(nn.Module):
    def __init__(self):
        super().__init__()
        self.npoint = None
        self.groupers = None
        self.mlps = None
        self.pool_method = 'max_pool'
        # fold the first 1x1 convolution of every scale into its grouping (EPNET_SA_FOLD_FIRST_LAYER=0: build the grouped
        # tensor and convolve it, as the reference does)
        self.fold_first_layer = os.environ.get("EPNET_SA_FOLD_FIRST_LAYER", "1") != "0"

    def _pool(self, x: torch.Tensor) -> torch.Tensor:
        window = [1, x.size(3)]
        if self.pool_method == 'max_pool':
            if x.is_cuda and x.dtype == torch.float32 and x.size(3) > 0:
                return pointnet2_utils.pool_max(x)  # same values as the stock op below, rows read with coalesced loads
            return F.max_pool2d(x, kernel_size=window)
        if self.pool_method == 'avg_pool':
            return F.avg_pool2d(x, kernel_size=window)
        raise NotImplementedError

    def forward(self, xyz: torch.Tensor, features: Optional[torch.Tensor] = None, new_xyz=None, presampled=None):
        """xyz (B,N,3), features (B,C,N) -> (new_xyz (B,npoint,3), new_features (B,sum mlp[-1],npoint),
        idx (B,npoint) int32 FPS indices or None). The 3-tuple (reference :72) is what
        lib/net/pointnet2_msg.py:214-218 unpacks; LI-Fusion consumes idx."""
        idx = index = None
        if presampled is not None:   # (idx, new_xyz, event, index) of pointnet2_utils.sample_pyramid: the level's sampling, done up front
            idx, new_xyz, ready, index = presampled
            assert new_xyz.shape[1] == self.npoint
            torch.cuda.current_stream(xyz.device).wait_event(ready)
        # one spatial sort of the level's points serves the sampling and every ball query of the level
        if index is None and self.npoint is not None and xyz.is_cuda:
            index = pointnet2_utils.scene_index(xyz)
        if new_xyz is None and self.npoint is not None:
            if xyz.is_cuda and xyz.is_contiguous() and not (torch.is_grad_enabled() and xyz.requires_grad):
                idx, new_xyz = pointnet2_utils.sample_and_gather(xyz, self.npoint, index)  # one kernel, same values
            else:
                idx = pointnet2_utils.furthest_point_sample(xyz, self.npoint, index)
                channels_first = xyz.transpose(1, 2).contiguous()
                new_xyz = pointnet2_utils.gather_operation(channels_first, idx).transpose(1, 2).contiguous()
                if not new_xyz.requires_grad:
                    new_xyz = pointnet2_utils._own(new_xyz)

        # the ball queries of all scales share one walk over the index (nested balls around the same centres)
        idxs = None
        if index is not None and len(self.groupers) > 1 and all(isinstance(g, pointnet2_utils.QueryAndGroup) for g in self.groupers):
            idxs = pointnet2_utils.ball_query_multi([g.radius for g in self.groupers], [g.nsample for g in self.groupers],
                                                    xyz, new_xyz.contiguous(), index)
        # first 1x1 convolution folded into the grouping (pointnet2_utils.group_linear): W . [dxyz ; F[:, idx]] =
        # W_xyz . dxyz + (W_f . F)[:, idx] -- the dense product runs over the N points instead of npoint * nsample columns
        # and the (3 + C, npoint, nsample) grouped tensor is never built
        coords_need_grad = torch.is_grad_enabled() and (xyz.requires_grad or (new_xyz is not None and new_xyz.requires_grad))
        foldable = [self.fold_first_layer and not coords_need_grad and _foldable(g, m, features)
                    for g, m in zip(self.groupers, self.mlps)]
        groups = None
        if (idxs is not None and len({g.use_xyz for g in self.groupers}) == 1 and features is not None and not any(foldable)
                and not coords_need_grad):
            groups = pointnet2_utils.group_concat_multi(xyz, new_xyz, features, idxs, self.groupers[0].use_xyz)
        pooled = []
        for k, (grouper, mlp) in enumerate(zip(self.groupers, self.mlps)):
            if foldable[k]:
                nidx = idxs[k] if idxs is not None else pointnet2_utils.ball_query(grouper.radius, grouper.nsample, xyz,
                                                                                 new_xyz.contiguous(), index)
                conv = mlp[0].conv
                w = conv.weight[:, :, 0, 0]
                z = torch.matmul(w[:, 3:], features)            # (B, C_out, N)
                hidden = pointnet2_utils.group_linear(xyz, new_xyz, z, nidx, w[:, :3], conv.bias)
                for name, layer in mlp[0].named_children():     # what follows the convolution inside the first unit
                    if layer is not conv:
                        hidden = layer(hidden)
                for layer in list(mlp.children())[1:]:
                    hidden = layer(hidden)
                pooled.append(self._pool(hidden).squeeze(-1))
                continue
            if groups is not None:
                grouped = groups[k]
            elif idxs is not None:
                grouped = grouper(xyz, new_xyz, features, index, idxs[k])
            else:
                grouped = grouper(xyz, new_xyz, features, index) if index is not None else grouper(xyz, new_xyz, features)
            pooled.append(self._pool(mlp(grouped)).squeeze(-1))  # (B, mlp[-1], npoint)
        return new_xyz, torch.cat(pooled, dim=1), idx


class PointnetSAModuleMSG(_PointnetSAModuleBase):
    """multi-scale grouping SA layer"""

    def __init__(self, *, npoint: int, radii: List[float], nsamples: List[int], mlps: List[List[int]],
                 bn: bool = True, use_xyz: bool = True, pool_method='max_pool', instance_norm=False):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint = npoint
        self.pool_method = pool_method
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, spec in zip(radii, nsamples, mlps):
            self.groupers.append(pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz)
                                 if npoint is not None else pointnet2_utils.GroupAll(use_xyz))
            if use_xyz:
                spec[0] += 3  # in place, like the reference (:105-106): callers may rely on the mutated spec
            self.mlps.append(pt_utils.SharedMLP(spec, bn=bn, instance_norm=instance_norm))


class PointnetSAModule(PointnetSAModuleMSG):
    """single-scale SA layer (npoint=None -> GroupAll)"""

    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None, nsample: int = None,
                 bn: bool = True, use_xyz: bool = True, pool_method='max_pool', instance_norm=False):
        super().__init__(mlps=[mlp], npoint=npoint, radii=[radius], nsamples=[nsample], bn=bn, use_xyz=use_xyz,
                         pool_method=pool_method, instance_norm=instance_norm)


class PointnetFPModule(nn.Module):
    """propagates features of the `known` set onto the `unknown` set (inverse-distance weights
    over the 3 nearest neighbours), then a shared MLP"""

    def __init__(self, *, mlp: List[int], bn: bool = True, activation=nn.ReLU(inplace=True)):
        super().__init__()
        self.mlp = pt_utils.SharedMLP(mlp, bn=bn, activation=activation)

    def forward(self, unknown: torch.Tensor, known: torch.Tensor, unknow_feats: torch.Tensor,
                known_feats: torch.Tensor) -> torch.Tensor:
        """unknown (B,n,3), known (B,m,3), unknow_feats (B,C1,n), known_feats (B,C2,m) -> (B,mlp[-1],n)"""
        if known is None:
            spread = known_feats.expand(*known_feats.size()[0:2], unknown.size(1))
        else:
            # the centres an SA level produced carry the index the next level built of them (pointnet2_utils.scene_index);
            # a caller's own tensor (the input cloud) is indexed afresh -- nothing remembered could be trusted for it
            known_index = pointnet2_utils.scene_index(known)
            unknown_index = pointnet2_utils.scene_index(unknown) if known_index is not None else None
            dist, idx = pointnet2_utils.three_nn(unknown, known, unknown_index, known_index)
            inv = 1.0 / (dist + 1e-8)                               # reference :157-159
            weight = inv / torch.sum(inv, dim=2, keepdim=True)
            spread = pointnet2_utils.three_interpolate(known_feats, idx, weight)
        feats = spread if unknow_feats is None else torch.cat([spread, unknow_feats], dim=1)
        return self.mlp(feats.unsqueeze(-1)).squeeze(-1)
