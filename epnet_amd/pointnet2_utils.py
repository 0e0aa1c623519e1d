"""PointNet++ operator surface backed by the gfx950 kernels.

Re-provides the public names of the reference's pointnet2_lib/pointnet2/pointnet2_utils.py
(``furthest_point_sample``:36, ``gather_operation``:73, ``three_nn``:105, ``three_interpolate``:153,
``grouping_operation``:197, ``ball_query``:228, ``QueryAndGroup``:231, ``GroupAll``:267) with the
same call signatures, dtypes and output shapes. Differences from the reference are limited to
plumbing: outputs are allocated on the INPUT's device (the reference uses
``torch.cuda.FloatTensor(...)``, i.e. whatever device is current), and errors raise instead of
killing the process.
"""
import os
import threading
from typing import Optional, Tuple

import torch
import torch.nn as nn
from torch.autograd import Function

from . import pointnet2_cuda as _ext


def _new(like: torch.Tensor, shape, dtype=torch.float32, zero: bool = False) -> torch.Tensor:
    alloc = torch.zeros if zero else torch.empty
    return alloc(tuple(shape), dtype=dtype, device=like.device)


class FurthestPointSampling(Function):
    """(B,N,3) xyz -> (B,npoint) int32 indices; index 0 is always selected first.
    reference: pointnet2_utils.py:10-33"""

    @staticmethod
    def forward(ctx, xyz: torch.Tensor, npoint: int, index: Optional[torch.Tensor] = None) -> torch.Tensor:
        """index: optional scene_index(xyz) shared with the ball queries of the level (same result either way)"""
        assert xyz.is_contiguous()
        batch, n = xyz.shape[0], xyz.shape[1]
        out = _new(xyz, (batch, npoint), torch.int32)
        running_min = torch.full((batch, n), 1e10, dtype=torch.float32, device=xyz.device)
        if index is None:
            _ext.furthest_point_sampling_wrapper(batch, n, npoint, xyz, running_min, out)
        else:
            _ext.furthest_point_sampling_indexed_wrapper(batch, n, npoint, xyz, index, running_min, out)
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, grad=None):
        return None, None, None


furthest_point_sample = FurthestPointSampling.apply


_CAPTURE = {"epoch": 0, "was_capturing": False}
_CAPTURE_LOCK = threading.Lock()   # nn.DataParallel runs replica forward passes in threads


def _capture_epoch() -> int:
    """0 outside HIP-graph capture, a fresh positive number for every capture. An index built inside a capture lives in
    the graph's memory pool and is only meaningful to launches recorded into the same graph; one built eagerly must not be
    baked into a graph."""
    capturing = torch.cuda.is_current_stream_capturing() if torch.cuda.is_available() else False
    with _CAPTURE_LOCK:
        if capturing and not _CAPTURE["was_capturing"]:
            _CAPTURE["epoch"] += 1
        _CAPTURE["was_capturing"] = capturing
        return _CAPTURE["epoch"] if capturing else 0


def _own(t: torch.Tensor) -> torch.Tensor:
    """marks a tensor this package has just allocated and filled itself (the centres of an SA level). Only such tensors
    may carry a remembered scene index: nobody else holds a reference to them yet, every later write through torch or
    through this package's stand-ins moves their version counter, and inside a HIP graph they are rewritten together with
    their index by the same replay. A CALLER's tensor never carries one -- a graph replay, a ``.data`` write or a foreign
    extension writing through ``data_ptr()`` changes its content without any trace, and a remembered index would then
    silently answer for the old coordinates."""
    t._epnet_owned = True
    return t


def scene_index(xyz: torch.Tensor, cached_only: bool = False) -> Optional[torch.Tensor]:
    """one spatial sort of the (B,N,3) points of an SA level for furthest_point_sample / ball_query /
    QueryAndGroup / three_nn (their optional trailing arguments); None where the library indexes nothing.
    Beyond the reference: its kernels scan all N points each.

    The index travels as an attribute of the tensor object it was built from, and only on tensors this package
    allocated itself (``_own``): the centres an SA level hands to the next level and to its FP module. It is served again
    only while the tensor's version counter and the capture epoch are the ones it was built under. For any other tensor
    every call builds a fresh index (``cached_only``: look up, do not build)."""
    if not xyz.is_cuda or not xyz.is_contiguous():
        return None
    epoch = _capture_epoch()
    version = None if xyz.is_inference() else xyz._version     # inference tensors track no version: nothing is remembered
    hit = getattr(xyz, "_epnet_index", None)
    if hit is not None and version is not None and hit[1] == version and hit[2] == epoch:
        return hit[0]
    if cached_only:
        return None
    index = _ext.scene_index(xyz.detach())
    if index is not None and version is not None and getattr(xyz, "_epnet_owned", False):
        xyz._epnet_index = (index, version, epoch)
    return index


def _chain_by_attribute() -> bool:
    """EPNET_SA_CHAIN=0 switches the attribute-carried knowledge off (every level runs its rounds unless the caller passes
    `prefix=` explicitly, as sample_pyramid does)"""
    return os.environ.get("EPNET_SA_CHAIN", "1") != "0"


def sample_and_gather(xyz: torch.Tensor, npoint: int, index: Optional[torch.Tensor] = None, next_npoint: int = 0,
                      prefix: Optional[torch.Tensor] = None, with_prefix: bool = False):
    """the head of an SA module in one call: ``idx = furthest_point_sample(xyz, npoint)`` and
    ``new_xyz = gather_operation(xyz.transpose(1, 2), idx).transpose(1, 2)`` (pointnet2_modules.py:39-45) -- the
    sampling kernel has every selected point in registers, so the centres come with the indices. Same values;
    no gradient flows to xyz here (use the composition when the coordinates need one).

    Sampling pyramids: the centres returned here remember (as an attribute, like a scene index: only on this package's own
    tensors, only while their version counter stands) for how many leading rounds the sampling was unambiguous. When such
    centres are sampled again -- the next SA level -- scenes whose first npoint rounds were unambiguous get
    ``idx = 0 .. npoint-1``: furthest point sampling is nested (include/epnet_ops.h, epnet_sample_centres_chain), so that IS
    what the rounds would compute. Three of the four sampling kernels of the RPN pyramid disappear that way.

    ``prefix`` (int32 (B,), the prefix_out of the sampling that produced ``xyz``) hands that knowledge over EXPLICITLY and
    ``with_prefix=True`` returns this call's own as a third result: sample_pyramid chains its levels that way, nothing is read
    from or attached to a tensor. The attribute form serves callers that chain SA modules one by one; its hazard: the centres
    are handed back to the caller, and a write that does not move the version counter (``new_xyz.data.copy_()``, ``set_()``, a
    foreign extension writing through ``data_ptr()``) leaves the attribute standing on changed coordinates -- such callers set
    EPNET_SA_CHAIN=0."""
    assert xyz.is_contiguous()
    batch, n = xyz.shape[0], xyz.shape[1]
    idx = _new(xyz, (batch, npoint), torch.int32)
    new_xyz = _new(xyz, (batch, npoint, 3))
    prefix_in = prefix
    by_attribute = prefix is None and not with_prefix and _chain_by_attribute()
    known = getattr(xyz, "_epnet_fps_prefix", None) if by_attribute else None
    if known is not None and getattr(xyz, "_epnet_owned", False) and not xyz.is_inference() and known[1] == xyz._version \
            and known[2] == _capture_epoch():
        prefix_in = known[0]
    prefix_out = _new(xyz, (batch,), torch.int32)
    _ext.sample_centres_wrapper(batch, n, npoint, xyz.detach(), index, idx, new_xyz, prefix_in, prefix_out, int(next_npoint))
    new_xyz = _own(new_xyz)
    if by_attribute and not new_xyz.is_inference():
        new_xyz._epnet_fps_prefix = (prefix_out, new_xyz._version, _capture_epoch())
    return (idx, new_xyz, prefix_out) if with_prefix else (idx, new_xyz)


_PYRAMID_STREAMS = {}


def sample_pyramid(xyz: torch.Tensor, npoints):
    """the sampling of ALL set-abstraction levels up front: furthest point sampling depends on the
    coordinates only (pointnet2_modules.py:39-45), never on features, so the chain xyz -> npoints[0] -> npoints[1] -> ...
    (scene index, FPS, centres of every level) is issued on a side stream and runs beside the shared MLPs of the levels
    above it instead of between them. Returns one ``(idx, new_xyz, event, index)`` per level (index: the scene index of the level's
    input, or None) for
    ``_PointnetSAModuleBase.forward(..., presampled=...)``, which waits for the event; same values as the modules
    compute themselves."""
    assert xyz.is_cuda and xyz.is_contiguous()
    dev = xyz.device
    main = torch.cuda.current_stream(dev)
    side = _PYRAMID_STREAMS.get(dev.index)
    if side is None:
        side = _PYRAMID_STREAMS[dev.index] = torch.cuda.Stream(device=dev)
    side.wait_stream(main)
    levels = []
    with torch.cuda.stream(side):
        cur = xyz.detach()
        known = None                               # the chain of tie-free round counts, handed from level to level right here
        for k, m in enumerate(npoints):
            src = xyz if k == 0 else cur          # the tensor object the SA module of this level will receive
            index = scene_index(src)
            idx, new_xyz, known = sample_and_gather(src, int(m), index, int(npoints[k + 1]) if k + 1 < len(npoints) else 1,
                                                    prefix=known, with_prefix=True)
            event = torch.cuda.Event()
            event.record(side)
            for t in (idx, new_xyz, index):       # allocated on the side stream, consumed on the caller's
                if t is not None:
                    t.record_stream(main)
            levels.append((idx, new_xyz, event, index))
            cur = new_xyz
    return levels


class GatherOperation(Function):
    """features (B,C,N), idx (B,npoint) int32 -> (B,C,npoint). reference: pointnet2_utils.py:39-70"""

    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        assert features.is_contiguous() and idx.is_contiguous()
        batch, npoint = idx.shape
        channels, n = features.shape[1], features.shape[2]
        out = _new(features, (batch, channels, npoint))
        _ext.gather_points_wrapper(batch, channels, n, npoint, features, idx, out)
        ctx.for_backwards = (idx, channels, n)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, channels, n = ctx.for_backwards
        batch, npoint = idx.shape
        grad_features = _new(grad_out, (batch, channels, n), zero=True)
        _ext.gather_points_grad_wrapper(batch, channels, n, npoint, grad_out.detach().contiguous(), idx, grad_features)
        return grad_features, None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    """unknown (B,n,3), known (B,m,3) -> (dist (B,n,3) L2 distances, idx (B,n,3) int32).
    reference: pointnet2_utils.py:76-102 (the extension returns squared distances; sqrt here)"""

    @staticmethod
    def forward(ctx, unknown: torch.Tensor, known: torch.Tensor, unknown_index: Optional[torch.Tensor] = None,
                known_index: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """*_index: optional scene_index() of the two point sets (same result either way)"""
        assert unknown.is_contiguous() and known.is_contiguous()
        batch, n = unknown.shape[0], unknown.shape[1]
        m = known.shape[1]
        dist2 = _new(unknown, (batch, n, 3))
        idx = _new(unknown, (batch, n, 3), torch.int32)
        if known_index is None:
            _ext.three_nn_wrapper(batch, n, m, unknown, known, dist2, idx)
        else:
            _ext.three_nn_indexed_wrapper(batch, n, m, unknown, known, unknown_index, known_index, dist2, idx)
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    """features (B,C,m), idx/weight (B,n,3) -> (B,C,n). reference: pointnet2_utils.py:108-150"""

    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
        assert features.is_contiguous() and idx.is_contiguous() and weight.is_contiguous()
        batch, channels, m = features.shape
        n = idx.shape[1]
        ctx.three_interpolate_for_backward = (idx, weight, m)
        out = _new(features, (batch, channels, n))
        _ext.three_interpolate_wrapper(batch, channels, m, n, features, idx, weight, out)
        return out

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        idx, weight, m = ctx.three_interpolate_for_backward
        batch, channels, n = grad_out.shape
        grad_features = _new(grad_out, (batch, channels, m), zero=True)
        _ext.three_interpolate_grad_wrapper(batch, channels, n, m, grad_out.detach().contiguous(), idx, weight,
                                            grad_features)
        return grad_features, None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    """features (B,C,N), idx (B,npoint,nsample) int32 -> (B,C,npoint,nsample).
    reference: pointnet2_utils.py:156-194"""

    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        assert features.is_contiguous() and idx.is_contiguous()
        batch, npoint, nsample = idx.shape
        channels, n = features.shape[1], features.shape[2]
        out = _new(features, (batch, channels, npoint, nsample))
        _ext.group_points_wrapper(batch, channels, n, npoint, nsample, features, idx, out)
        ctx.for_backwards = (idx, n)
        return out

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        idx, n = ctx.for_backwards
        batch, channels, npoint, nsample = grad_out.shape
        grad_features = _new(grad_out, (batch, channels, n), zero=True)
        _ext.group_points_grad_wrapper(batch, channels, n, npoint, nsample, grad_out.detach().contiguous(), idx,
                                       grad_features)
        return grad_features, None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    """radius, nsample, xyz (B,N,3), new_xyz (B,npoint,3) -> idx (B,npoint,nsample) int32: the first
    nsample points (index order) strictly inside the ball, padded with the first hit; zeros for an
    empty ball. reference: pointnet2_utils.py:200-225"""

    @staticmethod
    def forward(ctx, radius: float, nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor,
                index: Optional[torch.Tensor] = None) -> torch.Tensor:
        assert new_xyz.is_contiguous() and xyz.is_contiguous()
        batch, n = xyz.shape[0], xyz.shape[1]
        npoint = new_xyz.shape[1]
        idx = _new(xyz, (batch, npoint, nsample), torch.int32, zero=True)
        if index is None:
            _ext.ball_query_wrapper(batch, n, npoint, radius, nsample, new_xyz, xyz, idx)
        else:
            _ext.ball_query_indexed_wrapper(batch, n, npoint, radius, nsample, new_xyz, xyz, index, idx)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None, None


ball_query = BallQuery.apply


def ball_query_multi(radii, nsamples, xyz: torch.Tensor, new_xyz: torch.Tensor, index: Optional[torch.Tensor] = None):
    """the ball queries of all scales of an MSG level (same centres, same points) in one launch: a list of
    (B,npoint,nsample_k) int32 tensors, each equal to ``ball_query(radii[k], nsamples[k], xyz, new_xyz)``"""
    assert new_xyz.is_contiguous() and xyz.is_contiguous()
    batch, n = xyz.shape[0], xyz.shape[1]
    npoint = new_xyz.shape[1]
    if index is None:
        index = scene_index(xyz)
    if index is None:
        return [ball_query(r, ns, xyz, new_xyz) for r, ns in zip(radii, nsamples)]
    idxs = [_new(xyz, (batch, npoint, ns), torch.int32, zero=True) for ns in nsamples]
    _ext.ball_query_multi_wrapper(batch, n, npoint, list(radii), list(nsamples), new_xyz.detach(), xyz.detach(), index, idxs)
    return idxs


class _GroupConcat(Function):
    """fused tail of QueryAndGroup: [grouped xyz - centre ; grouped features] written once into one tensor
    (the reference composes two grouping ops, an in-place subtraction and a torch.cat copy)"""

    @staticmethod
    def forward(ctx, xyz, new_xyz, features, idx, use_xyz):
        batch, npoint, nsample = idx.shape
        n = xyz.shape[1]
        channels = 0 if features is None else features.shape[1]
        out = _new(xyz, (batch, (3 if use_xyz else 0) + channels, npoint, nsample))
        _ext.group_concat_wrapper(batch, channels, n, npoint, nsample, xyz, new_xyz, features, idx, out, use_xyz)
        ctx.for_backwards = (idx, channels, n, use_xyz)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, channels, n, use_xyz = ctx.for_backwards
        if channels == 0:
            return None, None, None, None, None
        batch, _, npoint, nsample = grad_out.shape
        grad_features = _new(grad_out, (batch, channels, n), zero=True)
        _ext.group_concat_grad_wrapper(batch, channels, n, npoint, nsample, grad_out.detach().contiguous(), idx,
                                       grad_features, use_xyz)
        return None, None, grad_features, None, None


class _GroupConcatMulti(Function):
    """_GroupConcat for all scales of an MSG level in one call (same tensors; with two scales the feature rows are
    staged on the chip once for both)"""

    @staticmethod
    def forward(ctx, xyz, new_xyz, features, use_xyz, *idxs):
        batch, npoint = idxs[0].shape[0], idxs[0].shape[1]
        n = xyz.shape[1]
        channels = 0 if features is None else features.shape[1]
        nsamples = [int(i.shape[2]) for i in idxs]
        outs = [_new(xyz, (batch, (3 if use_xyz else 0) + channels, npoint, ns)) for ns in nsamples]
        _ext.group_concat_multi_wrapper(batch, channels, n, npoint, nsamples, xyz, new_xyz, features, list(idxs), outs, use_xyz)
        ctx.for_backwards = (idxs, channels, n, use_xyz)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grad_outs):
        idxs, channels, n, use_xyz = ctx.for_backwards
        none = (None,) * (4 + len(idxs))
        if channels == 0:
            return none
        batch = grad_outs[0].shape[0]
        grad_features = _new(grad_outs[0], (batch, channels, n), zero=True)
        for idx, g in zip(idxs, grad_outs):   # the gradient op accumulates into its output, like the reference's atomicAdd
            npoint, nsample = idx.shape[1], idx.shape[2]
            _ext.group_concat_grad_wrapper(batch, channels, n, npoint, nsample, g.detach().contiguous(), idx, grad_features, use_xyz)
        return (None, None, grad_features, None) + (None,) * len(idxs)


def group_concat_multi(xyz, new_xyz, features, idxs, use_xyz=True):
    """[grouped xyz - centre ; grouped features] for every neighbour list in `idxs` (one per MSG scale), as
    QueryAndGroup produces them one by one"""
    feats = None if features is None else features.contiguous()
    return list(_GroupConcatMulti.apply(xyz.contiguous(), new_xyz.contiguous(), feats, use_xyz or features is None, *idxs))


class _GroupLinear(Function):
    """pre-activations of an SA level's first 1x1 convolution without the grouped tensor:
    out = z[:, :, idx] + w_xyz . (xyz[idx] - new_xyz) (+ bias), z = W_f . features computed by the caller on the N points.
    Equals conv(cat(grouped xyz - centre, grouped features)) of the reference (pointnet2_utils.py:250-257 +
    pointnet2_modules.py:61) up to the summation order of the dense product."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, z, idx, w_xyz, bias):
        b, c, n = z.shape
        npoint, nsample = idx.shape[1], idx.shape[2]
        out = _new(z, (b, c, npoint, nsample))
        _ext.group_linear_wrapper(b, c, n, npoint, nsample, xyz, new_xyz, z, idx, w_xyz, bias, out)
        ctx.save_for_backward(xyz, new_xyz, idx)
        ctx.dims = (b, c, n, npoint, nsample, bias is not None)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        xyz, new_xyz, idx = ctx.saved_tensors
        b, c, n, npoint, nsample, has_bias = ctx.dims
        g = grad_out.detach().contiguous()
        grad_z = grad_w = grad_b = None
        if ctx.needs_input_grad[2]:      # scatter-add over the neighbour lists, as for any grouped tensor
            grad_z = _new(g, (b, c, n), zero=True)
            _ext.group_points_grad_wrapper(b, c, n, npoint, nsample, g, idx, grad_z)
        if ctx.needs_input_grad[4]:      # d out / d w_xyz = the centred neighbour coordinates, rebuilt inside the reduction
            grad_w = _new(g, (c, 3), zero=True)
            _ext.group_linear_grad_w_wrapper(b, c, n, npoint, nsample, g, xyz, new_xyz, idx, grad_w)
        if has_bias and ctx.needs_input_grad[5]:
            grad_b = g.sum(dim=(0, 2, 3))
        return None, None, grad_z, None, grad_w, grad_b


def group_linear(xyz, new_xyz, z, idx, w_xyz, bias=None):
    """(B,N,3), (B,M,3), z (B,C,N), idx (B,M,ns) int32, w_xyz (C,3), bias (C) -> (B,C,M,ns)"""
    return _GroupLinear.apply(xyz.contiguous(), new_xyz.contiguous(), z.contiguous(), idx, w_xyz.contiguous(),
                              None if bias is None else bias.contiguous())


class _PoolMax(Function):
    """max over the last axis of a contiguous (..., nsample) tensor, keepdim -- the values of
    F.max_pool2d(x, kernel_size=[1, nsample]) (reference pointnet2_modules.py:61-68) from a kernel that reads the rows with
    coalesced 16-byte loads; the backward routes every gradient to the recorded position of its maximum"""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        nsample = x.shape[-1]
        rows = x.numel() // max(nsample, 1)
        out = _new(x, x.shape[:-1] + (1,))
        need_grad = ctx.needs_input_grad[0]
        arg = _new(x, x.shape[:-1], dtype=torch.int32) if need_grad else None
        _ext.pool_max_wrapper(rows, nsample, x, out, arg)
        ctx.for_backwards = (arg, tuple(x.shape))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        arg, shape = ctx.for_backwards
        grad_x = _new(grad_out, shape)
        _ext.pool_max_grad_wrapper(grad_x.numel() // shape[-1], shape[-1], grad_out.detach().contiguous(), arg, grad_x)
        return grad_x


def pool_max(x):
    """(B, C, npoint, nsample) -> (B, C, npoint, 1): the neighbourhood max-pool of an SA level"""
    return _PoolMax.apply(x)


class QueryAndGroup(nn.Module):
    """ball_query -> group xyz -> subtract the centre -> group features -> concat [xyz(3), features(C)].
    reference: pointnet2_utils.py:231-264. Same values as that composition; the grouped tensor is produced
    by one fused op unless a gradient w.r.t. the coordinates is required."""

    def __init__(self, radius: float, nsample: int, use_xyz: bool = True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: Optional[torch.Tensor] = None,
                index: Optional[torch.Tensor] = None, idx: Optional[torch.Tensor] = None):
        """index: optional scene_index(xyz); idx: optional precomputed ball_query result (ball_query_multi)"""
        if idx is None:
            idx = ball_query(self.radius, self.nsample, xyz, new_xyz, index)
        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
        coords_need_grad = torch.is_grad_enabled() and (xyz.requires_grad or new_xyz.requires_grad)
        if not coords_need_grad:
            feats = None if features is None else features.contiguous()
            return _GroupConcat.apply(xyz.contiguous(), new_xyz.contiguous(), feats, idx,
                                      self.use_xyz or features is None)
        # unfused composition, differentiable w.r.t. the coordinates exactly like the reference
        local_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)  # (B,3,npoint,nsample)
        local_xyz -= new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is None:
            return local_xyz
        grouped = grouping_operation(features, idx)
        return torch.cat([local_xyz, grouped], dim=1) if self.use_xyz else grouped


class GroupAll(nn.Module):
    """one group holding every point: (B, 3+C, 1, N). reference: pointnet2_utils.py:267-290"""

    def __init__(self, use_xyz: bool = True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: Optional[torch.Tensor] = None):
        all_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return all_xyz
        all_feat = features.unsqueeze(2)
        return torch.cat([all_xyz, all_feat], dim=1) if self.use_xyz else all_feat
