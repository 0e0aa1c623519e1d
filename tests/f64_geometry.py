"""Float64-capable stand-ins for the geometry ops of epnet_amd.pointnet2_utils, for a float64 run of a model built on them
(tests/test_two_stream.py: the yardstick both point-to-pixel samplers are held to). The INTEGER results -- furthest-point
indices, ball-query indices, the three nearest neighbours -- come from the HIP kernels on the float32 coordinates (which are the
float64 run's coordinates exactly); everything that carries values is plain torch in the tensors' own dtype and differentiable
by autograd."""
import contextlib

import torch


def _gather_cols(features, idx):
    """features (B,C,N), idx (B,...) -> (B,C,...) = features[b, :, idx[b, ...]]"""
    b, c, _ = features.shape
    flat = idx.reshape(b, 1, -1).long().expand(b, c, -1)
    return torch.gather(features, 2, flat).reshape(b, c, *idx.shape[1:])


@contextlib.contextmanager
def float64_geometry():
    from epnet_amd import pointnet2_utils as p2u
    real = {name: getattr(p2u, name) for name in ("scene_index", "sample_and_gather", "furthest_point_sample", "gather_operation",
                                                  "ball_query", "grouping_operation", "three_nn", "three_interpolate", "_GroupConcat")}

    def sample_and_gather(xyz, npoint, index=None, next_npoint=0):
        idx = real["furthest_point_sample"](xyz.float().contiguous(), npoint)
        return idx, torch.gather(xyz, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()

    def furthest_point_sample(xyz, npoint, index=None):
        return real["furthest_point_sample"](xyz.float().contiguous(), npoint)

    def ball_query(radius, nsample, xyz, new_xyz, index=None):
        return real["ball_query"](radius, nsample, xyz.float().contiguous(), new_xyz.float().contiguous())

    def three_nn(unknown, known, unknown_index=None, known_index=None):
        _, idx = real["three_nn"](unknown.float().contiguous(), known.float().contiguous())
        picked = torch.gather(known.unsqueeze(1).expand(-1, unknown.shape[1], -1, -1), 2, idx.long().unsqueeze(-1).expand(-1, -1, -1, 3))
        return (unknown.unsqueeze(2) - picked).pow(2).sum(-1).sqrt(), idx        # interpolate_gpu.cu:30-48 + pointnet2_utils.py:121

    def three_interpolate(features, idx, weight):
        return (_gather_cols(features, idx) * weight.unsqueeze(1)).sum(-1)      # interpolate_gpu.cu:86-106

    class GroupConcat:
        @staticmethod
        def apply(xyz, new_xyz, features, idx, use_xyz):                         # pointnet2_utils.py:249-257
            local = _gather_cols(xyz.transpose(1, 2), idx) - new_xyz.transpose(1, 2).unsqueeze(-1)
            if features is None:
                return local
            grouped = _gather_cols(features, idx)
            return torch.cat([local, grouped], dim=1) if use_xyz else grouped

    fake = {"scene_index": lambda *a, **k: None, "sample_and_gather": sample_and_gather, "furthest_point_sample": furthest_point_sample,
            "gather_operation": _gather_cols, "ball_query": ball_query, "grouping_operation": _gather_cols, "three_nn": three_nn,
            "three_interpolate": three_interpolate, "_GroupConcat": GroupConcat}
    for name, fn in fake.items():
        setattr(p2u, name, fn)
    try:
        yield
    finally:
        for name, fn in real.items():
            setattr(p2u, name, fn)
