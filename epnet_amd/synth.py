"""Synthetic KITTI-shaped inputs. No dataset is needed.

All generators are deterministic functions of an integer seed (``torch.Generator`` on the CPU) and
return contiguous fp32 CPU tensors; callers move them to the GPU.
"""
import math

import torch

# PC_AREA_SCOPE, reference lib/config.py:26-28: x in [-40,40], y in [-1,3], z in [0,70.4]
SCOPE = ((-40.0, 40.0), (-1.0, 3.0), (0.0, 70.4))
CLS_MEAN_SIZE = (1.52563191462, 1.62856739989, 3.88311640418)  # (h, w, l), reference yaml:19


def _gen(seed):
    return torch.Generator().manual_seed(int(seed))


def ubox_cloud(n, seed=0):
    """uniform in the detection scope: the worst case for ball query (almost no early exits)"""
    g = _gen(seed)
    u = torch.rand((n, 3), generator=g, dtype=torch.float32)
    lo = torch.tensor([s[0] for s in SCOPE], dtype=torch.float32)
    hi = torch.tensor([s[1] for s in SCOPE], dtype=torch.float32)
    return (lo + u * (hi - lo)).contiguous()


def object_boxes(num, seed=0):
    """(num,7) [x, y(bottom), z, h, w, l, ry] car-sized boxes on the ground plane"""
    g = _gen(seed + 7919)
    r = torch.rand((num, 6), generator=g, dtype=torch.float32)
    rho = 5.0 + r[:, 0] * 55.0
    az = (r[:, 1] - 0.5) * math.radians(70.0)
    x, z = rho * torch.sin(az), rho * torch.cos(az)
    size = torch.tensor(CLS_MEAN_SIZE, dtype=torch.float32) * (0.8 + 0.4 * r[:, 2:5])
    ry = (r[:, 5] * 2 - 1) * math.pi
    y = torch.full((num,), 1.65, dtype=torch.float32)
    return torch.cat([x[:, None], y[:, None], z[:, None], size, ry[:, None]], dim=1).contiguous()


def kitti_like_cloud(n, seed=0, num_objects=40, return_boxes=False):
    """85 % ground returns (y = 1.65 + N(0, 0.03), range pdf ~ 1/rho on [2,70] m, azimuth +-40 deg) and
    15 % points on the surfaces of `num_objects` car-sized boxes; clipped to the scope; shuffled"""
    g = _gen(seed)
    n_obj = int(round(n * 0.15))
    n_gnd = n - n_obj
    u = torch.rand((n_gnd, 2), generator=g, dtype=torch.float32)
    rho = 2.0 * (35.0 ** u[:, 0])  # log-uniform on [2,70]  <=> pdf ~ 1/rho
    az = (u[:, 1] - 0.5) * math.radians(80.0)
    gy = 1.65 + 0.03 * torch.randn((n_gnd,), generator=g, dtype=torch.float32)
    ground = torch.stack([rho * torch.sin(az), gy, rho * torch.cos(az)], dim=1)

    boxes = object_boxes(num_objects, seed)
    which = torch.randint(0, num_objects, (n_obj,), generator=g)
    b = boxes[which]
    uvw = torch.rand((n_obj, 3), generator=g, dtype=torch.float32) - 0.5  # box frame, in [-.5,.5]
    face = torch.randint(0, 3, (n_obj,), generator=g)
    sign = torch.randint(0, 2, (n_obj,), generator=g).float() - 0.5
    uvw[torch.arange(n_obj), face] = sign  # snap one coordinate to a face
    lx, hy, wz = uvw[:, 0] * b[:, 5], uvw[:, 1] * b[:, 3], uvw[:, 2] * b[:, 4]
    c, s = torch.cos(b[:, 6]), torch.sin(b[:, 6])
    ox = b[:, 0] + lx * c + wz * s
    oz = b[:, 2] - lx * s + wz * c
    oy = b[:, 1] - b[:, 3] / 2 + hy
    obj = torch.stack([ox, oy, oz], dim=1)

    pts = torch.cat([ground, obj], dim=0)
    for d in range(3):
        pts[:, d].clamp_(SCOPE[d][0], SCOPE[d][1])
    pts = pts[torch.randperm(n, generator=g)].contiguous()
    return (pts, boxes) if return_boxes else pts


def dup_cloud(n, seed=0, unique=12000):
    """`unique` KITTI-like points padded to n by re-drawing existing rows, as the dataset pads short
    clouds (reference lib/datasets/kitti_rcnn_dataset.py:338-342): exercises FPS tie-breaks"""
    base = kitti_like_cloud(min(unique, n), seed)
    if n <= base.shape[0]:
        return base[:n].contiguous()
    g = _gen(seed + 104729)
    extra = torch.randint(0, base.shape[0], (n - base.shape[0],), generator=g)
    return torch.cat([base, base[extra]], dim=0).contiguous()


def kitti_q_cloud(n, seed=0):
    """KITTI-like coordinates rounded to 1e-3 m, the resolution velodyne .bin files carry: decimal-quantised coordinates
    make equal fp32 distances (FPS ties) far likelier than the continuous generator does"""
    return (torch.round(kitti_like_cloud(n, seed) * 1000.0) / 1000.0).contiguous()


KINDS = {"ubox": ubox_cloud, "kitti": kitti_like_cloud, "dup": dup_cloud, "kitti_q": kitti_q_cloud}


def cloud(kind, n, seed=0):
    return KINDS[kind](n, seed)


def scenes(kind, batch, n, seed=0):
    """(batch, n, 3) stack of independent scenes; scene i uses seed + i"""
    return torch.stack([KINDS[kind](n, seed + i) for i in range(batch)], dim=0).contiguous()


def proposal_boxes(num, seed=0, num_objects=40, jitter=1.0):
    """(num,7) boxes scattered around the scene's objects (for iou3d / roipool3d / NMS inputs) + scores"""
    g = _gen(seed + 15485863)
    obj = object_boxes(num_objects, seed)
    pick = obj[torch.randint(0, num_objects, (num,), generator=g)].clone()
    pick[:, [0, 2]] += (torch.rand((num, 2), generator=g) * 2 - 1) * jitter
    pick[:, 3:6] *= 0.8 + 0.4 * torch.rand((num, 3), generator=g)
    pick[:, 6] += (torch.rand((num,), generator=g) * 2 - 1) * 0.3
    scores = torch.rand((num,), generator=g, dtype=torch.float32)
    return pick.contiguous(), scores
