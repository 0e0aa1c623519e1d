"""Stand-in for the reference's ``pointnet2_cuda`` extension module.

Same nine function names, positional argument orders and pre-allocated-output convention as
pointnet2_lib/pointnet2/src/pointnet2_api.cpp:10-24 (note the quirks kept on purpose: ball_query
takes new_xyz before xyz; three_interpolate is (b,c,m,n) but its grad is (b,c,n,m)). Each call
validates its tensors, then hands raw device pointers and the tensor's current HIP stream to the C
ABI in libepnet_hip.so. Failures raise RuntimeError; the reference prints and exit()s.
"""
import torch

from . import _lib
from ._tensor import dev_ptr, need, on_device_of, writes

_F, _I = torch.float32, torch.int32


@writes("idx")
def ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
    """ball_query_wrapper_fast, pointnet2_lib/pointnet2/src/ball_query.cpp:14-25"""
    pn, px, pi = dev_ptr(new_xyz, "new_xyz", _F), dev_ptr(xyz, "xyz", _F), dev_ptr(idx, "idx", _I)
    need(new_xyz, b * m * 3, "new_xyz"); need(xyz, b * n * 3, "xyz"); need(idx, b * m * nsample, "idx")
    l = _lib.lib()
    ws_bytes = l.epnet_ball_query_workspace_bytes(b, n, m)
    with on_device_of(xyz) as s:
        if ws_bytes:  # scratch from torch's caching allocator lets the library index the scene spatially
            ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=xyz.device)
            _lib.check(l.epnet_ball_query_ws(b, n, m, radius, nsample, pn, px, pi, ws.data_ptr(), ws_bytes, s), "ball_query")
        else:
            _lib.check(l.epnet_ball_query(b, n, m, radius, nsample, pn, px, pi, s), "ball_query")
    return 1


@writes("out")
def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
    """group_points_wrapper_fast, group_points.cpp:25-36"""
    pp, pi, po = dev_ptr(points, "points", _F), dev_ptr(idx, "idx", _I), dev_ptr(out, "out", _F)
    need(points, b * c * n, "points"); need(idx, b * npoints * nsample, "idx"); need(out, b * c * npoints * nsample, "out")
    with on_device_of(points) as s:
        _lib.check(_lib.lib().epnet_group_points(b, c, n, npoints, nsample, pp, pi, po, s), "group_points")
    return 1


@writes("grad_points")
def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
    """group_points_grad_wrapper_fast, group_points.cpp:11-22"""
    pg, pi, pp = dev_ptr(grad_out, "grad_out", _F), dev_ptr(idx, "idx", _I), dev_ptr(grad_points, "grad_points", _F)
    need(grad_out, b * c * npoints * nsample, "grad_out"); need(idx, b * npoints * nsample, "idx")
    need(grad_points, b * c * n, "grad_points")
    l = _lib.lib()
    ws_bytes = l.epnet_group_points_grad_workspace_bytes(b, n, npoints, nsample)
    with on_device_of(grad_out) as s:
        if ws_bytes:
            ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=grad_out.device)
            _lib.check(l.epnet_group_points_grad_ws(b, c, n, npoints, nsample, pg, pi, pp, ws.data_ptr(), ws_bytes, s),
                       "group_points_grad")
        else:
            _lib.check(l.epnet_group_points_grad(b, c, n, npoints, nsample, pg, pi, pp, s), "group_points_grad")
    return 1


@writes("out")
def gather_points_wrapper(b, c, n, npoints, points, idx, out):
    """gather_points_wrapper_fast, sampling.cpp:11-20"""
    pp, pi, po = dev_ptr(points, "points", _F), dev_ptr(idx, "idx", _I), dev_ptr(out, "out", _F)
    need(points, b * c * n, "points"); need(idx, b * npoints, "idx"); need(out, b * c * npoints, "out")
    with on_device_of(points) as s:
        _lib.check(_lib.lib().epnet_gather_points(b, c, n, npoints, pp, pi, po, s), "gather_points")
    return 1


@writes("grad_points")
def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
    """gather_points_grad_wrapper_fast, sampling.cpp:23-33"""
    pg, pi, pp = dev_ptr(grad_out, "grad_out", _F), dev_ptr(idx, "idx", _I), dev_ptr(grad_points, "grad_points", _F)
    need(grad_out, b * c * npoints, "grad_out"); need(idx, b * npoints, "idx"); need(grad_points, b * c * n, "grad_points")
    with on_device_of(grad_out) as s:
        _lib.check(_lib.lib().epnet_gather_points_grad(b, c, n, npoints, pg, pi, pp, s), "gather_points_grad")
    return 1


@writes("temp", "idx")
def furthest_point_sampling_wrapper(b, n, m, points, temp, idx):
    """furthest_point_sampling_wrapper, sampling.cpp:36-46"""
    pp, pt, pi = dev_ptr(points, "points", _F), dev_ptr(temp, "temp", _F), dev_ptr(idx, "idx", _I)
    need(points, b * n * 3, "points"); need(temp, b * n, "temp"); need(idx, b * m, "idx")
    l = _lib.lib()
    with on_device_of(points) as s:
        if 16384 < n <= 65536:  # scenes beyond the register file: the kernel works over a scene index (scratch)
            nbytes = l.epnet_scene_index_bytes(b, n)
            index = torch.empty((nbytes,), dtype=torch.uint8, device=points.device)
            _lib.check(l.epnet_scene_index_build(b, n, pp, index.data_ptr(), nbytes, s), "scene_index_build")
            _lib.check(l.epnet_furthest_point_sampling_indexed(b, n, m, pp, index.data_ptr(), nbytes, pt, pi, s),
                       "furthest_point_sampling")
        else:
            _lib.check(l.epnet_furthest_point_sampling(b, n, m, pp, pt, pi, s), "furthest_point_sampling")
    return 1


@writes("dist2", "idx")
def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
    """three_nn_wrapper_fast, interpolate.cpp:14-23"""
    pu, pk = dev_ptr(unknown, "unknown", _F), dev_ptr(known, "known", _F)
    pd, pi = dev_ptr(dist2, "dist2", _F), dev_ptr(idx, "idx", _I)
    need(unknown, b * n * 3, "unknown"); need(known, b * m * 3, "known"); need(dist2, b * n * 3, "dist2"); need(idx, b * n * 3, "idx")
    l = _lib.lib()
    ws_bytes = l.epnet_three_nn_workspace_bytes(b, n, m)
    with on_device_of(unknown) as s:
        if ws_bytes:
            ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=unknown.device)
            _lib.check(l.epnet_three_nn_ws(b, n, m, pu, pk, pd, pi, ws.data_ptr(), ws_bytes, s), "three_nn")
        else:
            _lib.check(l.epnet_three_nn(b, n, m, pu, pk, pd, pi, s), "three_nn")


@writes("out")
def three_interpolate_wrapper(b, c, m, n, points, idx, weight, out):
    """three_interpolate_wrapper_fast, interpolate.cpp:26-38"""
    pp, pi = dev_ptr(points, "points", _F), dev_ptr(idx, "idx", _I)
    pw, po = dev_ptr(weight, "weight", _F), dev_ptr(out, "out", _F)
    need(points, b * c * m, "points"); need(idx, b * n * 3, "idx"); need(weight, b * n * 3, "weight"); need(out, b * c * n, "out")
    with on_device_of(points) as s:
        _lib.check(_lib.lib().epnet_three_interpolate(b, c, m, n, pp, pi, pw, po, s), "three_interpolate")


@writes("grad_points")
def three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points):
    """three_interpolate_grad_wrapper_fast, interpolate.cpp:41-54"""
    pg, pi = dev_ptr(grad_out, "grad_out", _F), dev_ptr(idx, "idx", _I)
    pw, pp = dev_ptr(weight, "weight", _F), dev_ptr(grad_points, "grad_points", _F)
    need(grad_out, b * c * n, "grad_out"); need(idx, b * n * 3, "idx"); need(weight, b * n * 3, "weight")
    need(grad_points, b * c * m, "grad_points")
    l = _lib.lib()
    ws_bytes = l.epnet_three_interpolate_grad_workspace_bytes(b, n, m)
    with on_device_of(grad_out) as s:
        if ws_bytes:
            ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=grad_out.device)
            _lib.check(l.epnet_three_interpolate_grad_ws(b, c, n, m, pg, pi, pw, pp, ws.data_ptr(), ws_bytes, s),
                       "three_interpolate_grad")
        else:
            _lib.check(l.epnet_three_interpolate_grad(b, c, n, m, pg, pi, pw, pp, s), "three_interpolate_grad")


# ---- beyond the reference's nine entry points: fused QueryAndGroup tail -----------------


def scene_index_bytes(b, n):
    """bytes of the scene index of b scenes of n points; 0 where the library indexes nothing"""
    return int(_lib.lib().epnet_scene_index_bytes(b, n))


def scene_index(xyz):
    """one spatial sort of (B,N,3) points, shared by the sampling and the ball queries of an SA level; None where
    the library indexes nothing (N < 1024 or N > 65536). include/epnet_ops.h: epnet_scene_index_build"""
    b, n = xyz.shape[0], xyz.shape[1]
    l = _lib.lib()
    nbytes = l.epnet_scene_index_bytes(b, n)
    if not nbytes:
        return None
    px = dev_ptr(xyz, "xyz", _F)
    need(xyz, b * n * 3, "xyz")
    index = torch.empty((nbytes,), dtype=torch.uint8, device=xyz.device)
    with on_device_of(xyz) as s:
        _lib.check(l.epnet_scene_index_build(b, n, px, index.data_ptr(), nbytes, s), "scene_index_build")
    return index


@writes("index")
def scene_index_build_wrapper(b, n, xyz, index):
    """rebuild into a caller-owned buffer of epnet_scene_index_bytes(b, n) bytes (graph-capturable)"""
    px = dev_ptr(xyz, "xyz", _F)
    need(xyz, b * n * 3, "xyz")
    with on_device_of(xyz) as s:
        _lib.check(_lib.lib().epnet_scene_index_build(b, n, px, index.data_ptr(), index.numel(), s), "scene_index_build")
    return 1


@writes("new_xyz", "index")
def scene_index_build_gathered_wrapper(b, n_src, n, xyz_src, idx, new_xyz, index):
    """new_xyz (b,n,3) = the rows idx (b,n) of xyz_src (b,n_src,3) -- gather_points_wrapper on the (B,N,3) layout -- and the scene
    index of those n points into `index`, one launch (include/epnet_ops.h: epnet_scene_index_build_gathered; 1024 <= n <= 16384)"""
    px, pi, pn = dev_ptr(xyz_src, "xyz_src", _F), dev_ptr(idx, "idx", _I), dev_ptr(new_xyz, "new_xyz", _F)
    need(xyz_src, b * n_src * 3, "xyz_src")
    need(idx, b * n, "idx")
    need(new_xyz, b * n * 3, "new_xyz")
    with on_device_of(xyz_src) as s:
        _lib.check(_lib.lib().epnet_scene_index_build_gathered(b, n_src, n, px, pi, pn, index.data_ptr(), index.numel(), s),
                   "scene_index_build_gathered")
    return 1


def _index_args(index, like):
    if index is None:
        return None, 0
    if index.dtype != torch.uint8 or not index.is_contiguous() or index.device != like.device:
        raise RuntimeError("scene index must be the contiguous uint8 tensor scene_index() returned, on the points' device")
    return index.data_ptr(), index.numel()


@writes("temp", "idx")
def furthest_point_sampling_indexed_wrapper(b, n, m, points, index, temp, idx):
    """furthest_point_sampling_wrapper over a scene index of `points` (same results)"""
    pp, pt, pi = dev_ptr(points, "points", _F), dev_ptr(temp, "temp", _F), dev_ptr(idx, "idx", _I)
    need(points, b * n * 3, "points"); need(temp, b * n, "temp"); need(idx, b * m, "idx")
    px, nb = _index_args(index, points)
    with on_device_of(points) as s:
        _lib.check(_lib.lib().epnet_furthest_point_sampling_indexed(b, n, m, pp, px, nb, pt, pi, s), "furthest_point_sampling")
    return 1


@writes("idx", "new_xyz", "prefix_out")
def sample_centres_wrapper(b, n, m, points, index, idx, new_xyz, prefix_in=None, prefix_out=None, prefix_cap=0):
    """furthest_point_sampling from a fresh state + gather of the selected rows: idx (B,M) and new_xyz (B,M,3).
    prefix_in / prefix_out (int32 (B,) or None): the sampling pyramid's chain of knowledge (include/epnet_ops.h,
    epnet_sample_centres_chain): scenes whose input is known to be an unambiguous furthest-point sequence of at least m
    samples get idx = 0 .. m-1 without running the rounds -- the same result; prefix_cap = the next level's sample count
    (ties are looked for in that many rounds only; 0 = all)"""
    pp, pi = dev_ptr(points, "points", _F), dev_ptr(idx, "idx", _I)
    pn = dev_ptr(new_xyz, "new_xyz", _F) if new_xyz is not None else None   # (None: indices only, chained form)
    need(points, b * n * 3, "points"); need(idx, b * m, "idx")
    if new_xyz is not None:
        need(new_xyz, b * m * 3, "new_xyz")
    elif prefix_in is None and prefix_out is None:
        raise RuntimeError("new_xyz may be None only with prefix_in / prefix_out (epnet_sample_centres_chain)")
    px, nb = _index_args(index, points)
    temp = None if 64 <= n <= 16384 else torch.full((b, n), 1e10, dtype=_F, device=points.device)
    pt = None if temp is None else temp.data_ptr()
    with on_device_of(points) as s:
        if prefix_in is None and prefix_out is None:
            _lib.check(_lib.lib().epnet_sample_centres(b, n, m, pp, px, nb, pt, pi, pn, s), "sample_centres")
        else:
            pin = dev_ptr(prefix_in, "prefix_in", _I) if prefix_in is not None else None
            pout = dev_ptr(prefix_out, "prefix_out", _I) if prefix_out is not None else None
            if prefix_in is not None:
                need(prefix_in, b, "prefix_in")
            if prefix_out is not None:
                need(prefix_out, b, "prefix_out")
            _lib.check(_lib.lib().epnet_sample_centres_chain(b, n, m, pp, px, nb, pt, pi, pn, pin, pout, int(prefix_cap), s),
                       "sample_centres")
    return 1


@writes("idxs")
def ball_query_multi_wrapper(b, n, m, radii, nsamples, new_xyz, xyz, index, idxs):
    """the ball queries of all scales of an MSG level in one launch (same results as one ball_query per scale)"""
    import ctypes
    pn, pxyz = dev_ptr(new_xyz, "new_xyz", _F), dev_ptr(xyz, "xyz", _F)
    need(new_xyz, b * m * 3, "new_xyz"); need(xyz, b * n * 3, "xyz")
    k = len(radii)
    assert len(nsamples) == k and len(idxs) == k
    ptrs = []
    for ns, t in zip(nsamples, idxs):
        ptrs.append(dev_ptr(t, "idx", _I))
        need(t, b * m * ns, "idx")
    px, nb = _index_args(index, xyz)
    c_r = (ctypes.c_float * k)(*[float(r) for r in radii])
    c_n = (ctypes.c_int * k)(*[int(x) for x in nsamples])
    c_p = (ctypes.c_void_p * k)(*ptrs)
    with on_device_of(xyz) as s:
        _lib.check(_lib.lib().epnet_ball_query_indexed_multi(b, n, m, k, ctypes.cast(c_r, ctypes.c_void_p),
                                                             ctypes.cast(c_n, ctypes.c_void_p), pn, pxyz, px, nb,
                                                             ctypes.cast(c_p, ctypes.c_void_p), s), "ball_query")
    return 1


@writes("idxs")
def ball_query_ordered_wrapper(b, n, m, radii, nsamples, new_xyz, xyz, index, centre_index, idxs):
    """ball_query_multi_wrapper with the centres served in their own spatial order: centre_index = the scene index of the cloud new_xyz
    (the next SA level's index of its input points); None falls back to ball_query_multi_wrapper's path. Same results."""
    import ctypes
    pn, pxyz = dev_ptr(new_xyz, "new_xyz", _F), dev_ptr(xyz, "xyz", _F)
    need(new_xyz, b * m * 3, "new_xyz"); need(xyz, b * n * 3, "xyz")
    k = len(radii)
    assert len(nsamples) == k and len(idxs) == k
    ptrs = []
    for ns, t in zip(nsamples, idxs):
        ptrs.append(dev_ptr(t, "idx", _I))
        need(t, b * m * ns, "idx")
    px, nb = _index_args(index, xyz)
    pc, nc = _index_args(centre_index, new_xyz)
    c_r = (ctypes.c_float * k)(*[float(r) for r in radii])
    c_n = (ctypes.c_int * k)(*[int(x) for x in nsamples])
    c_p = (ctypes.c_void_p * k)(*ptrs)
    with on_device_of(xyz) as s:
        _lib.check(_lib.lib().epnet_ball_query_ordered(b, n, m, k, ctypes.cast(c_r, ctypes.c_void_p), ctypes.cast(c_n, ctypes.c_void_p),
                                                     pn, pxyz, px, nb, pc, nc, ctypes.cast(c_p, ctypes.c_void_p), s), "ball_query")
    return 1


@writes("dist2", "idx")
def three_nn_indexed_wrapper(b, n, m, unknown, known, unknown_index, known_index, dist2, idx):
    """three_nn_wrapper over scene indices of `known` and (optionally) of `unknown` (same results)"""
    pu, pk = dev_ptr(unknown, "unknown", _F), dev_ptr(known, "known", _F)
    pd, pi = dev_ptr(dist2, "dist2", _F), dev_ptr(idx, "idx", _I)
    need(unknown, b * n * 3, "unknown"); need(known, b * m * 3, "known"); need(dist2, b * n * 3, "dist2"); need(idx, b * n * 3, "idx")
    pux, nu = _index_args(unknown_index, unknown)
    pkx, nk = _index_args(known_index, known)
    with on_device_of(unknown) as s:
        _lib.check(_lib.lib().epnet_three_nn_indexed(b, n, m, pu, pk, pux, nu, pkx, nk, pd, pi, s), "three_nn")
    return 1


@writes("idx")
def ball_query_indexed_wrapper(b, n, m, radius, nsample, new_xyz, xyz, index, idx):
    """ball_query_wrapper over a scene index of `xyz` (same results)"""
    pn, pxyz, pi = dev_ptr(new_xyz, "new_xyz", _F), dev_ptr(xyz, "xyz", _F), dev_ptr(idx, "idx", _I)
    need(new_xyz, b * m * 3, "new_xyz"); need(xyz, b * n * 3, "xyz"); need(idx, b * m * nsample, "idx")
    px, nb = _index_args(index, xyz)
    with on_device_of(xyz) as s:
        _lib.check(_lib.lib().epnet_ball_query_indexed(b, n, m, radius, nsample, pn, pxyz, px, nb, pi, s), "ball_query")
    return 1

@writes("out")
def group_concat_wrapper(b, c, n, npoints, nsample, xyz, new_xyz, features, idx, out, use_xyz, workspace=None):
    """out (B, 3+C | C, M, ns) = [grouped xyz - centre ; grouped features]; features may be None when c == 0.
    workspace (optional): a uint8 tensor of group_concat_workspace_bytes(...) bytes the caller keeps (rows too long for on-chip
    staging go through a point-major copy of the features there); None: allocated per call"""
    px = dev_ptr(xyz, "xyz", _F) if use_xyz else None
    pn = dev_ptr(new_xyz, "new_xyz", _F) if use_xyz else None
    pf = dev_ptr(features, "features", _F) if c else None
    pi, po = dev_ptr(idx, "idx", _I), dev_ptr(out, "out", _F)
    need(idx, b * npoints * nsample, "idx"); need(out, b * ((3 if use_xyz else 0) + c) * npoints * nsample, "out")
    if use_xyz:
        need(xyz, b * n * 3, "xyz"); need(new_xyz, b * npoints * 3, "new_xyz")
    if c:
        need(features, b * c * n, "features")
    l = _lib.lib()
    ws_bytes = l.epnet_group_concat_workspace_bytes(b, c, n, npoints, nsample) if c else 0
    with on_device_of(idx) as s:
        if ws_bytes:   # rows too long for on-chip staging: scratch for a point-major copy of the features
            ws = workspace if workspace is not None else torch.empty((ws_bytes,), dtype=torch.uint8, device=idx.device)
            if ws.dtype != torch.uint8 or ws.numel() < ws_bytes or ws.device != idx.device or not ws.is_contiguous():
                raise RuntimeError("group_concat workspace: a contiguous uint8 tensor of >= %d bytes on the tensors' device" % ws_bytes)
            _lib.check(l.epnet_group_concat_ws(b, c, n, npoints, nsample, px, pn, pf, pi, po, int(bool(use_xyz)), ws.data_ptr(),
                                               ws.numel(), s), "group_concat")
        else:
            _lib.check(l.epnet_group_concat(b, c, n, npoints, nsample, px, pn, pf, pi, po, int(bool(use_xyz)), s), "group_concat")
    return 1


def group_concat_workspace_bytes(b, c, n, npoints, nsample):
    """scratch bytes group_concat_wrapper can use for this shape (0: none helps)"""
    return int(_lib.lib().epnet_group_concat_workspace_bytes(b, c, n, npoints, nsample)) if c else 0


@writes("outs")
def group_concat_multi_wrapper(b, c, n, npoints, nsamples, xyz, new_xyz, features, idxs, outs, use_xyz):
    """group_concat_wrapper for all scales of an MSG level in one call (the feature rows are staged once for two scales)"""
    import ctypes
    k = len(nsamples)
    assert len(idxs) == k and len(outs) == k
    px = dev_ptr(xyz, "xyz", _F) if use_xyz else None
    pn = dev_ptr(new_xyz, "new_xyz", _F) if use_xyz else None
    pf = dev_ptr(features, "features", _F) if c else None
    pis, pos = [], []
    for ns, i, o in zip(nsamples, idxs, outs):
        pis.append(dev_ptr(i, "idx", _I)); pos.append(dev_ptr(o, "out", _F))
        need(i, b * npoints * ns, "idx"); need(o, b * ((3 if use_xyz else 0) + c) * npoints * ns, "out")
    if use_xyz:
        need(xyz, b * n * 3, "xyz"); need(new_xyz, b * npoints * 3, "new_xyz")
    if c:
        need(features, b * c * n, "features")
    c_n = (ctypes.c_int * k)(*[int(x) for x in nsamples])
    c_i = (ctypes.c_void_p * k)(*pis)
    c_o = (ctypes.c_void_p * k)(*pos)
    with on_device_of(idxs[0]) as s:
        _lib.check(_lib.lib().epnet_group_concat_multi(b, c, n, npoints, k, ctypes.cast(c_n, ctypes.c_void_p), px, pn, pf,
                                                       ctypes.cast(c_i, ctypes.c_void_p), ctypes.cast(c_o, ctypes.c_void_p),
                                                       int(bool(use_xyz)), s), "group_concat")
    return 1


@writes("grad_features")
def group_concat_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_features, use_xyz):
    pg, pi, pp = dev_ptr(grad_out, "grad_out", _F), dev_ptr(idx, "idx", _I), dev_ptr(grad_features, "grad_features", _F)
    need(grad_out, b * ((3 if use_xyz else 0) + c) * npoints * nsample, "grad_out"); need(grad_features, b * c * n, "grad_features")
    l = _lib.lib()
    ws_bytes = l.epnet_group_points_grad_workspace_bytes(b, n, npoints, nsample)
    with on_device_of(grad_out) as s:
        if ws_bytes:
            ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=grad_out.device)
            _lib.check(l.epnet_group_concat_grad_ws(b, c, n, npoints, nsample, pg, pi, pp, int(bool(use_xyz)),
                                                    ws.data_ptr(), ws_bytes, s), "group_concat_grad")
        else:
            _lib.check(l.epnet_group_concat_grad(b, c, n, npoints, nsample, pg, pi, pp, int(bool(use_xyz)), s),
                       "group_concat_grad")
    return 1


@writes("out", "arg")
def pool_max_wrapper(rows, nsample, x, out, arg):
    """the SA level's neighbourhood max-pool (F.max_pool2d(kernel_size=[1, nsample]), pointnet2_modules.py:61-68): x
    (rows, nsample) -> out (rows), arg (rows) int32 or None (not in the reference extension; see epnet_ops.h)"""
    px, po = dev_ptr(x, "x", _F), dev_ptr(out, "out", _F)
    pa = dev_ptr(arg, "arg", _I) if arg is not None else None
    need(x, rows * nsample, "x"); need(out, rows, "out")
    if arg is not None:
        need(arg, rows, "arg")
    with on_device_of(x) as s:
        _lib.check(_lib.lib().epnet_pool_max(rows, nsample, px, po, pa, s), "pool_max")
    return 1


@writes("grad_x")
def pool_max_grad_wrapper(rows, nsample, grad_out, arg, grad_x):
    pg, pa, px = dev_ptr(grad_out, "grad_out", _F), dev_ptr(arg, "arg", _I), dev_ptr(grad_x, "grad_x", _F)
    need(grad_out, rows, "grad_out"); need(arg, rows, "arg"); need(grad_x, rows * nsample, "grad_x")
    with on_device_of(grad_out) as s:
        _lib.check(_lib.lib().epnet_pool_max_grad(rows, nsample, pg, pa, px, s), "pool_max_grad")
    return 1


@writes("out")
def group_linear_wrapper(b, c, n, npoints, nsample, xyz, new_xyz, z, idx, w_xyz, bias, out):
    """the first shared-MLP layer of an SA level folded into its grouping: out (b,c,npoints,nsample) =
    z[:, :, idx] + w_xyz . (xyz[idx] - centre) (+ bias); z (b,c,n) = W_f . features (not in the reference extension; see
    epnet_ops.h)"""
    px, pn, pz = dev_ptr(xyz, "xyz", _F), dev_ptr(new_xyz, "new_xyz", _F), dev_ptr(z, "z", _F)
    pi, pw, po = dev_ptr(idx, "idx", _I), dev_ptr(w_xyz, "w_xyz", _F), dev_ptr(out, "out", _F)
    pb = dev_ptr(bias, "bias", _F) if bias is not None else None
    need(xyz, b * n * 3, "xyz"); need(new_xyz, b * npoints * 3, "new_xyz"); need(z, b * c * n, "z")
    need(idx, b * npoints * nsample, "idx"); need(w_xyz, c * 3, "w_xyz"); need(out, b * c * npoints * nsample, "out")
    if bias is not None:
        need(bias, c, "bias")
    with on_device_of(z) as s:
        _lib.check(_lib.lib().epnet_group_linear(b, c, n, npoints, nsample, px, pn, pz, pi, pw, pb, po, s), "group_linear")
    return 1


@writes("grad_w")
def group_linear_grad_w_wrapper(b, c, n, npoints, nsample, grad_out, xyz, new_xyz, idx, grad_w):
    """grad_w (c,3), zero-filled by the caller, += sum of grad_out[b,co,m,s] * (xyz[idx] - centre)[k]"""
    pg, px, pn = dev_ptr(grad_out, "grad_out", _F), dev_ptr(xyz, "xyz", _F), dev_ptr(new_xyz, "new_xyz", _F)
    pi, pw = dev_ptr(idx, "idx", _I), dev_ptr(grad_w, "grad_w", _F)
    need(grad_out, b * c * npoints * nsample, "grad_out"); need(xyz, b * n * 3, "xyz"); need(new_xyz, b * npoints * 3, "new_xyz")
    need(idx, b * npoints * nsample, "idx"); need(grad_w, c * 3, "grad_w")
    with on_device_of(grad_out) as s:
        _lib.check(_lib.lib().epnet_group_linear_grad_w(b, c, n, npoints, nsample, pg, px, pn, pi, pw, s), "group_linear_grad_w")
    return 1


@writes("out", "xy_out")
def feature_gather_wrapper(b, c, h, w, n_src, n, align_corners, feature_map, xy, idx, out, xy_out):
    """LI-Fusion's point-to-pixel bilinear sampler (lib/net/pointnet2_msg.py:107-120) with the xy gather over the FPS
    indices folded in (not in the reference extension; see epnet_ops.h)"""
    pf, px, po = dev_ptr(feature_map, "feature_map", _F), dev_ptr(xy, "xy", _F), dev_ptr(out, "out", _F)
    pi = dev_ptr(idx, "idx", _I) if idx is not None else None
    pxo = dev_ptr(xy_out, "xy_out", _F) if xy_out is not None else None
    need(feature_map, b * c * h * w, "feature_map"); need(xy, b * n_src * 2, "xy"); need(out, b * c * n, "out")
    if idx is not None:
        need(idx, b * n, "idx")
    if xy_out is not None:
        need(xy_out, b * n * 2, "xy_out")
    with on_device_of(feature_map) as s:
        _lib.check(_lib.lib().epnet_feature_gather(b, c, h, w, n_src, n, int(bool(align_corners)), pf, px, pi, po, pxo, s),
                   "feature_gather")
    return 1


@writes("grad_feature_map")
def feature_gather_grad_wrapper(b, c, h, w, n, align_corners, grad_out, xy, grad_feature_map):
    pg, px, pf = dev_ptr(grad_out, "grad_out", _F), dev_ptr(xy, "xy", _F), dev_ptr(grad_feature_map, "grad_feature_map", _F)
    need(grad_out, b * c * n, "grad_out"); need(xy, b * n * 2, "xy"); need(grad_feature_map, b * c * h * w, "grad_feature_map")
    with on_device_of(grad_out) as s:
        _lib.check(_lib.lib().epnet_feature_gather_grad(b, c, h, w, n, int(bool(align_corners)), pg, px, pf, s), "feature_gather_grad")
    return 1

