"""A seeded sweep over the shapes BETWEEN the hand-picked parity cases: every sampling / query / interpolation entry point picks
its kernel by size (one wave, register-resident, pruned over the scene index, big-scene, streaming; one or two centres per wave,
bitmap or streaming hit lists; tile or per-unknown three_nn ...), so the cases below sit on and around those boundaries, with
odd counts, m = 1, nsample = 1, lattice clouds full of equal distances, duplicated rows and single-cell clouds. Bit-exact against
the oracle (indices, squared distances); float sums within 1e-5.

The generator is seeded: the same cases every run (a failure names its case, which can be replayed alone with -k)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _lib_loaded(hiplib):
    assert torch.cuda.is_available()
    return hiplib


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(autouse=True)
def guarded_allocations(monkeypatch):
    """canaries around everything the wrappers / the operator layer allocate and around the outputs below (conftest.GuardedAlloc)"""
    from conftest import install_guards
    g = install_guards(monkeypatch)
    _CURRENT.append(g)
    yield g
    _CURRENT.pop()
    torch.cuda.synchronize()
    g.check()


_CURRENT = []


def out_tensor(shape, dtype=torch.float32, fill=None):
    """an output tensor of a direct wrapper call, guarded like the library's own allocations"""
    t = _CURRENT[-1].alloc(shape, dtype, DEV)
    if fill is not None:
        t.fill_(fill)
    return t


def assert_scatter_sum(got, terms, flat, targets):
    """got (b, c, targets) against the float64 scatter-add of terms (b, c, p) to targets flat (b, p). The order of a target's terms
    is free (atomicAdd in the reference), so the bound is relative to the sum of magnitudes of its list, not to the result"""
    b, c, _ = terms.shape
    want = np.zeros((b, c, targets))
    mag = np.zeros((b, c, targets))
    for s_ in range(b):
        np.add.at(want[s_].T, flat[s_], terms[s_].T.astype(np.float64))
        np.add.at(mag[s_].T, flat[s_], np.abs(terms[s_].T).astype(np.float64))
    err = np.abs(got.astype(np.float64) - want)
    assert (err <= 2e-7 * mag + 1e-5 * np.maximum(1.0, np.abs(want))).all(), float(err.max())
    return want


KINDS = ("kitti", "ubox", "dup", "kitti_q", "lattice", "blob", "line")


def cloud(kind, b, n, seed):
    """(b, n, 3) float32. Beyond epnet_amd.synth's families: `lattice` = integer grid points in random order (exact distance ties
    between distinct points everywhere), `blob` = everything inside one cell of any spatial index (1 cm), `line` = collinear"""
    from epnet_amd import synth
    rng = np.random.default_rng(seed)
    if kind in synth.KINDS:
        return synth.scenes(kind, b, n, seed=seed).numpy()
    if kind == "lattice":
        side = int(np.ceil(n ** (1.0 / 3.0))) + 1
        g = np.stack(np.meshgrid(np.arange(side), np.arange(side), np.arange(side), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
        return np.stack([g[rng.permutation(len(g))[:n]] * np.float32(0.25) for _ in range(b)])
    if kind == "blob":
        return (rng.random((b, n, 3)) * 0.01 + np.array([3.0, -1.0, 20.0])).astype(np.float32)
    if kind == "line":
        t = rng.random((b, n, 1)) * 60.0
        return (t * np.array([0.6, 0.0, 0.8]) + np.array([-10.0, 1.0, 0.0])).astype(np.float32)
    raise ValueError(kind)


# sizes on and around the kernel-selection boundaries
_N_EDGES = (1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096,
            4097, 8191, 8192, 8193, 16383, 16384, 16385, 20000, 32768, 32769, 65535, 65536, 65537, 70001)


def _cases(count, seed, n_max, with_m=True):
    rng = np.random.default_rng(seed)
    edges = [n for n in _N_EDGES if n <= n_max]
    out = []
    for i in range(count):
        n = int(edges[i % len(edges)]) if i < 2 * len(edges) else int(np.exp(rng.uniform(0, np.log(n_max))))
        n = max(1, min(n, n_max))
        b = int(rng.integers(1, 4)) if n <= 20000 else 1
        m = int(min(n, max(1, np.exp(rng.uniform(0, np.log(max(2, min(n, 5000)))))))) if with_m else 0
        out.append((i, b, n, m, KINDS[int(rng.integers(0, len(KINDS)))]))
    return out


@pytest.mark.parametrize("case,b,n,m,kind", _cases(96, 11, 70001))
def test_sweep_furthest_point_sampling(oracle, case, b, n, m, kind):
    from epnet_amd import pointnet2_cuda as ext
    if n * m > 3.0e8:
        m = max(1, int(3.0e8 // n))            # (keeps the oracle's O(n m) scan within seconds)
    xyz = cloud(kind, b, n, seed=1000 + case)
    d_xyz = dev(xyz)
    want = oracle.furthest_point_sampling(xyz, m)
    # the extension's own entry point (picks wave / register-resident / big-scene / streaming kernels by n) ...
    temp = out_tensor((b, n), torch.float32, 1e10)
    idx = out_tensor((b, m), torch.int32, -9)
    ext.furthest_point_sampling_wrapper(b, n, m, d_xyz, temp, idx)
    np.testing.assert_array_equal(host(idx), want)
    # ... and the pruned kernels over a scene index, where one exists for this size
    index = ext.scene_index(d_xyz)
    if index is not None:
        temp.fill_(1e10)
        idx.fill_(-9)
        ext.furthest_point_sampling_indexed_wrapper(b, n, m, d_xyz, index, temp, idx)
        np.testing.assert_array_equal(host(idx), want)


@pytest.mark.parametrize("case,b,n,m,kind", _cases(96, 12, 65536))
@pytest.mark.parametrize("pair", ["0", "1"])
def test_sweep_ball_query(oracle, case, b, n, m, kind, pair, monkeypatch):
    from epnet_amd import pointnet2_utils as p2u
    monkeypatch.setenv("EPNET_BQ_PAIR", pair)
    rng = np.random.default_rng(5000 + case)
    m = min(m, 600)
    xyz = cloud(kind, b, n, seed=2000 + case)
    extent = float(np.ptp(xyz[0], axis=0).max()) or 1.0
    radius = float(np.exp(rng.uniform(np.log(extent / 300.0), np.log(extent * 1.5))))
    ns = int(rng.choice([1, 2, 5, 16, 32, 63, 64, 65, 100]))
    centres = np.ascontiguousarray(xyz[:, rng.permutation(n)[:m]] + (rng.random((b, m, 3)) < 0.3) * rng.normal(0, radius / 2, (b, m, 3))).astype(np.float32)
    got = host(p2u.ball_query(radius, ns, dev(xyz), dev(centres)))
    np.testing.assert_array_equal(got, oracle.ball_query(radius, ns, xyz, centres))


@pytest.mark.parametrize("case,b,n,m,kind", _cases(64, 13, 40000))
def test_sweep_three_nn_and_interpolation(oracle, case, b, n, m, kind):
    """unknown set of n points, known set = m of them (an FP module's shapes) or an unrelated cloud"""
    from epnet_amd import pointnet2_cuda as ext
    rng = np.random.default_rng(7000 + case)
    unknown = cloud(kind, b, n, seed=3000 + case)
    if case % 3 == 0:
        known = cloud(KINDS[case % len(KINDS)], b, m, seed=4000 + case)
    else:
        known = np.ascontiguousarray(unknown[:, rng.permutation(n)[:m]])
    d2 = out_tensor((b, n, 3), torch.float32, -1.0)
    idx = out_tensor((b, n, 3), torch.int32, -9)
    ext.three_nn_wrapper(b, n, m, dev(unknown), dev(known), d2, idx)
    o_d2, o_idx = oracle.three_nn(unknown, known)
    np.testing.assert_array_equal(host(idx), o_idx)
    np.testing.assert_array_equal(host(d2), o_d2)
    c = int(rng.choice([1, 3, 4, 7, 32, 128]))
    feats = rng.standard_normal((b, c, m)).astype(np.float32)
    safe = np.clip(o_idx, 0, m - 1)                         # (m < 3: the reference leaves unset slots at index 0 already)
    w = rng.random((b, n, 3)).astype(np.float32)
    w /= w.sum(-1, keepdims=True)
    out = out_tensor((b, c, n))
    ext.three_interpolate_wrapper(b, c, m, n, dev(feats), dev(safe), dev(w), out)
    np.testing.assert_array_equal(host(out), oracle.three_interpolate(feats, safe, w))
    go = rng.standard_normal((b, c, n)).astype(np.float32)
    grad = out_tensor((b, c, m), torch.float32, 0)
    ext.three_interpolate_grad_wrapper(b, c, n, m, dev(go), dev(safe), dev(w), grad)
    terms = (go[:, :, :, None] * w[:, None, :, :]).reshape(b, c, n * 3)          # (fp32 products, as the kernels form them)
    want = assert_scatter_sum(host(grad), terms, safe.reshape(b, -1).astype(np.int64), m)
    np.testing.assert_allclose(oracle.three_interpolate_grad(go, safe, w, m), want, rtol=1e-3, atol=1e-2 * max(1.0, n / m / 30.0))   # the oracle agrees with the yardstick


@pytest.mark.parametrize("case,b,n,m,kind", _cases(64, 14, 20000))
def test_sweep_grouping_and_gather(oracle, case, b, n, m, kind):
    from epnet_amd import pointnet2_cuda as ext
    rng = np.random.default_rng(9000 + case)
    c = int(rng.choice([1, 3, 5, 16, 64, 96, 130]))
    ns = int(rng.choice([1, 3, 16, 32, 64]))
    m = min(m, 700)
    feats = rng.standard_normal((b, c, n)).astype(np.float32)
    idx = rng.integers(0, n, size=(b, m, ns)).astype(np.int32)
    out = out_tensor((b, c, m, ns))
    ext.group_points_wrapper(b, c, n, m, ns, dev(feats), dev(idx), out)
    np.testing.assert_array_equal(host(out), oracle.group_points(feats, idx))
    go = rng.standard_normal((b, c, m, ns)).astype(np.float32)
    grad = out_tensor((b, c, n), torch.float32, 0)
    ext.group_points_grad_wrapper(b, c, n, m, ns, dev(go), dev(idx), grad)
    want = assert_scatter_sum(host(grad), go.reshape(b, c, -1), idx.reshape(b, -1).astype(np.int64), n)
    np.testing.assert_allclose(oracle.group_points_grad(go, idx, n), want, rtol=1e-3, atol=1e-2 * max(1.0, m * ns / n / 30.0))
    gi = rng.integers(0, n, size=(b, m)).astype(np.int32)
    g_out = out_tensor((b, c, m))
    ext.gather_points_wrapper(b, c, n, m, dev(feats), dev(gi), g_out)
    np.testing.assert_array_equal(host(g_out), oracle.gather_points(feats, gi))
    ggo = rng.standard_normal((b, c, m)).astype(np.float32)
    ggrad = out_tensor((b, c, n), torch.float32, 0)
    ext.gather_points_grad_wrapper(b, c, n, m, dev(ggo), dev(gi), ggrad)
    want = assert_scatter_sum(host(ggrad), ggo, gi.astype(np.int64), n)
    np.testing.assert_allclose(oracle.gather_points_grad(ggo, gi, n), want, rtol=1e-3, atol=1e-2 * max(1.0, m / n / 30.0))


def _pyramids(count, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(count):
        n = int(rng.choice([1024, 1025, 2048, 3000, 4096, 5000, 8192, 16384, 16385, 20000]))
        levels = int(rng.integers(2, 5))
        npoints, cur = [], n
        for _ in range(levels):
            cur = max(1, int(cur / float(rng.choice([1.0, 1.5, 2.0, 4.0, 7.3]))))
            npoints.append(cur)
        out.append((i, int(rng.integers(1, 4)), n, tuple(npoints), KINDS[int(rng.integers(0, len(KINDS)))]))
    return out


@pytest.mark.parametrize("case,b,n,npoints,kind", _pyramids(40, 15))
def test_sweep_sampling_pyramid(oracle, case, b, n, npoints, kind):
    """the chain of samplings of an SA pyramid (levels 2.. take the identity where the level above reports a tie-free prefix, replay
    or run their rounds otherwise) on clouds with and without ties: every level bit-equal to furthest point sampling of the
    level above's centres"""
    from epnet_amd import pointnet2_utils as p2u
    xyz = cloud(kind, b, n, seed=6000 + case)
    levels = p2u.sample_pyramid(dev(xyz), list(npoints))
    torch.cuda.synchronize()
    cur = xyz
    for (idx, new_xyz, _event, _index), m in zip(levels, npoints):
        want = oracle.furthest_point_sampling(cur, m)
        np.testing.assert_array_equal(host(idx), want)
        cur = np.ascontiguousarray(np.take_along_axis(cur, want[..., None].astype(np.int64), axis=1))
        np.testing.assert_array_equal(host(new_xyz), cur)


@pytest.mark.parametrize("case,b,n,m,kind", _cases(48, 16, 20000))
def test_sweep_query_and_group(oracle, case, b, n, m, kind):
    """QueryAndGroup (ball query over a scene index + the fused [xyz - centre ; features] grouping) against the oracle's ball
    query and grouping, composed as pointnet2_utils.py:249-257 composes them"""
    from epnet_amd import pointnet2_utils as p2u
    rng = np.random.default_rng(11000 + case)
    m = min(m, 500)
    c = int(rng.choice([0, 1, 3, 16, 19, 64, 96]))
    ns = int(rng.choice([1, 4, 7, 16, 32, 64]))
    use_xyz = bool(rng.integers(0, 2)) or c == 0
    xyz = cloud(kind, b, n, seed=8000 + case)
    extent = float(np.ptp(xyz[0], axis=0).max()) or 1.0
    radius = float(np.exp(rng.uniform(np.log(extent / 100.0), np.log(extent))))
    centres = np.ascontiguousarray(xyz[:, rng.permutation(n)[:m]])
    feats = rng.standard_normal((b, c, n)).astype(np.float32) if c else None
    got = p2u.QueryAndGroup(radius, ns, use_xyz=use_xyz)(dev(xyz), dev(centres), dev(feats) if c else None)
    idx = oracle.ball_query(radius, ns, xyz, centres)
    parts = []
    if use_xyz:
        parts.append(oracle.group_points(np.ascontiguousarray(xyz.transpose(0, 2, 1)), idx) - centres.transpose(0, 2, 1)[..., None])
    if c:
        parts.append(oracle.group_points(feats, idx))
    np.testing.assert_array_equal(host(got), np.concatenate(parts, axis=1))


@pytest.mark.parametrize("case,b,n,m,kind", _cases(40, 17, 40000))
@pytest.mark.parametrize("tile", ["1", "1000000000"])
def test_sweep_three_nn_over_scene_indices(oracle, case, b, n, m, kind, tile, monkeypatch):
    """epnet_three_nn_indexed: a wave per bucket of unknowns (forced with EPNET_NN_TILE_MIN_BUCKETS=1) or per unknown, over the scene
    indices of whichever of the two sets has one"""
    from epnet_amd import pointnet2_cuda as ext
    monkeypatch.setenv("EPNET_NN_TILE_MIN_BUCKETS", tile)
    rng = np.random.default_rng(13000 + case)
    m = max(m, int(rng.choice([1, 1024, 1100, 2259, 4096])))
    unknown = cloud(kind, b, n, seed=9000 + case)
    known = cloud(KINDS[(case + 2) % len(KINDS)], b, m, seed=9500 + case)
    if m >= 8 and n >= 8:
        known[:, :4] = unknown[:, :4]          # exact zero distances
        known[:, 4:8] = known[:, :4]           # ... and ties between equal known rows
    d_u, d_k = dev(unknown), dev(known)
    o_d2, o_i = oracle.three_nn(unknown, known)
    ui, ki = ext.scene_index(d_u), ext.scene_index(d_k)
    d2 = out_tensor((b, n, 3), torch.float32, -1.0)
    i = out_tensor((b, n, 3), torch.int32, -1)
    ext.three_nn_indexed_wrapper(b, n, m, d_u, d_k, ui, ki, d2, i)
    np.testing.assert_array_equal(host(i), o_i)
    np.testing.assert_array_equal(host(d2), o_d2)


def _pool_cases(count, seed):
    rng = np.random.default_rng(seed)
    return [(i, int(rng.integers(1, 4)), int(np.exp(rng.uniform(np.log(1), np.log(20000)))), int(rng.integers(1, 70)),
             int(rng.choice([0, 1, 3, 16, 130])), int(rng.choice([1, 16, 33, 512, 600]))) for i in range(count)]


@pytest.mark.parametrize("case,b,n,m,c,s", _pool_cases(32, 18))
def test_sweep_roipool3d(oracle, case, b, n, m, c, s):
    from epnet_amd import kitti_utils, roipool3d_cuda as ext, synth
    pts = cloud("kitti", b, n, seed=12000 + case)
    boxes = np.stack([synth.proposal_boxes(m, seed=12500 + case + i, num_objects=40, jitter=0.5)[0].numpy() for i in range(b)])
    boxes = kitti_utils.enlarge_box3d(boxes.reshape(-1, 7), 0.2 * (case % 3)).reshape(b, m, 7)
    feat = np.random.default_rng(case).standard_normal((b, n, c)).astype(np.float32)
    out = out_tensor((b, m, s, 3 + c), torch.float32, 0)
    flag = out_tensor((b, m), torch.int32, 0)
    ext.forward(dev(pts), dev(boxes), dev(feat), out, flag)
    o_pool, o_flag = oracle.roipool3d(pts, boxes, feat, s)
    np.testing.assert_array_equal(host(flag), o_flag)
    np.testing.assert_array_equal(host(out), o_pool)


@pytest.mark.parametrize("case", range(24))
def test_sweep_nms_and_iou(oracle, case):
    """axis-aligned NMS keep lists bit-exact; rotated overlap / IoU within 1e-5; rotated NMS with a threshold kept away from any
    pair's IoU (the trigonometry may differ in the last bit)"""
    from epnet_amd import iou3d_cuda as ext, kitti_utils, synth
    rng = np.random.default_rng(14000 + case)
    n = int(rng.choice([1, 2, 63, 64, 65, 300, 1000, 2049]))
    bx, sc = synth.proposal_boxes(n, seed=14500 + case, num_objects=max(2, n // int(rng.choice([3, 20]))), jitter=float(rng.choice([0.2, 1.5])))
    bev = kitti_utils.boxes3d_to_bev_torch(bx).numpy()
    sorted_boxes = np.ascontiguousarray(bev[np.argsort(-sc.numpy(), kind="stable")])
    thr = float(rng.choice([0.1, 0.5, 0.85]))
    keep = torch.zeros((n,), dtype=torch.int64)
    num = ext.nms_normal_gpu(dev(sorted_boxes), keep, thr)
    o_keep = oracle.nms(sorted_boxes, thr, False)
    assert num == len(o_keep)
    np.testing.assert_array_equal(keep[:num].numpy(), o_keep)
    nb = min(n, 400)
    a, b_ = sorted_boxes[:nb], sorted_boxes[::-1][:max(1, nb // 3)].copy()
    for fn, ofn in ((ext.boxes_overlap_bev_gpu, oracle.boxes_overlap_bev), (ext.boxes_iou_bev_gpu, oracle.boxes_iou_bev)):
        ans = out_tensor((a.shape[0], b_.shape[0]), torch.float32, 0)
        fn(dev(a), dev(b_), ans)
        np.testing.assert_allclose(host(ans), ofn(a, b_), rtol=0, atol=1e-5)
    small = sorted_boxes[:min(n, 700)]
    full = oracle.boxes_iou_bev(small, small)
    while np.abs(full - thr).min() < 1e-5:
        thr += 0.0137
    k2, n2 = ext.nms_device(dev(small), thr)
    np.testing.assert_array_equal(host(k2[:int(n2.item())]), oracle.nms(small, thr, True))


@pytest.mark.parametrize("case,b,n,m,kind", _cases(48, 19, 65536))
@pytest.mark.parametrize("pair", ["0", "1"])
def test_sweep_msg_level(oracle, case, b, n, m, kind, pair, monkeypatch):
    """one MSG level as the SA stack issues it: the ball queries of all scales in one launch (centre order and the centres' own
    spatial order), the groupings of all scales in one call, the neighbourhood max-pool -- 1 to 3 scales, nested or not"""
    from epnet_amd import pointnet2_cuda as ext
    monkeypatch.setenv("EPNET_BQ_PAIR", pair)
    rng = np.random.default_rng(15000 + case)
    if n < 1024:
        n = int(rng.choice([1024, 1500, 2048, 2049]))          # (the multi-scale entry points take a scene index)
    m = int(min(m, 400, n))
    k = int(rng.integers(1, 4))
    c = int(rng.choice([0, 4, 16, 19, 64]))
    xyz = cloud(kind, b, n, seed=10000 + case)
    extent = float(np.ptp(xyz[0], axis=0).max()) or 1.0
    radii = [float(np.exp(rng.uniform(np.log(extent / 200.0), np.log(extent)))) for _ in range(k)]
    nss = [int(rng.choice([1, 4, 8, 16, 32, 64, 72])) for _ in range(k)]
    centres = np.ascontiguousarray(xyz[:, rng.permutation(n)[:m]])
    d_xyz, d_c = dev(xyz), dev(centres)
    index = ext.scene_index(d_xyz)
    want = [oracle.ball_query(r, ns, xyz, centres) for r, ns in zip(radii, nss)]
    outs = [out_tensor((b, m, ns), torch.int32, -5) for ns in nss]
    ext.ball_query_multi_wrapper(b, n, m, radii, nss, d_c, d_xyz, index, outs)
    for got, w in zip(outs, want):
        np.testing.assert_array_equal(host(got), w)
    ci = ext.scene_index(d_c)                                     # None below 1024 centres: the wrapper falls back
    outs2 = [out_tensor((b, m, ns), torch.int32, -5) for ns in nss]
    ext.ball_query_ordered_wrapper(b, n, m, radii, nss, d_c, d_xyz, index, ci, outs2)
    for got, w in zip(outs2, want):
        np.testing.assert_array_equal(host(got), w)
    feats = rng.standard_normal((b, c, n)).astype(np.float32) if c else None
    grouped = [out_tensor((b, 3 + c, m, ns), torch.float32, float("nan")) for ns in nss]
    ext.group_concat_multi_wrapper(b, c, n, m, nss, d_xyz, d_c, dev(feats) if c else None, outs, grouped, True)
    xyz_t = np.ascontiguousarray(xyz.transpose(0, 2, 1))
    for got, w in zip(grouped, want):
        parts = [oracle.group_points(xyz_t, w) - centres.transpose(0, 2, 1)[..., None]]
        if c:
            parts.append(oracle.group_points(feats, w))
        ref = np.concatenate(parts, axis=1)
        np.testing.assert_array_equal(host(got), ref)
        rows, ns = ref.shape[0] * ref.shape[1] * ref.shape[2], ref.shape[3]
        pooled = out_tensor((rows,))
        arg = out_tensor((rows,), torch.int32)
        ext.pool_max_wrapper(rows, ns, got, pooled, arg)
        o_max, o_arg = oracle.pool_max(ref)
        np.testing.assert_array_equal(host(pooled), o_max.reshape(-1))
        np.testing.assert_array_equal(host(arg), o_arg.reshape(-1))


def _scene_cases(count, seed):
    rng = np.random.default_rng(seed)
    return [(i, int(rng.choice([5, 7, 8, 9, 15, 16, 17, 33, 64, 130])), int(rng.choice([64, 257, 1024, 1100, 2048, 4096])),
             KINDS[int(rng.integers(0, len(KINDS)))]) for i in range(count)]


@pytest.mark.parametrize("case,b,n,kind", _scene_cases(30, 20))
def test_sweep_many_scenes(oracle, case, b, n, kind):
    """scene counts that are no multiple of the 8 XCDs (every multi-workgroup kernel relabels its workgroups so that a scene's land
    on one XCD: common.h xcd_scene_map, a bijection for ANY grid) through one SA level + its FP twin: sampling, ball query, fused
    grouping, three_nn, interpolation and the gradients"""
    from epnet_amd import pointnet2_cuda as ext, pointnet2_utils as p2u
    rng = np.random.default_rng(17000 + case)
    m = max(1, n // int(rng.choice([2, 4, 5])))
    ns = int(rng.choice([4, 16, 32]))
    c = int(rng.choice([4, 16, 33]))
    xyz = cloud(kind, b, n, seed=16000 + case)
    d_xyz = dev(xyz)
    idx, centres = p2u.sample_and_gather(d_xyz, m)
    o_idx = oracle.furthest_point_sampling(xyz, m)
    np.testing.assert_array_equal(host(idx), o_idx)
    o_centres = np.ascontiguousarray(np.take_along_axis(xyz, o_idx[..., None].astype(np.int64), axis=1))
    np.testing.assert_array_equal(host(centres), o_centres)
    extent = float(np.ptp(xyz[0], axis=0).max()) or 1.0
    radius = extent / float(rng.choice([3.0, 10.0, 40.0]))
    feats = rng.standard_normal((b, c, n)).astype(np.float32)
    got = p2u.QueryAndGroup(radius, ns)(d_xyz, centres, dev(feats))
    bq = oracle.ball_query(radius, ns, xyz, o_centres)
    ref = np.concatenate([oracle.group_points(np.ascontiguousarray(xyz.transpose(0, 2, 1)), bq) - o_centres.transpose(0, 2, 1)[..., None],
                          oracle.group_points(feats, bq)], axis=1)
    np.testing.assert_array_equal(host(got), ref)
    go = rng.standard_normal((b, c, m, ns)).astype(np.float32)
    grad = out_tensor((b, c, n), torch.float32, 0)
    ext.group_points_grad_wrapper(b, c, n, m, ns, dev(go), dev(bq), grad)
    assert_scatter_sum(host(grad), go.reshape(b, c, -1), bq.reshape(b, -1).astype(np.int64), n)
    # the FP twin: the n points are the unknown set, the m centres the known one
    d2 = out_tensor((b, n, 3))
    nn = out_tensor((b, n, 3), torch.int32)
    ext.three_nn_wrapper(b, n, m, d_xyz, centres, d2, nn)
    o_d2, o_nn = oracle.three_nn(xyz, o_centres)
    np.testing.assert_array_equal(host(nn), o_nn)
    np.testing.assert_array_equal(host(d2), o_d2)
    safe = np.clip(o_nn, 0, m - 1)
    w = rng.random((b, n, 3)).astype(np.float32)
    w /= w.sum(-1, keepdims=True)
    known_f = rng.standard_normal((b, c, m)).astype(np.float32)
    out = out_tensor((b, c, n))
    ext.three_interpolate_wrapper(b, c, m, n, dev(known_f), dev(safe), dev(w), out)
    np.testing.assert_array_equal(host(out), oracle.three_interpolate(known_f, safe, w))
    gi = rng.standard_normal((b, c, n)).astype(np.float32)
    gk = out_tensor((b, c, m), torch.float32, 0)
    ext.three_interpolate_grad_wrapper(b, c, n, m, dev(gi), dev(safe), dev(w), gk)
    assert_scatter_sum(host(gk), (gi[:, :, :, None] * w[:, None, :, :]).reshape(b, c, n * 3), safe.reshape(b, -1).astype(np.int64), m)


def _stack_cases(count, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(count):
        n = int(rng.choice([2048, 3000, 4096, 8192, 16384, 20000]))
        levels = int(rng.integers(1, 5))
        npoints, radii, nsamples, chans, cur = [], [], [], [], n
        for lvl in range(levels):
            cur = max(8, cur // int(rng.choice([2, 4, 5])))
            k = int(rng.integers(1, 3))
            r0 = float(rng.choice([0.2, 0.5, 1.0])) * (lvl + 1)
            npoints.append(cur)
            radii.append(tuple(r0 * (j + 1) for j in range(k)))
            nsamples.append(tuple(int(rng.choice([4, 16, 32, 64])) for _ in range(k)))
            chans.append(0 if lvl == 0 and rng.integers(0, 2) else int(rng.choice([4, 16, 32, 96])))
        mode = ("plain", "two", "three")[i % 3]
        out.append((i, int(rng.integers(1, 4)), n, tuple(npoints), tuple(radii), tuple(nsamples), tuple(chans), mode))
    return out


@pytest.mark.parametrize("case,b,n,npoints,radii,nsamples,chans,mode", _stack_cases(12, 21))
def test_sweep_sa_stack_configurations(oracle, case, b, n, npoints, radii, nsamples, chans, mode):
    """the bench's product path (epnet_amd/sa_stack.py: HIP graphs, one / two / three stages) on pyramids other than the RPN's:
    1 - 4 levels, one or two scales, 2048 - 20000 points, a stream of different batches; every tensor of every scene checked
    by bench.verify_scene against the oracle"""
    import bench
    from epnet_amd import sa_stack
    kinds = ("kitti", "dup", "ubox", "kitti_q", "lattice")
    batches = [dev(cloud(kinds[(case + j) % len(kinds)], b, n, seed=18000 + 10 * case + j)) for j in range(4)]
    stack = sa_stack.SAStack(b, n=n, device=DEV, npoints=npoints, radii=radii, nsamples=nsamples, feat_channels=chans, seed=case,
                             pipelined=mode != "plain", fused_sampling=True, stages=3 if mode == "three" else None)
    stack.capture(batches[0])
    lag = {"plain": 0, "two": 1, "three": 2}[mode]
    for k in range(len(batches)):
        stack.replay(batches[k])
        torch.cuda.synchronize()
        if k < lag:
            continue
        for scene in range(b):
            bad = bench.verify_scene(stack, batches[k], scene, prev_xyz=batches[k - 1] if lag >= 1 else None,
                                     prev2_xyz=batches[k - 2] if lag >= 2 else None)
            assert bad == [], (mode, k, scene, bad)


def _linear_cases(count, seed):
    rng = np.random.default_rng(seed)
    return [(i, int(rng.integers(1, 4)), int(rng.choice([1, 3, 5, 8, 16, 33, 64, 96, 128, 130])), int(np.exp(rng.uniform(np.log(8), np.log(20000)))),
             int(rng.choice([1, 7, 64, 200, 1024])), int(rng.choice([1, 3, 4, 16, 32, 64])), bool(rng.integers(0, 2))) for i in range(count)]


@pytest.mark.parametrize("case,b,c,n,m,ns,bias", _linear_cases(40, 22))
def test_sweep_group_linear(oracle, case, b, c, n, m, ns, bias):
    """the first shared-MLP layer folded into the grouping (epnet_group_linear) and its weight gradient, against the oracle's
    restatement (exact) and a float64 evaluation of the weight gradient"""
    from epnet_amd import pointnet2_cuda as ext, pointnet2_utils as p2u
    rng = np.random.default_rng(19000 + case)
    m = min(m, n)
    xyz = cloud(KINDS[case % len(KINDS)], b, n, seed=19500 + case)
    new_xyz = np.ascontiguousarray(xyz[:, rng.permutation(n)[:m]])
    z = rng.standard_normal((b, c, n)).astype(np.float32)
    idx = rng.integers(0, n, size=(b, m, ns)).astype(np.int32)
    w_xyz = (rng.standard_normal((c, 3)) * 0.3).astype(np.float32)
    bv = rng.standard_normal((c,)).astype(np.float32) if bias else None
    got = p2u.group_linear(dev(xyz), dev(new_xyz), dev(z), dev(idx), dev(w_xyz), None if bv is None else dev(bv))
    np.testing.assert_array_equal(host(got), oracle.group_linear(xyz, new_xyz, z, idx, w_xyz, bv))
    go = rng.standard_normal((b, c, m, ns)).astype(np.float32)
    gw = out_tensor((c, 3), torch.float32, 0)
    ext.group_linear_grad_w_wrapper(b, c, n, m, ns, dev(go), dev(xyz), dev(new_xyz), dev(idx), gw)
    rel = xyz.astype(np.float64)[np.arange(b)[:, None, None], idx.astype(np.int64)] - new_xyz[:, :, None, :].astype(np.float64)   # (b, m, ns, 3)
    want = np.einsum("bcms,bmsk->ck", go.astype(np.float64), rel)
    mag = np.einsum("bcms,bmsk->ck", np.abs(go).astype(np.float64), np.abs(rel))
    assert (np.abs(host(gw) - want) <= 3e-7 * mag + 1e-5).all()


def _gather_cases(count, seed):
    rng = np.random.default_rng(seed)
    return [(i, int(rng.integers(1, 4)), int(rng.choice([1, 3, 5, 32, 64, 129, 256])), int(rng.choice([1, 2, 7, 24, 96])), int(rng.choice([1, 3, 9, 80, 320])),
             int(np.exp(rng.uniform(0, np.log(16384)))), bool(rng.integers(0, 2))) for i in range(count)]


@pytest.mark.parametrize("case,b,c,h,w,n,align", _gather_cases(40, 23))
def test_sweep_feature_gather(case, b, c, h, w, n, align):
    """the LI-Fusion point-to-pixel sampler against the op the reference calls (grid_sample, bilinear, zero padding), forward and the
    gradient w.r.t. the feature map, with the torch.gather of the coordinates over sampled indices folded in on odd cases"""
    import torch.nn.functional as F
    from epnet_amd.li_fusion import Feature_Gather
    g = torch.Generator().manual_seed(20000 + case)
    fmap = torch.randn((b, c, h, w), generator=g)
    n_src = n * 2 if case % 2 else n
    xy_src = torch.rand((b, n_src, 2), generator=g) * 2.6 - 1.3
    if case % 2:
        idx = torch.stack([torch.randperm(n_src, generator=g)[:n] for _ in range(b)]).to(torch.int32)
        xy = torch.gather(xy_src, 1, idx.long().unsqueeze(-1).expand(-1, -1, 2))
    else:
        idx, xy = None, xy_src
    want = F.grid_sample(fmap, xy.unsqueeze(1), mode="bilinear", padding_mode="zeros", align_corners=align).squeeze(2)   # the stock op, fp32, CPU
    fm = fmap.cuda().requires_grad_(True)
    got = Feature_Gather(fm, xy_src.cuda(), align_corners=align) if idx is None else Feature_Gather(fm, xy_src.cuda(), align_corners=align, idx=idx.cuda())
    if isinstance(got, tuple):
        got = got[0]
    # (as tests/test_li_fusion.py: align_corners=True agrees to 1e-5; with False the pixel coordinate carries one more rounding of a
    # value ~W between a fused and an unfused evaluation)
    tol = 1e-5 if align else 1e-4
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.numpy(), rtol=tol, atol=tol)
    gr = torch.randn((b, c, n), generator=g)
    got_g, = torch.autograd.grad(got, fm, gr.cuda())
    f64 = fmap.double().requires_grad_(True)
    ref = F.grid_sample(f64, xy.double().unsqueeze(1), mode="bilinear", padding_mode="zeros", align_corners=align).squeeze(2)
    want_g, = torch.autograd.grad(ref, f64, gr.double())
    scale = max(1.0, float(want_g.abs().max()))
    np.testing.assert_allclose(got_g.cpu().numpy(), want_g.numpy(), rtol=1e-4, atol=(1e-4 if align else 1e-3) * scale)


def _proposal_cases(count, seed):
    rng = np.random.default_rng(seed)
    # (distance-based only: the score-based path of the reference always takes the ROTATED NMS, :137, whose keep lists are held to
    # thresholds away from borderline pairs in tests/test_proposal_layer.py; the axis-aligned one below is bit-exact everywhere)
    return [(i, int(rng.integers(1, 4)), int(np.exp(rng.uniform(np.log(1), np.log(16384)))), True,
             int(rng.choice([1, 7, 64, 65, 1000, 6300, 9000])), int(rng.choice([1, 4, 100, 512, 900])), float(rng.choice([0.05, 0.5, 0.85]))) for i in range(count)]


@pytest.mark.parametrize("case,b,n,dist_based,pre,post,thresh", _proposal_cases(32, 24))
def test_sweep_rpn_proposals(oracle, case, b, n, dist_based, pre, post, thresh):
    """the fused proposal layer (score order -> distance bins -> axis-aligned NMS -> fixed-size outputs, no host sync) against the
    oracle's scene-by-scene restatement of the reference loop (lib/rpn/proposal_layer.py:40-119): identical rows, scores, counts"""
    from epnet_amd import iou3d_cuda, synth
    g = torch.Generator().manual_seed(21000 + case)
    xyz = synth.scenes("kitti", b, max(n, 2), seed=21500 + case)[:, :n]
    boxes = torch.zeros((b, n, 7))
    boxes[:, :, 0:3] = xyz + (torch.rand((b, n, 3), generator=g) - 0.5) * torch.tensor([2.0, 0.2, 2.0])
    boxes[:, :, 3:6] = torch.tensor([1.5, 1.6, 3.9]) * (0.8 + 0.4 * torch.rand((b, n, 3), generator=g))
    boxes[:, :, 6] = (torch.rand((b, n), generator=g) - 0.5) * 6
    if case % 5 == 0:
        boxes[:, :, 2] = boxes[:, :, 2] * 0.5 + 41          # the near bin empty
    scores = torch.randn((b, n), generator=g)
    order = torch.sort(scores, dim=1, descending=True)[1]
    want_b, want_s, want_c = oracle.rpn_proposals(boxes.numpy(), scores.numpy(), order.numpy(), dist_based, pre, post, thresh, False)
    rb = out_tensor((b, post, 7), torch.float32, float("nan"))
    rs = out_tensor((b, post), torch.float32, float("nan"))
    rc = out_tensor((b,), torch.int32, -1)
    iou3d_cuda.rpn_proposals_gpu(boxes.to(DEV), scores.to(DEV), order.to(DEV), dist_based, pre, post, thresh, False, rb, rs, rc)
    np.testing.assert_array_equal(host(rc), want_c)
    np.testing.assert_array_equal(host(rs), want_s)
    np.testing.assert_array_equal(host(rb), want_b)
