"""epnet_ball_query_ordered: the centres served in the order of their own scene index (neighbouring waves, neighbouring centres), against the oracle's
ball_query (the reference's one-thread-per-centre scan, ball_query_gpu.cu:9-45) -- indices bit-exact, every slot written.
"""
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


def _cloud(kind, b, n, seed):
    from epnet_amd import synth
    return synth.scenes(kind, b, n, seed=seed).numpy()


def _centres(oracle, xyz, m, how, seed):
    b, n = xyz.shape[:2]
    rng = np.random.default_rng(seed)
    if how == "fps":          # what an SA level feeds it: the sampled points, in sampling order
        idx = oracle.furthest_point_sampling(xyz, m)
        return np.ascontiguousarray(np.take_along_axis(xyz, idx[..., None].astype(np.int64), axis=1))
    if how == "subset":
        return np.ascontiguousarray(np.stack([xyz[s][rng.permutation(n)[:m]] for s in range(b)]))
    if how == "free":         # centres that are no points of the cloud, some far outside it
        c = np.stack([xyz[s][rng.integers(0, n, size=m)] for s in range(b)]) + rng.normal(0, 0.3, size=(b, m, 3)).astype(np.float32)
        c[:, ::17] += 500.0
        return np.ascontiguousarray(c.astype(np.float32))
    raise ValueError(how)


@pytest.fixture(autouse=True)
def _force_the_ordered_kernels(monkeypatch):
    monkeypatch.setenv("EPNET_BQ_ORDERED", "1")     # (by itself the library takes the order from 32768 points up only)


def _run(b, n, m, scales, xyz, centres):
    from epnet_amd import pointnet2_cuda as ext
    d_xyz, d_c = dev(xyz), dev(centres)
    index = ext.scene_index(d_xyz)
    centre_index = ext.scene_index(d_c)
    assert index is not None and centre_index is not None
    outs = [torch.full((b, m, ns), -5, dtype=torch.int32, device=DEV) for _r, ns in scales]
    ext.ball_query_ordered_wrapper(b, n, m, [r for r, _ in scales], [ns for _, ns in scales], d_c, d_xyz, index, centre_index, outs)
    return outs


@pytest.mark.parametrize("b,n,m,scales,kind,how", [
    (2, 16384, 4096, ((0.1, 16), (0.5, 32)), "kitti", "fps"),        # level 1 of the RPN pyramid
    (2, 4096, 1024, ((0.5, 16), (1.0, 32)), "kitti", "fps"),         # level 2
    (2, 16384, 4096, ((0.5, 32), (0.1, 16)), "dup", "fps"),          # radii the other way round; exact twins among the points
    (1, 16384, 1500, ((2.5, 64), (0.3, 8)), "kitti", "subset"),      # most balls crowded (> 64 hits): the bitmap path; odd centre count
    (2, 5000, 1100, ((0.7, 5), (0.7, 9)), "ubox", "subset"),         # equal radii; n not a power of two
    (1, 3000, 1024, ((5.0, 100), (0.5, 3)), "ubox", "free"),         # nsample beyond the 64-entry list
    (2, 2048, 1024, ((1.0, 8),), "kitti", "free"),                   # one scale; the smallest index
    (1, 40000, 2048, ((0.3, 8), (1.0, 64)), "kitti", "fps"),         # two scales on a big scene
    (1, 65536, 3000, ((0.5, 64),), "kitti", "fps"),                  # config-5 shape (fewer centres: the oracle scans n per centre)
    (1, 65536, 1024, ((4.0, 64),), "ubox", "subset"),
])
def test_ordered_ball_query_matches_oracle(oracle, b, n, m, scales, kind, how):
    xyz = _cloud(kind, b, n, seed=300 + n + m)
    centres = _centres(oracle, xyz, m, how, seed=n + m)
    outs = _run(b, n, m, scales, xyz, centres)
    for (r, ns), got in zip(scales, outs):
        np.testing.assert_array_equal(host(got), oracle.ball_query(r, ns, xyz, centres), err_msg="r=%g ns=%d" % (r, ns))


def test_ordered_ball_query_non_finite_centres_and_points(oracle):
    """a centre with a NaN / inf coordinate has an empty ball (all zeros, ball_query_gpu.cu:29-43 never finds a hit) and must not
    widen the bucket's box into skipping its neighbours' points; NaN points are nobody's neighbour"""
    b, n, m = 2, 16384, 4096
    xyz = _cloud("kitti", b, n, seed=9)
    centres = _centres(oracle, xyz, m, "fps", seed=9)
    bad = np.array([np.nan, np.inf, -np.inf], np.float32)
    rng = np.random.default_rng(3)
    for k in range(40):
        centres[rng.integers(0, b), rng.integers(0, m), rng.integers(0, 3)] = bad[k % 3]
        xyz[rng.integers(0, b), rng.integers(0, n), rng.integers(0, 3)] = bad[(k + 1) % 3]
    scales = ((0.1, 16), (0.5, 32))
    outs = _run(b, n, m, scales, xyz, centres)
    for (r, ns), got in zip(scales, outs):
        np.testing.assert_array_equal(host(got), oracle.ball_query(r, ns, xyz, centres))


def test_ordered_equals_the_per_centre_kernels_at_full_size():
    """BASELINE config 5 at full size (65536 points, 16384 centres, r 0.5, nsample 64) and the level-1 shape at 16 scenes: the ordered
    launch against the launch in centre order of the same library (itself held to the oracle in test_gpu_parity.py)"""
    from epnet_amd import pointnet2_cuda as ext, synth
    for b, n, m, scales in ((2, 65536, 16384, ((0.5, 64),)), (16, 16384, 4096, ((0.1, 16), (0.5, 32)))):
        xyz = synth.scenes("kitti", b, n, seed=41).to(DEV)
        index = ext.scene_index(xyz)
        fidx = torch.empty((b, m), dtype=torch.int32, device=DEV)
        centres = torch.empty((b, m, 3), device=DEV)
        ext.sample_centres_wrapper(b, n, m, xyz, index, fidx, centres)
        want = [torch.full((b, m, ns), -5, dtype=torch.int32, device=DEV) for _r, ns in scales]
        got = [torch.full((b, m, ns), -6, dtype=torch.int32, device=DEV) for _r, ns in scales]
        radii, nss = [r for r, _ in scales], [ns for _, ns in scales]
        ext.ball_query_multi_wrapper(b, n, m, radii, nss, centres, xyz, index, want)
        ext.ball_query_ordered_wrapper(b, n, m, radii, nss, centres, xyz, index, ext.scene_index(centres), got)
        for w, g in zip(want, got):
            assert torch.equal(w, g)


def test_ordered_falls_back_without_a_centre_index(oracle):
    from epnet_amd import pointnet2_cuda as ext
    b, n, m = 1, 4096, 200          # 200 centres: below the indexed range
    xyz = _cloud("kitti", b, n, seed=5)
    centres = _centres(oracle, xyz, m, "subset", seed=5)
    d_xyz, d_c = dev(xyz), dev(centres)
    out = [torch.full((b, m, 16), -5, dtype=torch.int32, device=DEV)]
    ext.ball_query_ordered_wrapper(b, n, m, [0.8], [16], d_c, d_xyz, ext.scene_index(d_xyz), None, out)
    np.testing.assert_array_equal(host(out[0]), oracle.ball_query(0.8, 16, xyz, centres))


def test_ordered_entry_point_contract(oracle):
    """the C entry point itself: empty launches are no-ops, an undersized index is refused (as the other indexed entry points
    do), three scales fall back to one launch per scale, NULL centre index = the plain indexed query"""
    import ctypes
    from epnet_amd import _lib, pointnet2_cuda as ext
    lib = _lib.lib()
    b, n, m, ns = 1, 32768, 1024, 16
    xyz = _cloud("kitti", b, n, seed=1)
    centres = _centres(oracle, xyz, m, "subset", seed=1)
    d_xyz, d_c = dev(xyz), dev(centres)
    index, cindex = ext.scene_index(d_xyz), ext.scene_index(d_c)
    out = torch.full((b, m, ns), -5, dtype=torch.int32, device=DEV)
    radii = (ctypes.c_float * 1)(0.6)
    nss = (ctypes.c_int * 1)(ns)
    ptrs = (ctypes.c_void_p * 1)(out.data_ptr())
    s = torch.cuda.current_stream().cuda_stream

    def call(bb, mm, ix_bytes, cix_ptr, cix_bytes):
        return lib.epnet_ball_query_ordered(bb, n, mm, 1, ctypes.cast(radii, ctypes.c_void_p), ctypes.cast(nss, ctypes.c_void_p),
                                            d_c.data_ptr(), d_xyz.data_ptr(), index.data_ptr(), ix_bytes, cix_ptr, cix_bytes,
                                            ctypes.cast(ptrs, ctypes.c_void_p), s)
    assert call(0, m, index.numel(), cindex.data_ptr(), cindex.numel()) == 0          # no scenes
    assert call(b, 0, index.numel(), cindex.data_ptr(), cindex.numel()) == 0          # no centres
    torch.cuda.synchronize()
    assert (out == -5).all()
    assert call(b, m, index.numel() - 16, cindex.data_ptr(), cindex.numel()) != 0     # undersized point index
    assert call(b, m, index.numel(), cindex.data_ptr(), cindex.numel() - 16) != 0     # undersized centre index
    assert call(b, m, index.numel(), None, 0) == 0                                    # no centre index: the plain indexed query
    torch.cuda.synchronize()
    want = oracle.ball_query(0.6, ns, xyz, centres)
    np.testing.assert_array_equal(host(out), want)
    out.fill_(-5)
    assert call(b, m, index.numel(), cindex.data_ptr(), cindex.numel()) == 0
    torch.cuda.synchronize()
    np.testing.assert_array_equal(host(out), want)
    # three scales: one launch per scale through the same entry point
    outs = [torch.full((b, m, k), -5, dtype=torch.int32, device=DEV) for k in (4, 8, 16)]
    ext.ball_query_ordered_wrapper(b, n, m, [0.3, 0.6, 1.2], [4, 8, 16], d_c, d_xyz, index, cindex, outs)
    for r, k, got in zip((0.3, 0.6, 1.2), (4, 8, 16), outs):
        np.testing.assert_array_equal(host(got), oracle.ball_query(r, k, xyz, centres))
