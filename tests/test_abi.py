"""The C-ABI library: loads without a GPU, exports every symbol include/epnet_ops.h declares, and the
Python binding table mirrors the header. No compute calls here (there is no GPU on the CPU runner)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def header_functions():
    text = open(os.path.join(ROOT, "include", "epnet_ops.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(epnet_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_hot_path():
    names = header_functions()
    for must in ["epnet_furthest_point_sampling", "epnet_gather_points", "epnet_gather_points_grad", "epnet_ball_query",
                 "epnet_group_points", "epnet_group_points_grad", "epnet_three_nn", "epnet_three_interpolate",
                 "epnet_three_interpolate_grad", "epnet_boxes_overlap_bev", "epnet_boxes_iou_bev", "epnet_nms",
                 "epnet_nms_normal", "epnet_roipool3d", "epnet_pts_in_boxes3d_host", "epnet_roipool3d_host"]:
        assert must in names


def test_library_exports_every_declared_symbol(hiplib):
    from epnet_amd import _lib
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_functions():
        assert hasattr(raw, name), "libepnet_hip.so lacks %s" % name
    assert sorted(_lib.SIGNATURES) == header_functions()
    assert hiplib.epnet_abi_version() == 1
    assert hiplib.epnet_strerror(0) == b"ok" and b"workspace" in hiplib.epnet_strerror(-3)


def test_argument_validation_without_gpu(hiplib):
    # negative sizes / NULL pointers are rejected before anything touches the device
    assert hiplib.epnet_ball_query(-1, 1, 1, 1.0, 1, None, None, None, None) == -1
    assert hiplib.epnet_furthest_point_sampling(1, 16, 4, None, None, None, None) == -1
    assert hiplib.epnet_group_points(1, 1, 1, 1, 1, None, None, None, None) == -1
    assert hiplib.epnet_nms_workspace_bytes(6300) == 6300 * 99 * 8 + 6300 * 80   # mask words + one 80-byte record per box
    assert hiplib.epnet_nms_workspace_bytes(0) == 0
    # positions to fill from an EMPTY cloud (n = 0): an error code, not a division by zero on the host
    import ctypes as C
    fake = C.c_void_p(4096)   # never dereferenced: the call must return before anything touches the device
    assert hiplib.epnet_group_points(1, 16, 0, 512, 8, fake, fake, fake, None) == -1
    assert hiplib.epnet_gather_points(1, 16, 0, 4096, fake, fake, fake, None) == -1
    assert hiplib.epnet_group_concat(1, 16, 0, 512, 8, fake, fake, fake, fake, fake, 1, None) == -1
    two = (C.c_int * 2)(8, 8)
    ptrs = (C.c_void_p * 2)(4096, 4096)
    assert hiplib.epnet_group_concat_multi(1, 16, 0, 512, 2, C.cast(two, C.c_void_p), fake, fake, fake, C.cast(ptrs, C.c_void_p),
                                           C.cast(ptrs, C.c_void_p), 1, None) == -1
    # empty problems are no-ops
    assert hiplib.epnet_ball_query(0, 10, 10, 1.0, 4, None, None, None, None) == 0
    assert hiplib.epnet_three_nn(1, 0, 5, None, None, None, None, None) == 0


def test_missing_library_is_loud(monkeypatch, tmp_path):
    from epnet_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


def test_cpu_tensors_are_rejected(hiplib):
    import torch
    from epnet_amd import pointnet2_utils, iou3d_utils
    xyz = torch.zeros((1, 8, 3))
    with pytest.raises(RuntimeError, match="CUDAtensor"):
        pointnet2_utils.ball_query(1.0, 4, xyz, xyz)
    with pytest.raises(RuntimeError, match="CUDAtensor"):
        iou3d_utils.boxes_iou_bev(torch.zeros((2, 5)), torch.zeros((2, 5)))


def test_host_ops_match_oracle_and_fixture(hiplib, oracle):
    """roipool3d's CPU entry points are host ops in the reference too (roipool3d.cpp:97-195)"""
    import numpy as np
    import torch
    from conftest import golden
    from epnet_amd import roipool3d_utils
    fx = golden("roipool3d_ref.npz")
    n = fx["pts"].shape[0]
    masks = roipool3d_utils.pts_in_boxes3d_cpu(torch.from_numpy(fx["pts"]), torch.from_numpy(fx["boxes3d"]))
    flags = np.unpackbits(fx["pts_flag"], axis=1)[:, :n].astype(bool)
    np.testing.assert_array_equal(torch.stack(masks).numpy(), flags)
    pp, pf, ef = roipool3d_utils.roipool_pc_cpu(torch.from_numpy(fx["pts"]), torch.from_numpy(fx["pts_feature"]),
                                               torch.from_numpy(fx["boxes3d"]), fx["pooled_pts"].shape[1])
    np.testing.assert_array_equal(pp.numpy(), fx["pooled_pts"])
    np.testing.assert_array_equal(pf.numpy(), fx["pooled_features"])
    np.testing.assert_array_equal(ef.numpy(), fx["pooled_empty_flag"])


def test_scene_index_size_is_the_documented_layout(hiplib):
    """epnet_scene_index_bytes (no GPU needed): per scene np float4 rows (np = the power of two >= max(n, 2048)), one box of 6
    floats per 64 and per 256 rows and, beyond 16384 points, np floats of sampling scratch (include/epnet_ops.h) -- a caller
    that sized the buffer by the formula of an older header would get EPNET_ENOMEM, so the formula is pinned here"""
    def want(b, n):
        if n < 1024 or n > 65536:
            return 0
        np_ = 2048
        while np_ < n:
            np_ *= 2
        return b * (np_ * 16 + (np_ // 64 + np_ // 256) * 24 + (np_ * 4 if n > 16384 else 0))
    for b, n in ((1, 1024), (3, 4096), (2, 16384), (2, 16385), (1, 40000), (4, 65536), (1, 1023), (1, 65537), (0, 4096)):
        assert hiplib.epnet_scene_index_bytes(b, n) == (want(b, n) if b > 0 else 0), (b, n)
