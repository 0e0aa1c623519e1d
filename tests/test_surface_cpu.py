"""Host-side logic on the CPU runner: epnet_amd's Python operator surface must reproduce what the
REFERENCE's Python surface produced (fixtures from tests/golden/make_golden.py). The kernels are not under
test here -- the three extension stand-ins are swapped for oracle-backed ones (tests/oracle_ext.py) so the
composition code (argument orders, transposes, centre subtraction, channel order, FP weights, sort/index
conventions, state_dict names, autograd wiring) runs without a GPU."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden


@pytest.fixture()
def surface(monkeypatch, oracle):
    import oracle_ext
    from epnet_amd import iou3d_cuda, pointnet2_cuda, roipool3d_cuda
    p2, iou, rp = oracle_ext.make_modules()
    for real, fake in ((pointnet2_cuda, p2), (iou3d_cuda, iou), (roipool3d_cuda, rp)):
        for name, fn in vars(fake).items():
            if callable(fn):
                monkeypatch.setattr(real, name, fn)
    from epnet_amd import iou3d_utils, kitti_utils, pointnet2_modules, pointnet2_utils, roipool3d_utils
    return dict(p2u=pointnet2_utils, p2m=pointnet2_modules, iou=iou3d_utils, rp=roipool3d_utils, ku=kitti_utils)


def load_state(module, fx):
    sd = {k[4:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("sd__")}
    assert sorted(sd) == sorted(module.state_dict().keys())
    module.load_state_dict(sd)
    return module.eval()


def test_state_dict_names_match_reference():
    from epnet_amd import pointnet2_modules as p2m, pytorch_utils as ptu
    ref = json.load(open(os.path.join(GOLDEN, "state_dict_names.json")))
    mine = {
        "SharedMLP_bn": ptu.SharedMLP([4, 8, 16], bn=True),
        "SharedMLP_plain": ptu.SharedMLP([4, 8], bn=False),
        "SharedMLP_preact_first": ptu.SharedMLP([4, 8, 16], bn=True, preact=True, first=True),
        "SharedMLP_instance_norm": ptu.SharedMLP([4, 8], bn=False, instance_norm=True),
        "Conv1d_bn": ptu.Conv1d(4, 8, bn=True),
        "Conv1d_noact": ptu.Conv1d(4, 1, activation=None),
        "Conv2d_named": ptu.Conv2d(4, 8, bn=True, name="x_"),
        "FC_bn": ptu.FC(4, 8, bn=True),
        "SA_MSG": p2m.PointnetSAModuleMSG(npoint=256, radii=[0.5, 1.0], nsamples=[16, 32], mlps=[[16, 16, 32], [16, 16, 32]]),
        "FP": p2m.PointnetFPModule(mlp=[48, 32]),
    }
    for key, mod in mine.items():
        assert list(mod.state_dict().keys()) == ref[key], key


def test_sa_module_mutates_mlp_spec_like_reference():
    from epnet_amd import pointnet2_modules as p2m
    spec = [[16, 32]]
    p2m.PointnetSAModuleMSG(npoint=8, radii=[1.0], nsamples=[4], mlps=spec, use_xyz=True)
    assert spec == [[19, 32]]  # reference pointnet2_modules.py:105-106


def test_cfg1_through_own_surface(surface):
    fx, p2u = golden("pointnet2_cfg1.npz"), surface["p2u"]
    xyz = torch.from_numpy(fx["xyz"])
    idx = p2u.furthest_point_sample(xyz, 1024)
    assert idx.dtype == torch.int32
    np.testing.assert_array_equal(idx.numpy(), fx["fps_idx"])
    new_xyz = p2u.gather_operation(xyz.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
    np.testing.assert_array_equal(new_xyz.numpy(), fx["new_xyz"])
    np.testing.assert_array_equal(p2u.ball_query(0.1, 32, xyz, new_xyz).numpy(), fx["ball_idx_r01"])
    out = p2u.QueryAndGroup(2.0, 32, use_xyz=True)(xyz, new_xyz, None)
    np.testing.assert_array_equal(out.numpy(), fx["query_and_group_r20"])


def test_sa_module_fixture(surface):
    fx = golden("sa_module.npz")
    sa = load_state(surface["p2m"].PointnetSAModuleMSG(npoint=256, radii=[0.5, 1.0], nsamples=[16, 32],
                                                       mlps=[[16, 16, 32], [16, 16, 32]], use_xyz=True, bn=True), fx)
    with torch.no_grad():
        new_xyz, feats, idx = sa(torch.from_numpy(fx["xyz"]), torch.from_numpy(fx["features"]))
    np.testing.assert_array_equal(idx.numpy(), fx["idx"])
    np.testing.assert_array_equal(new_xyz.numpy(), fx["new_xyz"])
    np.testing.assert_allclose(feats.numpy(), fx["out_features"], atol=1e-5, rtol=0)
    # caller-supplied new_xyz: no FPS, idx is None (reference :46-47)
    with torch.no_grad():
        nx2, f2, i2 = sa(torch.from_numpy(fx["xyz"]), torch.from_numpy(fx["features"]), new_xyz=new_xyz)
    assert i2 is None and torch.equal(f2, feats)


def test_groupall_fixture(surface):
    fx, sm = golden("sa_groupall.npz"), golden("sa_module.npz")
    ga = load_state(surface["p2m"].PointnetSAModule(mlp=[16, 32], npoint=None, radius=None, nsample=None, use_xyz=True), fx)
    with torch.no_grad():
        new_xyz, feats, idx = ga(torch.from_numpy(sm["xyz"][:, :128].copy()), torch.from_numpy(sm["features"][:, :, :128].copy()))
    assert new_xyz is None and idx is None
    np.testing.assert_allclose(feats.numpy(), fx["out_features"], atol=1e-5, rtol=0)


def test_fp_module_fixture(surface):
    fx = golden("fp_module.npz")
    fpm = load_state(surface["p2m"].PointnetFPModule(mlp=[48, 32]), fx)
    unknown, known = torch.from_numpy(fx["unknown"]), torch.from_numpy(fx["known"])
    dist, nn_idx = surface["p2u"].three_nn(unknown, known)
    np.testing.assert_array_equal(nn_idx.numpy(), fx["nn_idx"])
    np.testing.assert_array_equal(dist.numpy(), fx["dist"])
    with torch.no_grad():
        out = fpm(unknown, known, torch.from_numpy(fx["unknow_feats"]), torch.from_numpy(fx["known_feats"]))
    np.testing.assert_allclose(out.numpy(), fx["out"], atol=1e-5, rtol=0)
    with torch.no_grad():  # known=None branch: broadcast of a single global feature
        o2 = fpm(unknown, None, torch.from_numpy(fx["unknow_feats"]), torch.from_numpy(fx["known_feats"][:, :, :1].copy()))
    assert o2.shape == out.shape


def test_autograd_wiring(surface):
    fx, p2u = golden("grads.npz"), surface["p2u"]
    feat = torch.from_numpy(fx["feat"]).requires_grad_(True)
    grouped = p2u.grouping_operation(feat, torch.from_numpy(fx["idx"]))
    (grouped * torch.from_numpy(fx["upstream"])).sum().backward()
    np.testing.assert_allclose(feat.grad.numpy(), fx["grad_feat"], atol=1e-5, rtol=0)
    kf = torch.from_numpy(fx["known_feats"]).requires_grad_(True)
    interp = p2u.three_interpolate(kf, torch.from_numpy(fx["nn_idx"]), torch.from_numpy(fx["weight"]))
    (interp * torch.from_numpy(fx["upstream2"])).sum().backward()
    np.testing.assert_allclose(kf.grad.numpy(), fx["grad_known"], atol=1e-5, rtol=0)
    g = torch.from_numpy(fx["feat"]).requires_grad_(True)
    gathered = p2u.gather_operation(g, torch.from_numpy(fx["idx"][:, :, 0].copy()))
    gathered.sum().backward()
    assert g.grad.sum().item() == pytest.approx(gathered.numel())


def test_iou3d_fixture_through_own_surface(surface):
    fx, iou, ku = golden("iou3d.npz"), surface["iou"], surface["ku"]
    boxes_a, boxes_b, scores = (torch.from_numpy(fx[k]) for k in ("boxes_a", "boxes_b", "scores"))
    np.testing.assert_array_equal(ku.boxes3d_to_bev_torch(boxes_a).numpy(), fx["bev_a"])
    np.testing.assert_allclose(iou.boxes_iou3d_gpu(boxes_a, boxes_b).numpy(), fx["iou3d"], atol=1e-6, rtol=0)
    bev_a, bev_b = ku.boxes3d_to_bev_torch(boxes_a), ku.boxes3d_to_bev_torch(boxes_b)
    np.testing.assert_array_equal(iou.boxes_iou_bev(bev_a, bev_b).numpy(), fx["iou_bev"])
    for key, thr, fn in (("keep_rot_010", 0.1, iou.nms_gpu), ("keep_rot_050", 0.5, iou.nms_gpu),
                         ("keep_normal_085", 0.85, iou.nms_normal_gpu), ("keep_normal_050", 0.5, iou.nms_normal_gpu)):
        kept = fn(bev_a, scores, thr)
        assert kept.dtype == torch.int64
        np.testing.assert_array_equal(kept.numpy(), fx[key])


def test_roipool3d_fixture_through_own_surface(surface):
    fx = golden("roipool3d_surface.npz")
    pooled, empty = surface["rp"].roipool3d_gpu(torch.from_numpy(fx["pts"]), torch.from_numpy(fx["pts_feature"]),
                                                torch.from_numpy(fx["boxes3d"]), 0.2, sampled_pt_num=64)
    assert empty.dtype == torch.int32
    np.testing.assert_array_equal(empty.numpy(), fx["pooled_empty_flag"])
    np.testing.assert_array_equal(pooled.numpy(), fx["pooled_features"])


def test_kitti_helpers():
    from epnet_amd import kitti_utils as ku
    b = torch.tensor([[1.0, 2.0, 3.0, 1.5, 1.6, 3.9, 0.3]])
    assert torch.allclose(ku.enlarge_box3d(b, 0.2), torch.tensor([[1.0, 2.2, 3.0, 1.9, 2.0, 4.3, 0.3]]))
    assert np.allclose(ku.enlarge_box3d(b.numpy(), 0.2), [[1.0, 2.2, 3.0, 1.9, 2.0, 4.3, 0.3]])
    assert torch.allclose(ku.boxes3d_to_bev_torch(b), torch.tensor([[1 - 1.95, 3 - 0.8, 1 + 1.95, 3 + 0.8, 0.3]]))
    pc = torch.tensor([[[1.0, 5.0, 0.0, 7.0]]])
    out = ku.rotate_pc_along_y_torch(pc.clone(), torch.tensor([np.pi / 2]))
    assert torch.allclose(out, torch.tensor([[[0.0, 5.0, 1.0, 7.0]]]), atol=1e-6)  # x -> z, y and extras untouched


def test_compat_paths(hiplib):
    import importlib
    import sys
    from epnet_amd import compat
    saved = dict(sys.modules)
    try:
        compat.install_surface(include_kitti_utils=True)
        import pointnet2_cuda
        import iou3d_cuda
        import roipool3d_cuda
        assert pointnet2_cuda.__name__ == "epnet_amd.pointnet2_cuda"
        for fn in ("ball_query_wrapper", "group_points_wrapper", "group_points_grad_wrapper", "gather_points_wrapper",
                   "gather_points_grad_wrapper", "furthest_point_sampling_wrapper", "three_nn_wrapper",
                   "three_interpolate_wrapper", "three_interpolate_grad_wrapper"):
            assert callable(getattr(pointnet2_cuda, fn))          # pointnet2_api.cpp:10-24
        for fn in ("boxes_overlap_bev_gpu", "boxes_iou_bev_gpu", "nms_gpu", "nms_normal_gpu"):
            assert callable(getattr(iou3d_cuda, fn))              # iou3d.cpp:174-179
        for fn in ("pts_in_boxes3d_cpu", "roipool3d_cpu", "forward", "forward_slow"):
            assert callable(getattr(roipool3d_cuda, fn))          # roipool3d.cpp:198-203
        from pointnet2_lib.pointnet2.pointnet2_modules import PointnetFPModule, PointnetSAModuleMSG  # lib/net/pointnet2_msg.py:4
        import pointnet2_lib.pointnet2.pytorch_utils as pt_utils                                       # lib/net/rpn.py:5
        import lib.utils.iou3d.iou3d_utils as iou3d_utils
        import lib.utils.roipool3d.roipool3d_utils as roipool3d_utils
        assert hasattr(pt_utils, "SharedMLP") and hasattr(iou3d_utils, "boxes_iou3d_gpu") and hasattr(roipool3d_utils, "roipool3d_gpu")
        assert PointnetSAModuleMSG and PointnetFPModule
    finally:
        for k in list(sys.modules):
            if k not in saved:
                del sys.modules[k]


def test_synth_and_byte_model():
    from epnet_amd import sa_stack, synth
    a, b = synth.kitti_like_cloud(4096, 3), synth.kitti_like_cloud(4096, 3)
    assert torch.equal(a, b) and not torch.equal(a, synth.kitti_like_cloud(4096, 4))
    assert a.dtype == torch.float32 and a.is_contiguous() and a.shape == (4096, 3)
    for d, (lo, hi) in enumerate(synth.SCOPE):
        assert a[:, d].min() >= lo and a[:, d].max() <= hi
    assert len(torch.unique(synth.dup_cloud(2048, 1, unique=1500), dim=0)) == 1500
    by = sa_stack.sa_algorithmic_bytes()
    assert by["total"] == 51326720 and by["group_feat"] == 44298240 and by["fps"] == 282880
    assert sa_stack.fp_algorithmic_bytes()["total"] == 88087040 - 51326720
