"""Regenerates the committed fixtures under tests/golden/. Runs ONLY in the build container
(needs /root/reference); the fixtures themselves are plain data and travel everywhere.

Two kinds of fixture:

(i)  ``roipool3d_ref.npz`` -- outputs of the REFERENCE'S OWN compiled CPU ops
     (lib/utils/roipool3d/src/roipool3d.cpp:97-195, built unmodified by oracle/build_ref.py):
     a true reference pin of the point-in-box predicate and the first-S / cyclic-pad / empty-flag
     pooling semantics.

(iii) everything else -- the REFERENCE'S OWN PYTHON SURFACE (pointnet2_utils / pointnet2_modules /
     iou3d_utils / roipool3d_utils, imported unmodified from /root/reference) executed on the CPU
     on top of extension stand-ins backed by the oracle. The reference has no CPU kernels for
     these ops and nvcc is absent, so the arithmetic inside comes from oracle/epnet_oracle.c; what
     these fixtures pin is the Python-level composition (argument orders, transposes, centre
     subtraction, channel order, weight formula, sort/index conventions, state_dict names) that
     epnet_amd's own surface must reproduce.

To run the reference surface on CPU tensors the script aliases torch.cuda.FloatTensor/IntTensor to
the CPU constructors and makes Tensor.cuda() the identity (process-local monkeypatches).
"""
import json
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True  # importing the reference must not leave __pycache__ files in /root/reference

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import build_ref, oracle  # noqa: E402
from epnet_amd import synth  # noqa: E402
sys.path.insert(0, os.path.dirname(HERE))
import oracle_ext  # noqa: E402

REF = "/root/reference"
NMS_SEED = 22


def _np(t):
    return t.detach().cpu().numpy()


def import_reference_surface():
    torch.cuda.FloatTensor = torch.FloatTensor
    torch.cuda.IntTensor = torch.IntTensor
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)
    import pointnet2_lib.pointnet2.pointnet2_utils as p2u
    import pointnet2_lib.pointnet2.pointnet2_modules as p2m
    import pointnet2_lib.pointnet2.pytorch_utils as ptu
    import lib.utils.iou3d.iou3d_utils as iou_u
    import lib.utils.roipool3d.roipool3d_utils as rp_u
    import lib.utils.kitti_utils as k_u
    return p2u, p2m, ptu, iou_u, rp_u, k_u


def state_arrays(module):
    return {"sd__" + k: _np(v) for k, v in module.state_dict().items()}


def main():
    assert os.path.isdir(REF), "needs the reference checkout"
    oracle_ext.install_as_top_level()
    p2u, p2m, ptu, iou_u, rp_u, k_u = import_reference_surface()
    out = {}

    # ---- BASELINE config 1: 4096-pt U-box cloud, 1 SA level (npoint 1024, nsample 32, r 0.1 and 2.0)
    xyz = synth.ubox_cloud(4096, 0).unsqueeze(0)
    idx = p2u.furthest_point_sample(xyz, 1024)
    new_xyz = p2u.gather_operation(xyz.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
    bq_small = p2u.ball_query(0.1, 32, xyz, new_xyz)
    bq_large = p2u.ball_query(2.0, 32, xyz, new_xyz)
    qg = p2u.QueryAndGroup(2.0, 32, use_xyz=True)(xyz, new_xyz, None)
    np.savez_compressed(os.path.join(HERE, "pointnet2_cfg1.npz"), xyz=_np(xyz), fps_idx=_np(idx), new_xyz=_np(new_xyz),
                        ball_idx_r01=_np(bq_small), ball_idx_r20=_np(bq_large), query_and_group_r20=_np(qg))
    out["cfg1"] = dict(fps_head=_np(idx)[0, :8].tolist(), hits_r01=int((_np(bq_small) != 0).sum()))

    # ---- FPS tie-breaks: duplicated rows (dataset padding) and a non-power-of-two cloud
    dup = synth.dup_cloud(2048, 3, unique=1200).unsqueeze(0)
    odd = synth.kitti_like_cloud(1000, 5).unsqueeze(0)
    np.savez_compressed(os.path.join(HERE, "fps_ties.npz"), dup_xyz=_np(dup), dup_idx=_np(p2u.furthest_point_sample(dup, 1500)),
                        odd_xyz=_np(odd), odd_idx=_np(p2u.furthest_point_sample(odd, 300)))

    # ---- one MSG SA module with level-2-like shapes (scaled down), seeded weights, eval mode
    torch.manual_seed(1234)
    sa = p2m.PointnetSAModuleMSG(npoint=256, radii=[0.5, 1.0], nsamples=[16, 32], mlps=[[16, 16, 32], [16, 16, 32]],
                                 use_xyz=True, bn=True).eval()
    sxyz = synth.scenes("kitti", 2, 1024, seed=11)
    sfeat = torch.randn((2, 16, 1024), generator=torch.Generator().manual_seed(12))
    with torch.no_grad():
        s_new_xyz, s_feat, s_idx = sa(sxyz, sfeat)
    np.savez_compressed(os.path.join(HERE, "sa_module.npz"), xyz=_np(sxyz), features=_np(sfeat), new_xyz=_np(s_new_xyz),
                        out_features=_np(s_feat), idx=_np(s_idx), **state_arrays(sa))

    # ---- single-scale SA module with GroupAll (RCNN head's last layer, lib/net/rcnn_net.py:36-44)
    torch.manual_seed(99)
    ga = p2m.PointnetSAModule(mlp=[16, 32], npoint=None, radius=None, nsample=None, use_xyz=True, bn=True).eval()
    with torch.no_grad():
        g_new_xyz, g_feat, g_idx = ga(sxyz[:, :128].contiguous(), sfeat[:, :, :128].contiguous())
    assert g_new_xyz is None and g_idx is None
    np.savez_compressed(os.path.join(HERE, "sa_groupall.npz"), out_features=_np(g_feat), **state_arrays(ga))

    # ---- FP module
    torch.manual_seed(4321)
    fpm = p2m.PointnetFPModule(mlp=[32 + 16, 32]).eval()
    known_feats = torch.randn((2, 32, 256), generator=torch.Generator().manual_seed(13))
    with torch.no_grad():
        dist, nn_idx = p2u.three_nn(sxyz, s_new_xyz)
        fp_out = fpm(sxyz, s_new_xyz, sfeat, known_feats)
    np.savez_compressed(os.path.join(HERE, "fp_module.npz"), unknown=_np(sxyz), known=_np(s_new_xyz), unknow_feats=_np(sfeat),
                        known_feats=_np(known_feats), dist=_np(dist), nn_idx=_np(nn_idx), out=_np(fp_out),
                        **state_arrays(fpm))

    # ---- autograd through grouping / gather / three_interpolate (backward wrappers)
    gfeat = torch.randn((2, 8, 1024), generator=torch.Generator().manual_seed(14), requires_grad=True)
    gidx = p2u.ball_query(1.0, 16, sxyz, s_new_xyz)
    grouped = p2u.grouping_operation(gfeat, gidx)
    wgt = torch.randn(grouped.shape, generator=torch.Generator().manual_seed(15))
    (grouped * wgt).sum().backward()
    kf = known_feats.clone().requires_grad_(True)
    inv = 1.0 / (dist + 1e-8)
    w3 = inv / inv.sum(dim=2, keepdim=True)
    interp = p2u.three_interpolate(kf, nn_idx, w3)
    wgt2 = torch.randn(interp.shape, generator=torch.Generator().manual_seed(16))
    (interp * wgt2).sum().backward()
    np.savez_compressed(os.path.join(HERE, "grads.npz"), feat=_np(gfeat), idx=_np(gidx), upstream=_np(wgt), grad_feat=_np(gfeat.grad),
                        known_feats=_np(kf), nn_idx=_np(nn_idx), weight=_np(w3), upstream2=_np(wgt2), grad_known=_np(kf.grad))

    # ---- iou3d: 3-D IoU matrix, BEV IoU, rotated and axis-aligned NMS
    # seed chosen by search_nms_seed(): no rotated IoU of the fixture lies within 1e-5 of a threshold
    # used below, so that ulp-level libm differences cannot flip a keep decision
    boxes_a, scores = synth.proposal_boxes(300, seed=NMS_SEED, num_objects=12, jitter=0.8)
    boxes_b, _ = synth.proposal_boxes(24, seed=22, num_objects=12, jitter=0.3)
    iou3d = iou_u.boxes_iou3d_gpu(boxes_a, boxes_b)
    bev_a, bev_b = k_u.boxes3d_to_bev_torch(boxes_a), k_u.boxes3d_to_bev_torch(boxes_b)
    iou_bev = iou_u.boxes_iou_bev(bev_a, bev_b)
    fx = dict(boxes_a=_np(boxes_a), boxes_b=_np(boxes_b), scores=_np(scores), bev_a=_np(bev_a), iou3d=_np(iou3d),
              iou_bev=_np(iou_bev))
    order = scores.sort(0, descending=True)[1]
    full = oracle.boxes_iou_bev(_np(bev_a[order]), _np(bev_a[order]))
    for name, thr in (("rot", 0.1), ("rot", 0.5), ("normal", 0.85), ("normal", 0.5)):
        fn = iou_u.nms_gpu if name == "rot" else iou_u.nms_normal_gpu
        fx["keep_%s_%03d" % (name, int(thr * 100))] = _np(fn(bev_a, scores, thr))
        if name == "rot":  # libm-robustness of the fixture: no rotated IoU within 1e-4 of the threshold
            assert np.abs(full - thr).min() > 1e-5, "borderline IoU in NMS fixture; change NMS_SEED"
    np.savez_compressed(os.path.join(HERE, "iou3d.npz"), **fx)
    out["iou3d"] = {k: int(len(v)) for k, v in fx.items() if k.startswith("keep_")}

    # ---- roipool3d_gpu through the reference surface (enlarge_box3d + zero fill + ext forward)
    pts, obj = synth.kitti_like_cloud(4096, 31, num_objects=20, return_boxes=True)
    rois = torch.cat([obj[:12], synth.proposal_boxes(4, seed=33, num_objects=20)[0]], dim=0)
    rois[-1, 0] = 500.0  # certainly empty
    pts_b = torch.stack([pts, synth.kitti_like_cloud(4096, 32, num_objects=20)], dim=0)
    feat_b = torch.randn((2, 4096, 6), generator=torch.Generator().manual_seed(34))
    rois_b = torch.stack([rois, rois], dim=0)
    pooled, empty = rp_u.roipool3d_gpu(pts_b, feat_b, rois_b, 0.2, sampled_pt_num=64)
    np.savez_compressed(os.path.join(HERE, "roipool3d_surface.npz"), pts=_np(pts_b), pts_feature=_np(feat_b), boxes3d=_np(rois_b),
                        pooled_features=_np(pooled), pooled_empty_flag=_np(empty))
    out["roipool3d_surface"] = dict(empty=_np(empty).tolist())

    # ---- (i) the reference's own compiled CPU ops
    ref = build_ref.load()
    assert ref is not None
    big = k_u.enlarge_box3d(rois, 0.2).contiguous()
    flag = torch.zeros((big.shape[0], pts.shape[0]), dtype=torch.int64)
    ref.pts_in_boxes3d_cpu(flag, pts.contiguous(), big)
    pooled_pts = torch.zeros((big.shape[0], 64, 3))
    pooled_feat = torch.zeros((big.shape[0], 64, 6))
    eflag = torch.zeros((big.shape[0],), dtype=torch.int64)
    ref.roipool3d_cpu(pts.contiguous(), big, feat_b[0].contiguous(), pooled_pts, pooled_feat, eflag)
    np.savez_compressed(os.path.join(HERE, "roipool3d_ref.npz"), pts=_np(pts), boxes3d=_np(big), pts_feature=_np(feat_b[0]),
                        pts_flag=np.packbits(_np(flag).astype(np.uint8), axis=1), pooled_pts=_np(pooled_pts),
                        pooled_features=_np(pooled_feat), pooled_empty_flag=_np(eflag))
    out["roipool3d_ref"] = dict(in_box=int(flag.sum()), empty=_np(eflag).tolist())

    # ---- parameter names the reference's builders create (checkpoint compatibility)
    names = {
        "SharedMLP_bn": list(ptu.SharedMLP([4, 8, 16], bn=True).state_dict().keys()),
        "SharedMLP_plain": list(ptu.SharedMLP([4, 8], bn=False).state_dict().keys()),
        "SharedMLP_preact_first": list(ptu.SharedMLP([4, 8, 16], bn=True, preact=True, first=True).state_dict().keys()),
        "SharedMLP_instance_norm": list(ptu.SharedMLP([4, 8], bn=False, instance_norm=True).state_dict().keys()),
        "Conv1d_bn": list(ptu.Conv1d(4, 8, bn=True).state_dict().keys()),
        "Conv1d_noact": list(ptu.Conv1d(4, 1, activation=None).state_dict().keys()),
        "Conv2d_named": list(ptu.Conv2d(4, 8, bn=True, name="x_").state_dict().keys()),
        "FC_bn": list(ptu.FC(4, 8, bn=True).state_dict().keys()),
        "SA_MSG": list(sa.state_dict().keys()),
        "FP": list(fpm.state_dict().keys()),
    }
    with open(os.path.join(HERE, "state_dict_names.json"), "w") as f:
        json.dump(names, f, indent=1, sort_keys=True)

    print(json.dumps(out))
    for fn in sorted(os.listdir(HERE)):
        print("%9d  %s" % (os.path.getsize(os.path.join(HERE, fn)), fn))


if __name__ == "__main__":
    main()
