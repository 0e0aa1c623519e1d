"""BASELINE config 4 (per-rank part) as an integration case: one rcnn_online training step of the point stream --
RPN backbone (SA + FP modules), proposal layer, target layer (IoU, ROI augmentation, roipool3d), RCNN SA stack --
forward + backward through every op of the hot path (bench_step.py's harness at a reduced size)."""
import numpy as np
import pytest
import torch


@pytest.mark.gpu
def test_rcnn_online_step_forward_backward(hiplib):
    import bench_step
    from epnet_amd import proposal_layer as pl, proposal_target_layer as ptl
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    np.random.seed(0)
    model = bench_step.build_model(scale=8).to(dev)
    layers = (pl.ProposalLayer("TRAIN").to(dev), ptl.ProposalTargetLayer())
    xyz, gts = bench_step.synthetic_batch(2, 2048, 7, dev)
    opt = torch.optim.SGD(model.parameters(), lr=1e-3)
    losses = []
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        loss, out = bench_step.run_step(model, layers, xyz, gts)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses))
    assert tuple(out["rois"].shape) == (2, 512, 7)
    t = out["target"]
    assert tuple(t["sampled_pts"].shape) == (128, 512, 3) and tuple(t["pts_feature"].shape) == (128, 512, 130)
    assert tuple(out["rcnn_cls"].shape) == (128, 1, 1) and tuple(out["rcnn_reg"].shape) == (128, 46, 1)
    missing = [n for n, p in model.named_parameters() if p.grad is None]
    assert not missing, missing
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())
    # the gradient reaches the first SA level of the backbone through FP modules, grouping and interpolation
    first = next(p for n, p in model.named_parameters() if n.startswith("backbone.SA_modules.0") and n.endswith("conv.weight"))
    assert float(first.grad.abs().sum()) > 0


@pytest.mark.gpu
def test_rcnn_online_step_full_size(hiplib, oracle, monkeypatch):
    """BASELINE config 4, the per-rank part at FULL size: 2 scenes x 16384 points, a 384 x 1280 image, the whole pyramid (scale 1),
    one rcnn_online training step of the two-stream model, forward + backward. Shapes and finite gradients as in the reduced
    cases below -- and what the step's own proposal layer (score sort, distance-based split, rotated NMS, 512 proposals per scene)
    and ROI pooling produced is held to the oracle on exactly the tensors those ops were handed."""
    import bench_step
    from epnet_amd import iou3d_cuda, roipool3d_cuda, proposal_layer as pl, proposal_target_layer as ptl
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    np.random.seed(0)
    seen = {"proposals": [], "pool": []}
    real_proposals, real_pool = iou3d_cuda.rpn_proposals_gpu, roipool3d_cuda.forward

    def proposals_spy(proposals, scores, order, distance_based, pre, post, thresh, rotated, ret_bbox3d, ret_scores, ret_count=None):
        args = (proposals.detach().cpu().numpy().copy(), scores.detach().cpu().numpy().copy(), order.cpu().numpy().copy(),
                bool(distance_based), int(pre), int(post), float(thresh), bool(rotated))
        r = real_proposals(proposals, scores, order, distance_based, pre, post, thresh, rotated, ret_bbox3d, ret_scores, ret_count)
        seen["proposals"].append((args, ret_bbox3d.cpu().numpy().copy(), ret_scores.cpu().numpy().copy(),
                                  None if ret_count is None else ret_count.cpu().numpy().copy()))
        return r

    def pool_spy(xyz, boxes3d, pts_feature, pooled_features, pooled_empty_flag):
        args = (xyz.detach().cpu().numpy().copy(), boxes3d.detach().cpu().numpy().copy(), pts_feature.detach().cpu().numpy().copy())
        r = real_pool(xyz, boxes3d, pts_feature, pooled_features, pooled_empty_flag)
        seen["pool"].append((args, pooled_features.cpu().numpy().copy(), pooled_empty_flag.cpu().numpy().copy()))
        return r

    monkeypatch.setattr(iou3d_cuda, "rpn_proposals_gpu", proposals_spy)
    monkeypatch.setattr(roipool3d_cuda, "forward", pool_spy)
    b, n = 2, 16384
    model = bench_step.build_model(scale=1, image=True).to(dev)
    layers = (pl.ProposalLayer("TRAIN").to(dev), ptl.ProposalTargetLayer())
    xyz, gts = bench_step.synthetic_batch(b, n, 7, dev)
    g = torch.Generator().manual_seed(1)
    image = torch.randn((b, 3, 384, 1280), generator=g).to(dev)
    xy = (torch.rand((b, n, 2), generator=g) * torch.tensor([1280.0, 384.0])).to(dev)
    loss, out = bench_step.run_step(model, layers, xyz, gts, None, image, xy, False)
    loss.backward()
    assert np.isfinite(float(loss.detach()))
    assert tuple(out["rois"].shape) == (b, 512, 7)
    t = out["target"]
    assert tuple(t["sampled_pts"].shape) == (b * 64, 512, 3) and tuple(t["pts_feature"].shape) == (b * 64, 512, 130)
    assert tuple(out["rcnn_cls"].shape) == (b * 64, 1, 1) and tuple(out["rcnn_reg"].shape) == (b * 64, 46, 1)
    missing = [name for name, p in model.named_parameters() if p.grad is None]
    assert not missing, missing
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())
    assert sum(p.numel() for p in model.backbone.parameters()) == 14131949
    for name in ("backbone.SA_modules.0.mlps.0.layer0.conv.weight", "backbone.Img_Block.0.conv1.weight"):
        assert float(dict(model.named_parameters())[name].grad.abs().sum()) > 0, name
    # ---- the proposal layer of this very step against the oracle (keep lists bit-exact: boxes, scores and counts are copies)
    assert len(seen["proposals"]) == 1 and len(seen["pool"]) == 1
    (proposals, scores, order, distance_based, pre, post, thresh, rotated), got_boxes, got_scores, got_count = seen["proposals"][0]
    assert proposals.shape == (b, n, 7) and pre == 9000 and post == 512
    want_boxes, want_scores, want_count = oracle.rpn_proposals(proposals, scores, order, distance_based, pre, post, thresh, rotated)
    np.testing.assert_array_equal(got_boxes, want_boxes)
    np.testing.assert_array_equal(got_scores, want_scores)
    if got_count is not None:
        np.testing.assert_array_equal(got_count.reshape(-1), want_count.reshape(-1))
    assert (want_count > 50).all()                       # (a real NMS problem: hundreds of the 9000 candidates survive)
    np.testing.assert_array_equal(out["rois"].cpu().numpy(), want_boxes)
    # ---- and its ROI pooling (64 sampled ROIs x 512 points x (3 + 130) per scene)
    (p_xyz, p_boxes, p_feat), got_pooled, got_flag = seen["pool"][0]
    assert p_xyz.shape == (b, n, 3) and p_boxes.shape == (b, 64, 7) and got_pooled.shape == (b, 64, 512, 3 + p_feat.shape[2])
    want_pooled, want_flag = oracle.roipool3d(p_xyz, p_boxes, p_feat, 512)
    np.testing.assert_array_equal(got_flag, want_flag)
    np.testing.assert_array_equal(got_pooled, want_pooled)
    assert (want_flag == 0).any()                        # some ROIs do hold points


@pytest.mark.gpu
@pytest.mark.parametrize("rpn_only", [True, False])
def test_two_stream_step_forward_backward(hiplib, rpn_only):
    """BASELINE configs 3 (rpn_only) and 4 with the image stream, at a reduced size: every parameter of the two-stream
    model -- image blocks, attention fusion, deconvolutions included -- receives a finite gradient through the HIP sampler"""
    import bench_step
    from epnet_amd import proposal_layer as pl, proposal_target_layer as ptl
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    np.random.seed(0)
    model = bench_step.build_model(scale=8, image=True).to(dev)
    layers = (pl.ProposalLayer("TRAIN").to(dev), ptl.ProposalTargetLayer())
    xyz, gts = bench_step.synthetic_batch(2, 2048, 7, dev)
    g = torch.Generator().manual_seed(1)
    image = torch.randn((2, 3, 96, 320), generator=g).to(dev)
    xy = (torch.rand((2, 2048, 2), generator=g) * torch.tensor([1280.0, 384.0])).to(dev)
    xy_before = xy.clone()
    loss, out = bench_step.run_step(model, layers, xyz, gts, None, image, xy, rpn_only)
    loss.backward()
    assert torch.equal(xy, xy_before)                       # run_step hands the model a copy: the model normalises in place
    assert np.isfinite(float(loss.detach()))
    skip = ("rcnn.",) if rpn_only else ()
    missing = [n for n, p in model.named_parameters() if p.grad is None and not n.startswith(skip)] if skip else \
        [n for n, p in model.named_parameters() if p.grad is None]
    assert not missing, missing
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    img_w = dict(model.named_parameters())["backbone.Img_Block.0.conv1.weight"]
    assert float(img_w.grad.abs().sum()) > 0
    assert sum(p.numel() for p in model.backbone.parameters()) == 14131949


@pytest.mark.gpu
def test_rpn_stage_records_into_a_hip_graph(hiplib):
    """backbone (sampling pyramid on a side stream, SA / FP modules with folded first layers and the pool kernel), heads and
    proposal layer in eval mode: nothing synchronises with the host, so the stage captures into a HIP graph whose replay
    reproduces the eager outputs, also on new input coordinates"""
    import bench_step
    from epnet_amd import proposal_layer as pl, synth
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = bench_step.build_model(scale=8).to(dev).eval()
    layer = pl.ProposalLayer("TEST").to(dev)
    xyz = synth.scenes("kitti", 1, 2048, seed=3).to(dev)

    def stage():
        with torch.no_grad():
            _, feats = model.backbone(xyz)
            cls = model.rpn_cls(feats).transpose(1, 2).contiguous()
            reg = model.rpn_reg(feats).transpose(1, 2).contiguous()
            return proposal_layer_out(cls, reg)

    def proposal_layer_out(cls, reg):
        return layer(cls[:, :, 0].contiguous(), reg, xyz)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            eager = stage()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = stage()
    graph.replay()
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(eager, captured))
    xyz.copy_(synth.scenes("kitti", 1, 2048, seed=4).to(dev))
    graph.replay()
    torch.cuda.synchronize()
    fresh = stage()
    assert all(torch.equal(a, b) for a, b in zip(fresh, captured)) and not torch.equal(fresh[0], eager[0])
