"""SURVEY.md section 8(f) row N2 -- epnet_amd.proposal_layer / bbox_transform against what the REFERENCE'S OWN
lib/rpn/proposal_layer.py and lib/utils/bbox_transform.py produced (tests/golden/proposal_layer.npz, written by
make_golden_rcnn.py running the unmodified reference classes on the CPU over oracle-backed NMS stand-ins).

CPU half: decoding and the Python side over the oracle's restatement of the per-scene loop; GPU half (-m gpu): the same
fixtures and random cases through epnet_rpn_proposals (bin compaction + batched NMS with device-side counts + gather)."""
import numpy as np
import pytest
import torch

from conftest import golden

CASES = ("train_normal", "test_rotate", "train_tight", "score_based", "padded_normal", "padded_rotate", "padded_score")


def T(a, device="cpu"):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


@pytest.fixture()
def cpu_surface(monkeypatch, oracle):
    import oracle_ext
    from epnet_amd import iou3d_cuda
    _, iou, _ = oracle_ext.make_modules()
    for name, fn in vars(iou).items():
        if callable(fn):
            monkeypatch.setattr(iou3d_cuda, name, fn)
    return "cpu"


def layer_for(fx, tag):
    from epnet_amd import proposal_layer as pl
    is_train, dist_based, rotate, pre, post, thresh = fx[tag + "__cfg"]
    cfg = pl.default_cfg()
    mode = "TRAIN" if is_train else "TEST"
    m = getattr(cfg, mode)
    m.RPN_PRE_NMS_TOP_N, m.RPN_POST_NMS_TOP_N, m.RPN_NMS_THRESH = int(pre), int(post), float(thresh)
    cfg.TEST.RPN_DISTANCE_BASED_PROPOSE = bool(dist_based)
    cfg.RPN.NMS_TYPE = "rotate" if rotate else "normal"
    return pl.ProposalLayer(mode=mode, cfg=cfg)


def check_decode(device):
    from epnet_amd import bbox_transform as bt, proposal_layer as pl
    fx = golden("proposal_layer.npz")
    anchor = torch.from_numpy(pl.default_cfg().CLS_MEAN_SIZE[0])
    xyz, reg = T(fx["xyz"], device), T(fx["rpn_reg_f16"], device).float()
    for avg in (True, False):
        got = bt.decode_bbox_target(xyz.view(-1, 3), reg.view(-1, reg.shape[-1]), anchor_size=anchor, loc_scope=3.0, loc_bin_size=0.5,
                                    num_head_bin=12, get_xz_fine=True, get_y_by_bin=False, get_ry_fine=False, bbox_avg_by_bin=avg)
        np.testing.assert_allclose(got.cpu().numpy(), fx["decode_rpn_avg%d" % avg], rtol=1e-5, atol=1e-5)
    got = bt.decode_bbox_target(T(fx["rcnn_rois"], device), T(fx["rcnn_reg_f16"], device).float(), anchor_size=anchor, loc_scope=1.5,
                                loc_bin_size=0.5, num_head_bin=9, get_xz_fine=True, get_y_by_bin=False, loc_y_scope=0.5,
                                loc_y_bin_size=0.25, get_ry_fine=True, bbox_avg_by_bin=False)
    np.testing.assert_allclose(got.cpu().numpy(), fx["decode_rcnn"], rtol=1e-5, atol=1e-5)


def check_layer(device, tag):
    fx = golden("proposal_layer.npz")
    layer = layer_for(fx, tag).to(device)
    boxes, scores = layer(T(fx["rpn_scores"], device), T(fx["rpn_reg_f16"], device).float(), T(fx["xyz"], device))
    want_b, want_s = fx[tag + "__bbox3d"], fx[tag + "__scores"]
    assert tuple(boxes.shape) == want_b.shape and tuple(scores.shape) == want_s.shape
    # the same proposals in the same order (scores are copies: exact), boxes to the decoding's tolerance, zero padding behind
    np.testing.assert_array_equal(scores.cpu().numpy(), want_s)
    np.testing.assert_allclose(boxes.cpu().numpy(), want_b, rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------------------------------------------ CPU half
def test_decode_matches_reference_cpu():
    check_decode("cpu")


@pytest.mark.parametrize("tag", CASES)
def test_layer_matches_reference_cpu(cpu_surface, tag):
    check_layer("cpu", tag)


def test_oracle_far_bin_falls_back_to_near_bin(oracle):
    """no box beyond 40 m: the far bin is served by the near bin's boxes after its own pre-NMS budget (:92-100)"""
    n, pre, post = 50, 20, 10
    p = np.zeros((1, n, 7), np.float32)
    p[0, :, 0] = np.arange(n) * 10.0          # far apart: nothing suppresses anything
    p[0, :, 2] = 10.0
    p[0, :, 3:6] = 1.0
    s = np.linspace(1, 0, n, dtype=np.float32)[None]
    order = np.arange(n)[None]
    rb, rs, cnt = oracle.rpn_proposals(p, s, order, True, pre, post, 0.5, False)
    # near bin: first int(20*.7)=14 boxes -> first 7 kept; far bin: boxes 14..19 (6 of them) -> first 3 kept
    assert cnt[0] == 10 and rb[0, :7, 0].tolist() == [0, 10, 20, 30, 40, 50, 60] and rb[0, 7:, 0].tolist() == [140, 150, 160]


# ------------------------------------------------------------------------------------------------------ GPU half
@pytest.mark.gpu
def test_decode_matches_reference_gpu(hiplib):
    check_decode("cuda")


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_layer_matches_reference_gpu(hiplib, tag):
    check_layer("cuda", tag)


@pytest.mark.gpu
@pytest.mark.parametrize("b,n,dist_based,rotated,pre,post,thresh,kind", [
    (1, 16384, True, False, 9000, 512, 0.85, "kitti"),     # the training call (yaml :171-174)
    (2, 16384, True, False, 9000, 100, 0.8, "kitti"),      # the evaluation call (yaml :185-188)
    (2, 4096, True, True, 3000, 300, 0.3, "kitti"),
    (3, 2000, False, True, 1500, 700, 0.1, "kitti"),
    (2, 3000, True, False, 1000, 900, 0.05, "near"),       # far bin empty: fallback
    (2, 3000, True, True, 1000, 900, 0.05, "far"),         # near bin empty
    (2, 500, True, False, 1000, 900, 0.5, "kitti"),        # fewer boxes than the budgets
    (1, 64, True, False, 10, 4, 0.5, "outside"),           # nothing inside either bin
    (2, 777, True, False, 1, 1, 0.5, "kitti"),             # int(1 * 0.7) = 0: the near bin has no budget
])
def test_rpn_proposals_equals_oracle(hiplib, oracle, b, n, dist_based, rotated, pre, post, thresh, kind):
    """epnet_rpn_proposals against the oracle's scene-by-scene restatement of the reference loop: identical scores (same
    boxes kept in the same order), identical box rows, identical counts"""
    from epnet_amd import iou3d_cuda, synth
    g = torch.Generator().manual_seed(b * 1000 + n)
    xyz = synth.scenes("kitti", b, n, seed=n)
    boxes = torch.zeros((b, n, 7))
    boxes[:, :, 0:3] = xyz + (torch.rand((b, n, 3), generator=g) - 0.5) * torch.tensor([2.0, 0.2, 2.0])
    boxes[:, :, 3:6] = torch.tensor([1.5, 1.6, 3.9]) * (0.8 + 0.4 * torch.rand((b, n, 3), generator=g))
    boxes[:, :, 6] = (torch.rand((b, n), generator=g) - 0.5) * 6
    if kind == "near":
        boxes[:, :, 2] = boxes[:, :, 2] * 0.5 + 0.5
    elif kind == "far":
        boxes[:, :, 2] = boxes[:, :, 2] * 0.5 + 41
    elif kind == "outside":
        boxes[:, :, 2] = -boxes[:, :, 2] - 1
    scores = torch.randn((b, n), generator=g)
    order = torch.sort(scores, dim=1, descending=True)[1]
    want_b, want_s, want_c = oracle.rpn_proposals(boxes.numpy(), scores.numpy(), order.numpy(), dist_based, pre, post, thresh, rotated)
    d = "cuda"
    rb = torch.full((b, post, 7), float("nan"), device=d)
    rs = torch.full((b, post), float("nan"), device=d)
    rc = torch.full((b,), -1, dtype=torch.int32, device=d)
    iou3d_cuda.rpn_proposals_gpu(boxes.to(d), scores.to(d), order.to(d), dist_based, pre, post, thresh, rotated, rb, rs, rc)
    np.testing.assert_array_equal(rc.cpu().numpy(), want_c)
    np.testing.assert_array_equal(rs.cpu().numpy(), want_s)
    np.testing.assert_array_equal(rb.cpu().numpy(), want_b)


@pytest.mark.gpu
def test_proposal_layer_captures_into_a_hip_graph(hiplib):
    """nothing in the layer reads the device back, so it records into a graph and replays on new inputs"""
    from epnet_amd import proposal_layer as pl, synth
    d = "cuda"
    layer = pl.ProposalLayer("TRAIN").to(d)
    b, n = 2, 16384
    g = torch.Generator().manual_seed(3)
    xyz = synth.scenes("kitti", b, n, seed=4).to(d)
    reg = (torch.randn((b, n, 76), generator=g) * 0.5).to(d)
    sc = torch.randn((b, n), generator=g).to(d)
    eager_b, eager_s = layer(sc, reg, xyz)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(2):
            layer(sc, reg, xyz)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        gb, gs = layer(sc, reg, xyz)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(gb, eager_b) and torch.equal(gs, eager_s)
    sc.copy_(torch.randn((b, n), generator=g))
    graph.replay()
    torch.cuda.synchronize()
    new_b, new_s = layer(sc, reg, xyz)
    assert torch.equal(gb, new_b) and torch.equal(gs, new_s) and not torch.equal(new_s, eager_s)


def test_callers_resolve_under_the_reference_paths_and_read_its_cfg(monkeypatch):
    """compat.install_callers(): `from lib.rpn.proposal_layer import ProposalLayer` (lib/net/rpn.py:4) and
    `from lib.rpn.proposal_target_layer import ProposalTargetLayer` (lib/net/rcnn_net.py:5) give this package's classes,
    constructed the reference's way (`ProposalLayer(mode=...)`, `ProposalTargetLayer()`), reading a loaded lib.config.cfg"""
    import sys
    import types
    from epnet_amd import compat, proposal_layer as pl, proposal_target_layer as ptl
    saved = {k: sys.modules.get(k) for k in list(sys.modules) if k == "lib" or k.startswith("lib.") or k.endswith("_cuda")}
    try:
        compat.install_callers()
        from lib.rpn.proposal_layer import ProposalLayer
        from lib.rpn.proposal_target_layer import ProposalTargetLayer
        assert ProposalLayer is pl.ProposalLayer and ProposalTargetLayer is ptl.ProposalTargetLayer
        assert ProposalLayer(mode='TEST')._mode_cfg().RPN_POST_NMS_TOP_N == 100        # yaml defaults
        fake = types.ModuleType("lib.config")
        fake.cfg = pl.default_cfg()
        fake.cfg.TEST.RPN_POST_NMS_TOP_N = 77
        fake.cfg.RCNN = ptl.default_cfg().RCNN
        fake.cfg.RCNN.ROI_PER_IMAGE = 32
        fake.cfg.AUG_DATA, fake.cfg.AUG_ROT_RANGE = False, 18
        monkeypatch.setitem(sys.modules, "lib.config", fake)
        assert ProposalLayer(mode='TEST')._mode_cfg().RPN_POST_NMS_TOP_N == 77
        assert ProposalTargetLayer().cfg.RCNN.ROI_PER_IMAGE == 32
    finally:
        for k in [k for k in sys.modules if k == "lib" or k.startswith("lib.") or k.endswith("_cuda")]:
            if k not in saved:
                del sys.modules[k]
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v


def test_heading_as_weighted_mean_of_the_likelier_side_matches_the_reference():
    """RY_WITH_BIN (lib/utils/bbox_transform.py:146-238; off in every shipped config, lib/config.py:199,209): the fixture is what
    the reference's own decode_bbox_target returned with the switch on (tests/golden/make_golden_ry_bin.py), RPN and RCNN call
    shapes, peaked and exactly flat bin distributions included"""
    from epnet_amd import bbox_transform as bt, proposal_layer as pl
    fx = golden("ry_with_bin.npz")
    anchor = torch.from_numpy(pl.default_cfg().CLS_MEAN_SIZE[0])
    got = bt.decode_bbox_target(T(fx["rpn_xyz"], "cpu"), T(fx["rpn_reg_f16"], "cpu").float(), anchor_size=anchor, loc_scope=3.0,
                                loc_bin_size=0.5, num_head_bin=12, get_xz_fine=True, get_y_by_bin=False, get_ry_fine=False,
                                bbox_avg_by_bin=False, ry_with_bin=True)
    np.testing.assert_allclose(got.numpy(), fx["rpn_decoded"], rtol=1e-5, atol=1e-5)
    got = bt.decode_bbox_target(T(fx["rcnn_rois"], "cpu"), T(fx["rcnn_reg_f16"], "cpu").float(), anchor_size=anchor, loc_scope=1.5,
                                loc_bin_size=0.5, num_head_bin=9, get_xz_fine=True, get_y_by_bin=False, loc_y_scope=0.5,
                                loc_y_bin_size=0.25, get_ry_fine=True, bbox_avg_by_bin=False, ry_with_bin=True)
    np.testing.assert_allclose(got.numpy(), fx["rcnn_decoded"], rtol=1e-5, atol=1e-5)
    # the switch changes the heading column only, and does change it
    plain = bt.decode_bbox_target(T(fx["rcnn_rois"], "cpu"), T(fx["rcnn_reg_f16"], "cpu").float(), anchor_size=anchor, loc_scope=1.5,
                                  loc_bin_size=0.5, num_head_bin=9, get_xz_fine=True, get_y_by_bin=False, loc_y_scope=0.5,
                                  loc_y_bin_size=0.25, get_ry_fine=True, bbox_avg_by_bin=False, ry_with_bin=False)
    assert torch.equal(plain[:, :6], got[:, :6]) and not torch.allclose(plain[:, 6], got[:, 6])
