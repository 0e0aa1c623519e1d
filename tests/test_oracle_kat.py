"""Analytic known-answer tests that pin the CPU oracle (the reference ships no tests or fixtures,
SURVEY.md section 4; these are hand-derived from the reference kernels' text)."""
import math

import numpy as np
import pytest


def test_opt_n_threads(oracle):
    # cuda_utils.h:10-14: largest power of two <= n, capped at 1024
    for n, bs in [(1, 1), (2, 2), (3, 2), (64, 64), (100, 64), (128, 128), (256, 256), (512, 512), (1000, 512),
                  (1024, 1024), (4096, 1024), (16384, 1024), (65536, 1024)]:
        assert oracle.opt_n_threads(n) == bs


def test_fps_line_tiebreak(oracle):
    """8 collinear points, block size 8 (one point per thread). After {0, 7} points 3 and 4 tie at
    d2 = 9: the shared-memory tree (sampling_gpu.cu:143-203) pairs slot t with t+4, t+2, t+1 and the
    lower slot wins ties, so the winner has the smallest bit-reversed tid: 4 (0b100 -> 1) beats
    3 (0b011 -> 6). Continuing by hand: 2, then the four-way tie {1,3,5,6} -> 6, 1, 5, 3."""
    xyz = np.zeros((1, 8, 3), np.float32)
    xyz[0, :, 0] = np.arange(8)
    assert oracle.furthest_point_sampling(xyz, 8)[0].tolist() == [0, 7, 4, 2, 6, 1, 5, 3]


def test_fps_within_thread_first_max(oracle):
    """n = 16, bs = 16... use n = 3 (bs = 2): thread 0 holds k = 0, 2; thread 1 holds k = 1.
    points x = 0, 5, 5: after picking 0, k=1 and k=2 tie at 25; the tree compares slot 0 (thread 0's
    best = k 2) with slot 1 (k 1): tie -> slot 0 -> index 2, although 1 < 2."""
    xyz = np.zeros((1, 3, 3), np.float32)
    xyz[0, :, 0] = [0, 5, 5]
    assert oracle.furthest_point_sampling(xyz, 2)[0].tolist() == [0, 2]


def test_fps_lattice_and_distances(oracle):
    # 4x4 lattice: second pick is the far corner, running distances are the min over picks
    g = np.stack(np.meshgrid(np.arange(4), np.arange(4), indexing="ij"), -1).reshape(-1, 2)
    xyz = np.zeros((1, 16, 3), np.float32)
    xyz[0, :, :2] = g
    idx, temp = oracle.furthest_point_sampling(xyz, 2, return_temp=True)
    assert idx[0].tolist() == [0, 15]
    # temp was updated with the distance to pick 0 only (the loop runs m-1 = 1 times)
    np.testing.assert_array_equal(temp[0], (g ** 2).sum(1).astype(np.float32))


def test_fps_all_duplicates(oracle):
    xyz = np.ones((2, 64, 3), np.float32)
    idx = oracle.furthest_point_sampling(xyz, 5)
    # every distance is 0: the winner of each round is the tree's default, slot 0's first point
    assert idx.tolist() == [[0, 0, 0, 0, 0]] * 2


def test_ball_query_semantics(oracle):
    # points on the x axis at 0, 1, 2, ..., 9; centre at 4, radius 2 -> strict '<': 3, 4, 5 only
    xyz = np.zeros((1, 10, 3), np.float32)
    xyz[0, :, 0] = np.arange(10)
    centre = np.array([[[4, 0, 0]]], np.float32)
    assert oracle.ball_query(2.0, 5, xyz, centre)[0, 0].tolist() == [3, 4, 5, 3, 3]      # padded with first hit
    assert oracle.ball_query(2.0, 2, xyz, centre)[0, 0].tolist() == [3, 4]               # truncated in index order
    assert oracle.ball_query(2.0001, 8, xyz, centre)[0, 0].tolist() == [2, 3, 4, 5, 6, 2, 2, 2]
    far = np.array([[[100, 0, 0]]], np.float32)
    assert oracle.ball_query(2.0, 4, xyz, far)[0, 0].tolist() == [0, 0, 0, 0]             # empty ball: zeros


def test_group_and_gather(oracle):
    feats = np.arange(2 * 3 * 5, dtype=np.float32).reshape(2, 3, 5)
    idx = np.array([[[4, 0], [1, 1]], [[2, 3], [0, 4]]], np.int32)
    out = oracle.group_points(feats, idx)
    assert out.shape == (2, 3, 2, 2)
    for b in range(2):
        for c in range(3):
            np.testing.assert_array_equal(out[b, c], feats[b, c][idx[b]])
    g = oracle.gather_points(feats, idx[:, :, 0].copy())
    np.testing.assert_array_equal(g[1, 2], feats[1, 2][[2, 0]])
    grad = oracle.group_points_grad(np.ones_like(out), idx, 5)
    assert grad[0, 0].tolist() == [1, 2, 0, 0, 1] and grad[1, 1].tolist() == [1, 0, 1, 1, 1]


def test_three_nn_ties_and_short(oracle):
    known = np.zeros((1, 4, 3), np.float32)
    known[0, :, 0] = [1, -1, 2, 1]      # distances from origin: 1, 1, 4, 1  -> earliest indices win ties
    unknown = np.zeros((1, 1, 3), np.float32)
    d2, idx = oracle.three_nn(unknown, known)
    assert idx[0, 0].tolist() == [0, 1, 3] and d2[0, 0].tolist() == [1, 1, 1]
    d2, idx = oracle.three_nn(unknown, known[:, :2])   # m = 2 < 3: third slot keeps (1e40 -> inf, 0)
    assert idx[0, 0].tolist() == [0, 1, 0] and d2[0, 0, 2] == np.inf


def test_three_interpolate(oracle):
    pts = np.arange(8, dtype=np.float32).reshape(1, 2, 4)
    idx = np.array([[[0, 1, 3]]], np.int32)
    w = np.array([[[0.5, 0.25, 0.25]]], np.float32)
    out = oracle.three_interpolate(pts, idx, w)
    assert out[0, :, 0].tolist() == [0.5 * 0 + 0.25 * 1 + 0.25 * 3, 0.5 * 4 + 0.25 * 5 + 0.25 * 7]
    g = oracle.three_interpolate_grad(np.ones((1, 2, 1), np.float32), idx, w, 4)
    assert g[0, 0].tolist() == [0.5, 0.25, 0, 0.25]


def bev(x1, y1, x2, y2, ry):
    return np.array([[x1, y1, x2, y2, ry]], np.float32)


def test_box_overlap_known_areas(oracle):
    unit = bev(0, 0, 1, 1, 0)
    # identical boxes: all 8 corners pass the MARGIN test, collinear edges are rejected by s1*s2 > 0
    assert oracle.boxes_overlap_bev(unit, unit)[0, 0] == pytest.approx(1.0, abs=1e-6)
    assert oracle.boxes_iou_bev(unit, unit)[0, 0] == pytest.approx(1.0, abs=1e-6)
    # half overlap, axis aligned: intersection 0.5, union 1.5
    shifted = bev(0.5, 0, 1.5, 1, 0)
    assert oracle.boxes_overlap_bev(unit, shifted)[0, 0] == pytest.approx(0.5, abs=1e-6)
    assert oracle.boxes_iou_bev(unit, shifted)[0, 0] == pytest.approx(1 / 3, abs=1e-6)
    # unit square vs itself rotated by 45 degrees about the common centre: regular octagon 2(sqrt2-1)
    rot = bev(0, 0, 1, 1, math.pi / 4)
    assert oracle.boxes_overlap_bev(unit, rot)[0, 0] == pytest.approx(2 * (math.sqrt(2) - 1), abs=1e-5)
    # rotating BOTH by the same angle changes nothing
    a, b = bev(0, 0, 2, 1, 0.3), bev(1, 0, 3, 1, 0.3)
    centre_shift = oracle.boxes_overlap_bev(a, b)[0, 0]
    assert 0 < centre_shift < 2
    # disjoint
    assert oracle.boxes_overlap_bev(unit, bev(5, 5, 6, 6, 1.0))[0, 0] == 0.0
    # a 90-degree rotated 2x1 box centred on a 2x1 box: intersection is the 1x1 centre square
    assert oracle.boxes_overlap_bev(bev(0, 0, 2, 1, 0), bev(0, 0, 2, 1, math.pi / 2))[0, 0] == pytest.approx(1.0, abs=1e-5)


def test_nms_chain(oracle):
    # three axis-aligned boxes: 0 overlaps 1 heavily, 1 overlaps 2 heavily, 0 and 2 barely
    boxes = np.array([[0, 0, 10, 10, 0], [1, 0, 11, 10, 0], [2.5, 0, 12.5, 10, 0], [50, 50, 60, 60, 0]], np.float32)
    for rotated in (False, True):
        assert oracle.nms(boxes, 0.7, rotated).tolist() == [0, 2, 3]   # 1 suppressed by 0; 2 survives (IoU 0.6)
        assert oracle.nms(boxes, 0.5, rotated).tolist() == [0, 3]
        mask = oracle.nms_mask(boxes, 0.7, rotated)
        assert mask[:, 0].tolist() == [0b0010, 0b0100, 0, 0]            # only bits to the right of the diagonal


def test_nms_many_blocks(oracle):
    # 130 boxes in 3 mask blocks: identical pairs (2k, 2k+1) on a 9x8 grid -> every odd box is suppressed.
    # (coordinates stay small: identical rotated boxes only overlap through check_in_box2d's 1e-5 MARGIN,
    #  iou3d_kernel.cu:52, which fp32 rounding defeats beyond ~64 m -- reference behaviour, kept)
    k = np.arange(65)
    x0, y0 = (k % 9).astype(np.float32) * 3, (k // 9).astype(np.float32) * 3
    boxes = np.repeat(np.stack([x0, y0, x0 + 2, y0 + 2, np.full_like(x0, 0.3)], 1), 2, 0)
    assert oracle.nms(boxes, 0.9, False).tolist() == list(range(0, 130, 2))
    assert oracle.nms(boxes, 0.9, True).tolist() == list(range(0, 130, 2))


def test_pt_in_box_and_roipool(oracle):
    # box centred at (0, y, 0), bottom at y = 1, h = 2 -> y in [-1, 1]; l = 4 along x, w = 2 along z, ry = 0
    box = np.array([[[0, 1, 0, 2, 2, 4, 0]]], np.float32)
    pts = np.array([[[0, 0, 0], [2, 0, 1], [2.01, 0, 0], [0, 1.01, 0], [0, -1, -1], [-2, 1, 1], [9, 9, 9]]], np.float32)
    feat = np.arange(7, dtype=np.float32).reshape(1, 7, 1) + 10
    pooled, flag = oracle.roipool3d(pts, box, feat, 6)
    assert flag.tolist() == [[0]]
    # in-box (inclusive bounds): 0, 1, 4, 5 -> cyclic pad 0, 1
    assert pooled[0, 0, :, 3].tolist() == [10, 11, 14, 15, 10, 11]
    np.testing.assert_array_equal(pooled[0, 0, :4, :3], pts[0, [0, 1, 4, 5]])
    assert oracle.pts_in_boxes3d(pts[0], box[0])[0].tolist() == [1, 1, 0, 0, 1, 1, 0]
    # rotated by 90 degrees the long side lies along z
    box90 = box.copy(); box90[0, 0, 6] = math.pi / 2
    assert oracle.pts_in_boxes3d(np.array([[0, 0, 1.9], [1.9, 0, 0]], np.float32), box90[0])[0].tolist() == [1, 0]
    # empty box: flag 1 and zeros left untouched
    pooled, flag = oracle.roipool3d(pts, box + np.array([100, 0, 0, 0, 0, 0, 0], np.float32), feat, 4)
    assert flag.tolist() == [[1]] and not pooled.any()
    # truncation keeps the first S in index order
    pooled, _ = oracle.roipool3d(pts, box, feat, 2)
    assert pooled[0, 0, :, 3].tolist() == [10, 11]
