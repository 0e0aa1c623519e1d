"""Second, independent derivations of the oracle functions the reference cannot pin here (no reference
tests, no nvcc) -- they do not change "parity unpinned", they lower the risk that oracle/epnet_oracle.c misreads the
.cu text:

  * furthest point sampling: the C oracle SIMULATES the reference's block (per-thread strided scan, shared-memory tree,
    sampling_gpu.cu:86-209). Here the same selection is derived as a closed-form rule -- among the points holding the
    maximum running distance the winner minimises (bit-reverse(k mod bs), k div bs), bs = opt_n_threads(n) -- and
    evaluated with whole-array numpy float32 arithmetic;
  * rotated-rectangle overlap: the reference collects edge intersections and inside corners, sorts them by angle and
    sums triangle areas in float32 (iou3d_kernel.cu:108-212). Here the overlap is the area of rectangle A clipped
    against the four half-planes of rectangle B (Sutherland-Hodgman) in float64;
  * hand-derived known answers for the tie-break across every block size and for NMS suppression chains;
  * round 3: the scan kernels in closed form -- ball query (mask + stable sort of a whole distance matrix instead of the sequential
    scan with early exit, ball_query_gpu.cu:9-45), three_nn (stable argsort = the strict-'<' insertion's (d, k) order,
    interpolate_gpu.cu:30-48), gather / grouping / interpolation and their gradients by numpy indexing; and the nesting of furthest
    point sampling through exact twins (what the sampling chain's twin rule rests on).
"""
import math

import numpy as np
import pytest


# ------------------------------------------------------------------------------------------------ furthest point sampling

def bit_reverse(v, bits):
    out = np.zeros_like(v)
    for i in range(bits):
        out |= ((v >> i) & 1) << (bits - 1 - i)
    return out


def fps_rank_rule(xyz, m, bs):
    """furthest point sampling of ONE scene by the rank rule; xyz (n,3) float32. Every arithmetic step is a float32
    numpy operation on whole arrays: (x2-x1)*(x2-x1) + (y2-y1)*(y2-y1) + (z2-z1)*(z2-z1) summed left to right, min with the
    running distance (sampling_gpu.cu:131-135)."""
    n = xyz.shape[0]
    k = np.arange(n)
    bits = int(math.log2(bs))
    rank = bit_reverse(k % bs, bits).astype(np.int64) * (n // bs + 2) + k // bs     # lexicographic (bitrev(tid), slot)
    temp = np.full(n, 1e10, np.float32)
    picks = [0]
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    for _ in range(1, m):
        o = picks[-1]
        dx, dy, dz = x - x[o], y - y[o], z - z[o]
        d = (dx * dx + dy * dy) + dz * dz
        assert d.dtype == np.float32
        temp = np.minimum(d, temp)
        cand = np.flatnonzero(temp == temp.max())
        picks.append(int(cand[np.argmin(rank[cand])]))
    return np.array(picks, np.int32), temp


def clouds_with_duplicates(n, seed):
    """points on a coarse lattice (many exactly equal distances) with repeated rows (the reference's own padding repeats
    points, kitti_rcnn_dataset.py:338-342)"""
    rng = np.random.default_rng(seed)
    base = rng.integers(-8, 9, size=(max(4, (n * 3) // 4), 3)).astype(np.float32) * np.float32(0.5)
    extra = base[rng.integers(0, len(base), size=n - len(base))]
    pts = np.concatenate([base, extra])
    return pts[rng.permutation(n)].astype(np.float32)


@pytest.mark.parametrize("n", [37, 64, 1000, 4096, 16384])
def test_fps_tree_simulation_equals_rank_rule(oracle, n):
    m = max(2, n // 4)
    xyz = clouds_with_duplicates(n, seed=n)
    bs = oracle.opt_n_threads(n)
    want, want_temp = fps_rank_rule(xyz, m, bs)
    got, got_temp = oracle.furthest_point_sampling(xyz[None], m, return_temp=True)
    np.testing.assert_array_equal(got[0], want)
    np.testing.assert_array_equal(got_temp[0], want_temp)
    # the lattice really produces ties: at least one round of the big scenes had several points at the maximum
    if n >= 1000:
        d = np.sort(got_temp[0])[::-1]
        assert (d[:8] == d[0]).all() or len(set(want.tolist())) < m


def test_fps_rank_rule_on_continuous_cloud(oracle):
    rng = np.random.default_rng(5)
    xyz = (rng.random((3000, 3)) * np.array([80, 4, 70])).astype(np.float32)
    want, _ = fps_rank_rule(xyz, 700, oracle.opt_n_threads(3000))
    np.testing.assert_array_equal(oracle.furthest_point_sampling(xyz[None], 700)[0], want)


def first_tie_round(xyz, m, bs, twins_are_ties=True):
    """the first round of the furthest point sampling of ONE scene whose maximum running distance is held by several points
    (m if there is none): numpy, the rank rule of fps_rank_rule.
    twins_are_ties=False: holders with IDENTICAL coordinates do not count as a tie while the maximum is positive (whichever
    of them the tie-break picks, the others drop to distance 0 in the next round and can never be sampled while some point
    is further than 0 from the samples: the sequence of sampled COORDINATES does not depend on the tie-break)."""
    n = xyz.shape[0]
    k = np.arange(n)
    rank = bit_reverse(k % bs, int(math.log2(bs))).astype(np.int64) * (n // bs + 2) + k // bs
    temp = np.full(n, 1e10, np.float32)
    o, picks = 0, [0]
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    for it in range(1, m):
        dx, dy, dz = x - x[o], y - y[o], z - z[o]
        temp = np.minimum((dx * dx + dy * dy) + dz * dz, temp)
        top = temp.max()
        cand = np.flatnonzero(temp == top)
        if cand.size > 1:
            harmless = (not twins_are_ties) and top > 0 and len(np.unique(xyz[cand].view(np.uint32), axis=0)) == 1
            if not harmless:
                return it, np.array(picks, np.int32)
        o = int(cand[np.argmin(rank[cand])])
        picks.append(o)
    return m, np.array(picks, np.int32)


@pytest.mark.parametrize("kind,n,m,m2", [("kitti", 4096, 1024, 256), ("ubox", 2000, 500, 125), ("dup", 4096, 1024, 256), ("lattice", 4096, 1024, 256)])
def test_fps_is_nested_while_the_maxima_are_unique(oracle, kind, n, m, m2):
    """what epnet_sample_centres_chain rests on: sample a cloud (n -> m), then sample the samples (m -> m2) -- as long as the
    maximum of every round of the first sampling was unique, the second sampling picks 0, 1, 2, ... (the reference's kernel on the
    centres: the oracle). With ties the identity holds up to the first tied round."""
    from epnet_amd import synth
    xyz = clouds_with_duplicates(n, seed=3) if kind == "lattice" else synth.scenes(kind, 1, n, seed=11).numpy()[0]
    tie, _ = first_tie_round(xyz, m, oracle.opt_n_threads(n))
    idx = oracle.furthest_point_sampling(xyz[None], m)
    centres = xyz[idx[0]]
    idx2 = oracle.furthest_point_sampling(centres[None], m2)[0]
    upto = min(tie, m2)
    np.testing.assert_array_equal(idx2[:upto], np.arange(upto))
    if kind in ("kitti", "ubox"):
        assert tie == m            # continuous coordinates: no exact ties, the whole second level is the identity
    if kind == "lattice":
        assert tie < m2            # the lattice ties early: the case the chain must NOT shortcut


@pytest.mark.parametrize("kind,n,m,m2", [("dup", 4096, 1024, 256), ("dup", 16384, 4096, 1024), ("dup_shuffled", 8192, 2048, 700),
                                         ("lattice", 4096, 1024, 256), ("few_distinct", 2048, 512, 450)])
def test_exact_twins_do_not_end_the_nesting(oracle, kind, n, m, m2):
    """The reference's loader pads short scenes by re-drawing rows (kitti_rcnn_dataset.py:338-342): exact twins, which tie at the
    round one of them is picked. Such a tie is harmless for the next level -- the unpicked twin never becomes a sample while
    the maximum is positive -- so the nesting holds up to the first round whose maximum is held by DIFFERENT coordinates (or
    is zero). Checked against the oracle's own rounds on the centres."""
    from epnet_amd import synth
    rng = np.random.default_rng(5)
    if kind == "lattice":
        xyz = clouds_with_duplicates(n, seed=3)
    elif kind == "few_distinct":      # more samples than distinct points: the maximum reaches zero inside the first sampling
        xyz = synth.kitti_like_cloud(400, 3).numpy()[rng.integers(0, 400, size=n)]
    else:
        xyz = synth.dup_cloud(n, 11, unique=(n * 3) // 4).numpy()
        if kind == "dup_shuffled":
            xyz = xyz[rng.permutation(n)]
    xyz = np.ascontiguousarray(xyz, np.float32)
    bs = oracle.opt_n_threads(n)
    strict, _ = first_tie_round(xyz, m, bs)
    relaxed, picks = first_tie_round(xyz, m, bs, twins_are_ties=False)
    idx = oracle.furthest_point_sampling(xyz[None], m)[0]
    np.testing.assert_array_equal(idx[:len(picks)], picks)      # (the rank rule picks what the oracle picks, through the twin rounds too)
    assert relaxed >= strict
    centres = xyz[idx]
    idx2 = oracle.furthest_point_sampling(centres[None], m2)[0]
    upto = min(relaxed, m2)
    np.testing.assert_array_equal(idx2[:upto], np.arange(upto))
    if kind.startswith("dup"):
        assert strict < m2 <= relaxed    # the twins tie early, yet the whole second level is the identity
    if kind == "few_distinct":
        assert relaxed <= 400 and not np.array_equal(idx2, np.arange(m2))    # zero maximum: the identity ends, and must be seen to


def _tie_scene(n, at_unit_distance):
    xyz = np.zeros((1, n, 3), np.float32)
    xyz[0, list(at_unit_distance), 0] = 1.0
    return xyz


@pytest.mark.parametrize("bs", [64, 128, 256, 512, 1024])
def test_fps_tie_break_across_the_reduction_tree_by_hand(oracle, bs):
    """n = bs: one point per thread, so only the shared-memory tree decides (sampling_gpu.cu:143-203). The tree's last
    comparison is slot 0 against slot 1, the one before it slots {0,1} against {2,3}, ...: the lowest bit of tid is the
    most significant key, i.e. the smallest bit-reversed tid wins a tie. Everything except the listed points sits on
    point 0 (distance 0); the listed ones are at distance 1, so the second pick is decided among them alone."""
    assert oracle.opt_n_threads(bs) == bs
    half = bs // 2
    # all other points at distance 1: tid = bs/2 (binary 10..0, reversed 0..01) has the smallest reversed value after tid 0
    every = oracle.furthest_point_sampling(_tie_scene(bs, range(1, bs)), 3)[0].tolist()
    assert every == [0, half, 0]          # third round: every distance is 0, slot 0 wins with its only point
    # odd tids lose to any even one (bit 0 is the most significant key): 1 vs bs - 2
    assert oracle.furthest_point_sampling(_tie_scene(bs, [1, bs - 2]), 2)[0].tolist() == [0, bs - 2]
    # tids 2 and 3 differ in bit 0 only after reversal ordering: 2 = ..010 -> 010.. ; 3 = ..011 -> 110.. : 2 wins
    assert oracle.furthest_point_sampling(_tie_scene(bs, [3, 2]), 2)[0].tolist() == [0, 2]
    # bs/4 (01 0..0 -> 0..0 10 = 2) against bs/2 + bs/4 (11 0..0 -> 0..0 11 = 3) against 6 (..110 -> 011..): bs/4 wins
    assert oracle.furthest_point_sampling(_tie_scene(bs, [6, half + bs // 4, bs // 4]), 2)[0].tolist() == [0, bs // 4]


def test_fps_tie_break_between_slots_of_a_thread_by_hand(oracle):
    """n = 1000 -> bs = 512 (cuda_utils.h:10-14): thread t holds k = t and k = t + 512 and keeps the FIRST maximum
    (strict '>', sampling_gpu.cu:136-137). Points 256 and 768 (both thread 256) at distance 1: the thread proposes its
    first, 256. Add point 512 -- thread 0's second point, its first one is the query itself at distance 0 -- and thread 0
    proposes 512; slot 0 beats every other slot on a tie."""
    n, bs = 1000, 512
    assert oracle.opt_n_threads(n) == bs
    assert oracle.furthest_point_sampling(_tie_scene(n, [256, 768]), 2)[0].tolist() == [0, 256]
    assert oracle.furthest_point_sampling(_tie_scene(n, [256, 768, 512]), 2)[0].tolist() == [0, 512]
    # 9-bit reversals: tid 200 = 011001000 -> 000100110 = 38; tid 130 = 010000010 -> itself; tid 5 -> 101000000 = 320;
    # tid 3 -> 110000000 = 384. Thread 200 holds k = 200 (distance 0) and k = 712 (distance 1): it proposes 712 and wins
    assert oracle.furthest_point_sampling(_tie_scene(n, [3, 5, 130, 200 + 512]), 2)[0].tolist() == [0, 712]


# ------------------------------------------------------------------------------------------------ rotated overlap

def rect_corners64(box):
    """(x1, y1, x2, y2, angle) -> 4 corners in float64, rotated about the centre as iou3d_kernel.cu:98-102 does"""
    x1, y1, x2, y2, a = [float(v) for v in box]
    cx, cy = (x1 + x2) / 2, (y1 + y2) / 2
    c, s = math.cos(a), math.sin(a)
    out = []
    for px, py in ((x1, y1), (x2, y1), (x2, y2), (x1, y2)):
        dx, dy = px - cx, py - cy
        out.append((dx * c + dy * s + cx, -dx * s + dy * c + cy))
    return out


def polygon_area(poly):
    return 0.5 * sum(poly[i][0] * poly[(i + 1) % len(poly)][1] - poly[(i + 1) % len(poly)][0] * poly[i][1] for i in range(len(poly)))


def clip_area64(box_a, box_b):
    """area of rectangle A clipped against the half-planes of rectangle B (Sutherland-Hodgman)"""
    subject, clip = rect_corners64(box_a), rect_corners64(box_b)
    if polygon_area(clip) < 0:
        clip = clip[::-1]
    for i in range(4):
        (ax, ay), (bx, by) = clip[i], clip[(i + 1) % 4]
        side = lambda p: (bx - ax) * (p[1] - ay) - (by - ay) * (p[0] - ax)   # noqa: E731  (>= 0: left of the edge, inside)
        out = []
        for j in range(len(subject)):
            p, q = subject[j], subject[(j + 1) % len(subject)]
            fp, fq = side(p), side(q)
            if (fp >= 0) != (fq >= 0):
                t = fp / (fp - fq)
                cross = (p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1]))
                if fp >= 0:
                    out.append(cross)
                else:
                    out.extend([cross, q])
            elif fq >= 0:
                out.append(q)
        subject = out
        if len(subject) < 3:
            return 0.0
    return abs(polygon_area(subject))


def random_bev_boxes(rng, k):
    cx, cy = rng.uniform(-3, 3, k), rng.uniform(-3, 3, k)
    w, l = rng.uniform(0.8, 5.0, k), rng.uniform(0.8, 5.0, k)
    ang = rng.uniform(-math.pi, math.pi, k)
    return np.stack([cx - w / 2, cy - l / 2, cx + w / 2, cy + l / 2, ang], 1).astype(np.float32)


def test_clip_area_on_known_shapes():
    unit = [-1, -1, 1, 1, 0.0]
    assert clip_area64(unit, unit) == pytest.approx(4.0)
    assert clip_area64(unit, [0, -1, 2, 1, 0.0]) == pytest.approx(2.0)
    assert clip_area64(unit, [-1, -1, 1, 1, math.pi / 4]) == pytest.approx(8 * (math.sqrt(2) - 1))      # regular octagon
    assert clip_area64(unit, [5, 5, 6, 6, 0.3]) == 0.0


def test_box_overlap_equals_float64_polygon_clip(oracle):
    """24 000 random pairs (more than 10 000 of them overlapping solidly): the oracle's float32 restatement of box_overlap against
    the float64 clip. Both agree to 1e-4 of the overlap (+ 2e-4 absolute: float32 corners of boxes ~10 units across);
    pairs whose contact is a sliver (overlap below 1 % of the smaller box) are compared absolutely only."""
    rng = np.random.default_rng(42)
    a, b = random_bev_boxes(rng, 24000), random_bev_boxes(rng, 24000)
    # row-by-row calls are slow: use the (N, M) form on blocks and take the diagonal
    got = np.concatenate([np.diagonal(oracle.boxes_overlap_bev(a[i:i + 200], b[i:i + 200])) for i in range(0, 24000, 200)]).astype(np.float64)
    want = np.array([clip_area64(a[i], b[i]) for i in range(24000)])
    small = np.minimum((a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]), (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]))
    solid = want > 0.01 * small
    assert solid.sum() > 10000 and (want == 0).sum() > 2000, (solid.sum(), (want == 0).sum())
    np.testing.assert_allclose(got[solid], want[solid], rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(got[~solid], want[~solid], rtol=0, atol=2e-3)
    assert (got[want == 0] == 0).all()


def test_box_overlap_analytic_rotated_rectangles(oracle):
    """rectangles whose intersection is known in closed form"""
    def overlap(p, q):
        return float(oracle.boxes_overlap_bev(np.array([p], np.float32), np.array([q], np.float32))[0, 0])
    # a 4 x 2 rectangle and the same turned by 90 degrees: the central 2 x 2 square
    assert overlap([-2, -1, 2, 1, 0.0], [-2, -1, 2, 1, math.pi / 2]) == pytest.approx(4.0, rel=1e-5)
    # a square and the same square turned by 45 degrees: regular octagon, 8 (sqrt 2 - 1) r^2 with r = half side
    assert overlap([-3, -3, 3, 3, 0.0], [-3, -3, 3, 3, math.pi / 4]) == pytest.approx(8 * (math.sqrt(2) - 1) * 9, rel=1e-5)
    # a long thin bar turned by theta across a wide one: parallelogram of area w1 * w2 / sin(theta)
    for theta in (math.pi / 6, math.pi / 3, 1.0):
        assert overlap([-20, -0.5, 20, 0.5, 0.0], [-20, -1, 20, 1, theta]) == pytest.approx(1.0 * 2.0 / math.sin(theta), rel=1e-4)
    # the turn direction does not matter for the area, the shift does: half overlap of a turned square with itself shifted
    s = math.sqrt(2)
    assert overlap([-1, -1, 1, 1, math.pi / 4], [-1 + s, -1, 1 + s, 1, math.pi / 4]) == pytest.approx(1.0, rel=1e-4)   # two diamonds (diagonal 2 sqrt 2) a half-diagonal apart: a square of diagonal sqrt 2
    # identical boxes at any angle: their own area (every corner is "inside", no proper edge crossing)
    for ang in (0.0, 0.3, -2.0):
        assert overlap([1, 2, 4, 8, ang], [1, 2, 4, 8, ang]) == pytest.approx(18.0, rel=1e-5)
    # disjoint
    assert overlap([0, 0, 1, 1, 0.2], [3, 3, 4, 4, -0.4]) == 0.0


def test_nms_suppression_chains_by_hand(oracle):
    """boxes sorted by score; unit-height strips [x, x + 1] x [0, 1] shifted by 0.3: IoU(i, i+1) = 0.7 / 1.3 = 0.538,
    IoU(i, i+2) = 0.4 / 1.6 = 0.25, IoU(i, i+3) = 0.1 / 1.9 = 0.053. Greedy NMS (iou3d.cpp:100-116: a box survives iff no
    KEPT earlier box overlaps it above the threshold -- a suppressed box suppresses nothing)."""
    boxes = np.array([[0.3 * i, 0.0, 0.3 * i + 1.0, 1.0, 0.0] for i in range(10)], np.float32)
    for rotated in (False, True):
        assert oracle.nms(boxes, 0.5, rotated).tolist() == [0, 2, 4, 6, 8]      # every second box: 1 falls to 0, 2 survives
        assert oracle.nms(boxes, 0.2, rotated).tolist() == [0, 3, 6, 9]         # 1, 2 fall to 0; 3 survives (0.053)
        assert oracle.nms(boxes, 0.6, rotated).tolist() == list(range(10))      # nothing reaches 0.6
        assert oracle.nms(boxes, 0.04, rotated).tolist() == [0, 4, 8]           # IoU(i, i+4) = 0
    # a chain across a 64-box tile boundary: 130 strips shifted by 0.3, threshold 0.5 keeps the even ones
    long = np.array([[0.3 * i, 0.0, 0.3 * i + 1.0, 1.0, 0.0] for i in range(130)], np.float32)
    assert oracle.nms(long, 0.5, False).tolist() == list(range(0, 130, 2))
    assert oracle.nms(long, 0.5, True).tolist() == list(range(0, 130, 2))


# ------------------------------------------------------------------------------------------------ the scan kernels, closed form
# The C oracle walks the reference's loops (one thread per centre / unknown, k = 0 .. n-1, early exit). Here the same results are
# derived from whole distance matrices with numpy: selections by sorting / masking instead of sequential scans. float32 arithmetic
# in the reference's association: (dx*dx + dy*dy) + dz*dz.

def _d2_matrix(a, b):
    """(len(a), len(b)) squared distances, float32, (ax-bx)*(ax-bx) + (ay-by)*(ay-by) + (az-bz)*(az-bz) left to right"""
    dx = a[:, None, 0] - b[None, :, 0]
    dy = a[:, None, 1] - b[None, :, 1]
    dz = a[:, None, 2] - b[None, :, 2]
    d = (dx * dx + dy * dy) + dz * dz
    assert d.dtype == np.float32
    return d


def ball_query_closed_form(radius, nsample, xyz, new_xyz):
    """ball_query_gpu.cu:9-45: the first nsample points with d2 < radius^2 in index order, the rest of the row filled with the first
    hit; a ball without a hit keeps the zeros its tensor was created with (pointnet2_utils.py:221)"""
    r2 = np.float32(radius) * np.float32(radius)
    out = np.zeros((xyz.shape[0], new_xyz.shape[1], nsample), np.int32)
    for b in range(xyz.shape[0]):
        inside = _d2_matrix(new_xyz[b], xyz[b]) < r2
        order = np.argsort(~inside, axis=1, kind="stable")[:, :nsample]          # hits first, each group in index order
        count = inside.sum(axis=1)
        first = order[:, 0]
        row = np.where(np.arange(nsample)[None, :] < count[:, None], order, first[:, None])
        out[b] = np.where(count[:, None] > 0, row, 0)
    return out


def three_nn_closed_form(unknown, known):
    """interpolate_gpu.cu:30-48: a strict-'<' insertion over k = 0 .. m-1 keeps the three smallest (d, k) pairs in
    lexicographic order"""
    bsz, n, m = unknown.shape[0], unknown.shape[1], known.shape[1]
    d_out = np.empty((bsz, n, 3), np.float32)
    i_out = np.empty((bsz, n, 3), np.int32)
    for b in range(bsz):
        d = _d2_matrix(unknown[b], known[b])
        order = np.argsort(d, axis=1, kind="stable")[:, :3]                      # stable: equal distances by index
        d_out[b] = np.take_along_axis(d, order, axis=1)
        i_out[b] = order
    return d_out, i_out


@pytest.mark.parametrize("n,m,radius,ns,kind", [(3000, 400, 0.8, 16, "kitti"), (2048, 256, 0.3, 32, "kitti"), (500, 77, 50.0, 8, "ubox"),
                                                (1500, 300, 1e-4, 4, "dup"), (700, 64, 3.0, 64, "lattice")])
def test_ball_query_scan_equals_the_closed_form(oracle, n, m, radius, ns, kind):
    from epnet_amd import synth
    xyz = (np.stack([clouds_with_duplicates(n, 7), clouds_with_duplicates(n, 8)]) if kind == "lattice"
           else synth.scenes(kind, 2, n, seed=31).numpy())
    rng = np.random.default_rng(n)
    centres = np.ascontiguousarray(xyz[:, rng.permutation(n)[:m]]) + (np.float32(0.01) if kind == "kitti" else np.float32(0))
    got = oracle.ball_query(radius, ns, xyz, centres)
    np.testing.assert_array_equal(got, ball_query_closed_form(radius, ns, xyz, centres))
    if kind == "dup":                       # a tiny ball around a duplicated point: exactly the twins, padded with the first
        assert (got[:, :, 0] >= 0).all() and ((got == got[:, :, :1]).any(axis=2)).all()


@pytest.mark.parametrize("n,m,kind", [(2000, 500, "kitti"), (1024, 64, "ubox"), (900, 300, "lattice"), (300, 3, "kitti")])
def test_three_nn_scan_equals_the_closed_form(oracle, n, m, kind):
    from epnet_amd import synth
    src = (np.stack([clouds_with_duplicates(n + m, 5), clouds_with_duplicates(n + m, 6)]) if kind == "lattice"
           else synth.scenes(kind, 2, n + m, seed=17).numpy())
    unknown, known = np.ascontiguousarray(src[:, :n]), np.ascontiguousarray(src[:, n:])
    d2, idx = oracle.three_nn(unknown, known)
    want_d, want_i = three_nn_closed_form(unknown, known)
    np.testing.assert_array_equal(idx, want_i)              # lattice clouds: many equal distances, the smaller index first
    np.testing.assert_array_equal(d2, want_d)


def test_gather_group_interpolate_equal_numpy_indexing(oracle):
    rng = np.random.default_rng(3)
    b, c, n, m, ns = 2, 7, 300, 40, 5
    feats = rng.standard_normal((b, c, n)).astype(np.float32)
    idx = rng.integers(0, n, size=(b, m, ns)).astype(np.int32)
    want = np.stack([feats[s][:, idx[s]] for s in range(b)])                      # (b, c, m, ns)
    np.testing.assert_array_equal(oracle.group_points(feats, idx), want)
    np.testing.assert_array_equal(oracle.gather_points(feats, idx[:, :, 0].copy()), want[:, :, :, 0])
    grad = rng.standard_normal((b, c, m, ns)).astype(np.float32)
    acc = np.zeros((b, c, n), np.float64)
    for s in range(b):
        for ch in range(c):
            np.add.at(acc[s, ch], idx[s].ravel(), grad[s, ch].ravel().astype(np.float64))
    np.testing.assert_allclose(oracle.group_points_grad(grad, idx, n), acc, rtol=1e-5, atol=1e-5)
    nn = rng.integers(0, n, size=(b, m, 3)).astype(np.int32)
    w = rng.random((b, m, 3)).astype(np.float32)
    p = np.stack([feats[s][:, nn[s]] for s in range(b)])                          # (b, c, m, 3)
    want_i = (p[..., 0] * w[:, None, :, 0] + p[..., 1] * w[:, None, :, 1]) + p[..., 2] * w[:, None, :, 2]   # interpolate_gpu.cu:86-106
    np.testing.assert_array_equal(oracle.three_interpolate(feats, nn, w), want_i.astype(np.float32))
    gout = rng.standard_normal((b, c, m)).astype(np.float32)
    acc = np.zeros((b, c, n), np.float64)
    for s in range(b):
        for ch in range(c):
            for k in range(3):
                np.add.at(acc[s, ch], nn[s, :, k], (gout[s, ch] * w[s, :, k]).astype(np.float64))
    np.testing.assert_allclose(oracle.three_interpolate_grad(gout, nn, w, n), acc, rtol=1e-5, atol=1e-5)
