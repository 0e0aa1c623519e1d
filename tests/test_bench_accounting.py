"""bench.py's per-launch byte formulas add up to SURVEY.md section 8(d)'s per-scene figure (51 326 720 B for one
16384-point scene through the SA op stack), whichever way the stack issues its calls."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _calls(fused, multi):
    from epnet_amd import sa_stack
    calls = []
    n = 16384
    for lvl, m in enumerate(sa_stack.RPN_NPOINTS):
        c = sa_stack.RPN_FEAT_CHANNELS[lvl]
        calls.append(("furthest_point_sampling_indexed_wrapper", (1, n, m)))
        calls.append(("gather_points_wrapper", (1, 3, n, m)))
        radii, nss = sa_stack.RPN_RADII[lvl], sa_stack.RPN_NSAMPLES[lvl]
        if multi == "ordered":     # what the stack issues since round 3: the same bytes under the ordered entry point's name
            calls.append(("ball_query_ordered_wrapper", (1, n, m, list(radii), list(nss))))
        elif multi:
            calls.append(("ball_query_multi_wrapper", (1, n, m, list(radii), list(nss))))
        else:
            calls += [("ball_query_indexed_wrapper", (1, n, m, r, ns)) for r, ns in zip(radii, nss)]
        if fused == "multi":
            calls.append(("group_concat_multi_wrapper", (1, c, n, m, list(nss))))
            n = m
            continue
        for ns in nss:
            if fused:
                calls.append(("group_concat_wrapper", (1, c, n, m, ns)))
            else:
                calls.append(("group_points_wrapper", (1, 3, n, m, ns)))
                if c:
                    calls.append(("group_points_wrapper", (1, c, n, m, ns)))
        n = m
    return calls


def test_launch_bytes_sum_to_the_survey_figure():
    import bench
    from epnet_amd import sa_stack
    want = sa_stack.sa_algorithmic_bytes(16384)
    assert want["total"] == 51326720
    for fused in (False, True, "multi"):
        for multi in (False, True, "ordered"):
            total = sum(bench.op_family(name, head)[1] for name, head in _calls(fused, multi))
            assert total == want["total"], (fused, multi, total)


def test_scene_index_is_not_counted_as_algorithmic():
    import bench
    label, nbytes = bench.op_family("scene_index_build_wrapper", (1, 16384))
    assert label.startswith("scene_index") and nbytes > 0   # reported per launch, but not part of the stack total


def test_dominant_family_is_stable_when_sampling_and_grouping_trade_places():
    """roofline names the longest family of a step; within 10 % of the longest the one that moves the most bytes (the level-1
    sampling, a latency chain of 72 MB, and the grouping calls, 13 GB, are 2.09 - 2.22 ms each from run to run)"""
    import bench
    def fam(step_ms, launches, nbytes):
        return {"step_ms": step_ms, "launches_per_step": launches, "bytes_per_launch": nbytes, "avg_ms": step_ms / launches, "GBps": 0.0}
    k = {"fps N=16384 M=4096": fam(2.116, 1, 72_000_000), "group": fam(2.090, 4, 3_100_000_000), "ball_query x": fam(0.45, 1, 400_000_000)}
    assert bench.pick_dominant(k) == ("fps N=16384 M=4096", "group")
    k["group"] = fam(2.224, 4, 3_100_000_000)
    assert bench.pick_dominant(k) == ("group", "group")
    k["group"] = fam(1.2, 4, 3_100_000_000)                   # far behind: the longest family is named, whatever it moves
    assert bench.pick_dominant(k) == ("fps N=16384 M=4096", "fps N=16384 M=4096")
