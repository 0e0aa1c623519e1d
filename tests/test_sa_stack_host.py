"""Host logic of epnet_amd.sa_stack (no kernel is launched: buffers on the CPU): which buffer belongs to which batch after a step
(owners / owners3: what bench.verify_scene and tests/test_stream_of_batches.py rely on), the ring sizes of the two- and three-stage
schedules, where the ball queries are placed, and the compulsory-byte formulas of SURVEY.md section 8(d)."""
import pytest


@pytest.fixture()
def stack_of(hiplib):
    from epnet_amd import sa_stack

    def make(**kw):
        kw.setdefault("n", 2048)
        kw.setdefault("npoints", (512, 128, 32, 8))
        return sa_stack.SAStack(2, device="cpu", **kw)
    return make


def test_algorithmic_bytes_of_the_metric_configuration():
    from epnet_amd import sa_stack
    by = sa_stack.sa_algorithmic_bytes()
    assert by == {"fps": 282880, "gather": 348160, "ball_query": 1697280, "group_xyz": 4700160, "group_feat": 44298240, "total": 51326720}
    fp = sa_stack.fp_algorithmic_bytes()
    assert by["total"] + fp["total"] == 88087040          # the figure the `with_fp` sub-line of the bench quotes
    cfg5 = sa_stack.CONFIGS[5]
    assert sa_stack.sa_algorithmic_bytes(cfg5["n"], cfg5["npoints"], cfg5["nsamples"], cfg5["feat_channels"])["total"] == 314048512


def test_rings_and_owners(stack_of):
    plain = stack_of()
    assert (plain.stages, plain.ring, plain.owners()) == (1, 1, (0, 0)) and plain.s_query_levels == frozenset()
    assert plain.static_xyz is None and all(len(L["sets"]) == 1 for L in plain.levels)

    two = stack_of(pipelined=True, fused_sampling=True)
    assert (two.stages, two.ring) == (2, 2) and all(len(L["sets"]) == 2 for L in two.levels)
    assert all(L["fps_idx_sets"][0] is L["fps_idx_sets"][1] is L["fps_idx"] for L in two.levels)   # one sampling chain writes all levels
    two.replays = 1                     # after step 0: sampled into set 0, grouped set 1
    assert two.owners() == (0, 1)
    two.replays = 4                     # after step 3
    assert two.owners() == (1, 0)
    with pytest.raises(RuntimeError):
        stack_of(pipelined=True, fused_sampling=True, stages=3).owners()

    three = stack_of(pipelined=True, fused_sampling=True, stages=3)
    assert (three.stages, three.ring) == (3, 3) and three.s_query_levels == frozenset(range(4))
    assert all(len(L["sets"]) == 3 and len({id(t) for t in L["fps_idx_sets"]}) == 3 for L in three.levels)
    assert all(len(S["idx_sets"]) == 3 for L in three.levels for S in L["scales"])
    three.replays = 1
    assert three.owners3() == (0, 2, 1)                    # step 0 sampled slot 0; slots 2 / 1 hold the (primed) batches before it
    three.replays = 6                                       # after step 5: S1 slot 2, S2 slot 1, G slot 0
    assert three.owners3() == (2, 1, 0)
    # three stages need the fused sampling chain over shared indices: anything else falls back to two
    assert stack_of(pipelined=True, fused_sampling=False, stages=3).stages == 2
    assert stack_of(pipelined=True, fused_sampling=True, shared_index=False, stages=3).stages == 2
    assert stack_of(pipelined=False, stages=3).stages == 1


def test_where_the_ball_queries_run(stack_of, monkeypatch):
    monkeypatch.delenv("EPNET_SA_S_QUERY_LEVELS", raising=False)
    monkeypatch.delenv("EPNET_SA_QUERIES_IN_S", raising=False)
    small = stack_of(pipelined=True, fused_sampling=True)                       # 2 x 2048 points: stage S is the longer one
    assert small.s_query_levels == frozenset()
    assert stack_of(pipelined=True, fused_sampling=True, with_fp=True, fp=((8, 8, 32), (8, 32, 128), (8, 128, 512), (8, 512, 2048))).s_query_levels == frozenset(range(4))
    assert stack_of(pipelined=True, fused_sampling=True, s_query_levels=(1, 3)).s_query_levels == frozenset({1, 3})
    monkeypatch.setenv("EPNET_SA_S_QUERY_LEVELS", "2")
    assert stack_of(pipelined=True, fused_sampling=True).s_query_levels == frozenset({2})
    # per-slot index tensors only for the levels whose queries run in stage S
    picked = stack_of(pipelined=True, fused_sampling=True, s_query_levels=(1,))
    assert [len(S["idx_sets"]) for L in picked.levels for S in L["scales"]] == [1, 1, 2, 2, 1, 1, 1, 1]


def test_a_pipelined_stack_has_no_single_input_buffer(stack_of):
    import torch
    two = stack_of(pipelined=True, fused_sampling=True)
    two.inputs = [torch.zeros(1), torch.ones(1)]
    with pytest.raises(AttributeError):
        two.static_xyz
    assert two.input_buffer(0) is two.inputs[0] and two.input_buffer(3) is two.inputs[1] and two.input_buffer() is two.inputs[0]
