"""The software-pipelined SA stack on a STREAM of different batches (VERDICT r02 "what's missing" 1): the reference consumes a new
batch every iteration (tools/train_rcnn.py:221-223, lib/net/train_functions.py), so stage S of step k (sampling batch k) runs
beside stage G of step k-1's batch -- each from its own resident input buffer (SAStack.inputs[parity]). After step k+1 every tensor
of batch k is complete; bench.verify_scene holds each buffer to the batch that owns it (SAStack.owners()), against the oracle.
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


def _twin_cloud(oracle, n, seed):
    """a kitti-like cloud with an exact twin of the point picked in round 300 written over a never-picked point: a tie in the
    middle of the level-1 rounds, so the deeper levels resume their rounds from a known prefix"""
    from epnet_amd import synth
    c = synth.kitti_like_cloud(n, seed).numpy()
    seq = oracle.furthest_point_sampling(c[None], 4096)[0]
    never = np.setdiff1d(np.arange(n), seq)[7]
    c[never] = c[seq[300]]
    return torch.from_numpy(c)


def _batches(oracle, b, n):
    from epnet_amd import synth
    kinds = ["kitti", "dup", "ubox", "twin", "kitti_q"]
    out = []
    for k, kind in enumerate(kinds):
        if kind == "twin":
            out.append(torch.stack([_twin_cloud(oracle, n, 900 + 10 * k + s) for s in range(b)]))
        else:
            out.append(synth.scenes(kind, b, n, seed=900 + 10 * k))
    return kinds, [x.to(DEV) for x in out]


def _poison(stack):
    """nothing of the capture-time warm-up may pass for a result (indices stay valid: the next step's grouping reads them)"""
    for L in stack.levels:
        L["fps_idx"].fill_(-1)
        for S in L["scales"]:
            S["grouped"].fill_(float("nan"))
    for F in stack.fp_bufs:
        F["out"].fill_(float("nan"))


@pytest.mark.parametrize("with_fp,in_s", [(False, (1, 2, 3)), (False, ()), (True, None), (True, ())])
def test_pipelined_graphs_on_a_stream_of_different_batches(oracle, with_fp, in_s):
    import bench
    from epnet_amd import sa_stack
    b, n = 2, 16384
    kinds, batches = _batches(oracle, b, n)
    stack = sa_stack.SAStack(b, n=n, device=DEV, with_fp=with_fp, seed=5, pipelined=True, fused_sampling=True, s_query_levels=in_s)
    stack.capture(batches[0])
    _poison(stack)
    stack.replay(batches[0])
    for k in range(1, len(batches)):
        stack.replay(batches[k])               # samples batch k, groups batch k-1
        torch.cuda.synchronize()
        for scene in range(b):
            assert bench.verify_scene(stack, batches[k], scene, prev_xyz=batches[k - 1]) == [], (kinds[k - 1], kinds[k], scene)
    # the check tells the batches apart: held to the wrong "previous" batch, the grouping stage's outputs do not verify
    wrong = bench.verify_scene(stack, batches[-1], 0, prev_xyz=batches[0])
    assert any(name.endswith("grouped") for name in wrong) and "level1.fps_idx" not in wrong
    # ... and the sampling stage's do not verify against the wrong "current" one
    wrong = bench.verify_scene(stack, batches[0], 0, prev_xyz=batches[-2])
    assert "level1.fps_idx" in wrong and not any(name.endswith("grouped") for name in wrong)


@pytest.mark.parametrize("with_fp,in_s", [(False, None), (True, None), (False, (1, 2, 3))])
def test_three_stage_graphs_on_a_stream_of_different_batches(oracle, with_fp, in_s):
    """stages=3: level-1 sampling of batch k beside the rest of the sampling chain + queries of batch k-1 beside the grouping of
    batch k-2, three graphs over a ring of three buffer sets. After step k batch k-2 is complete; the slots of the younger
    batches hold what their stages have written so far (bench.verify_scene knows which: SAStack.owners3())"""
    import bench
    from epnet_amd import sa_stack
    b, n = 2, 16384
    kinds, batches = _batches(oracle, b, n)
    stack = sa_stack.SAStack(b, n=n, device=DEV, with_fp=with_fp, seed=5, pipelined=True, fused_sampling=True, s_query_levels=in_s, stages=3)
    assert stack.stages == 3 and stack.ring == 3 and len(stack.levels[0]["sets"]) == 3
    assert stack.s_query_levels == (frozenset(range(4)) if in_s is None else frozenset(in_s))
    stack.capture(batches[0])
    _poison(stack)
    stack.replay(batches[0])
    stack.replay(batches[1])
    for k in range(2, len(batches)):
        stack.replay(batches[k])               # samples level 1 of batch k, finishes the chain of batch k-1, groups batch k-2
        torch.cuda.synchronize()
        for scene in range(b):
            bad = bench.verify_scene(stack, batches[k], scene, prev_xyz=batches[k - 1], prev2_xyz=batches[k - 2])
            assert bad == [], (kinds[k - 2], kinds[k - 1], kinds[k], scene, bad)
    # the check tells the three batches apart
    k = len(batches) - 1
    wrong = bench.verify_scene(stack, batches[k], 0, prev_xyz=batches[k - 1], prev2_xyz=batches[0])
    assert any(name.endswith("grouped") for name in wrong) and not any(name.startswith("level1.fps_idx[set %d]" % (k % 3)) for name in wrong)
    wrong = bench.verify_scene(stack, batches[0], 0, prev_xyz=batches[k - 1], prev2_xyz=batches[k - 2])
    assert wrong == ["level1.fps_idx[set %d]" % (k % 3)]       # all a slot holds of the youngest batch is its level-1 sampling


def test_three_stage_eager_steps(oracle):
    import bench
    from epnet_amd import sa_stack
    b, n = 1, 16384
    kinds, batches = _batches(oracle, b, n)
    stack = sa_stack.SAStack(b, n=n, device=DEV, with_fp=True, seed=5, pipelined=True, fused_sampling=True, stages=3)
    stack.step(batches[0])
    stack.step(batches[1])
    for L in stack.levels:               # (stage G's outputs only: the sampling stages of these batches have run already)
        for S in L["scales"]:
            S["grouped"].fill_(float("nan"))
    for k in range(2, 5):
        stack.step(batches[k])
        torch.cuda.synchronize()
        assert bench.verify_scene(stack, batches[k], 0, prev_xyz=batches[k - 1], prev2_xyz=batches[k - 2]) == [], kinds[k - 2]


def test_three_stages_on_one_level_config5(oracle):
    """BASELINE config 5 (one level, 65536 points): S1 = sampling, S2 = the ball query, G = the grouping"""
    import bench
    from epnet_amd import sa_stack, synth
    cfg = sa_stack.CONFIGS[5]
    b = 1
    batches = [synth.scenes(kind, b, cfg["n"], seed=70 + i).to(DEV) for i, kind in enumerate(("kitti", "ubox", "kitti", "kitti_q"))]
    stack = sa_stack.SAStack(b, n=cfg["n"], device=DEV, npoints=cfg["npoints"], radii=cfg["radii"], nsamples=cfg["nsamples"],
                             feat_channels=cfg["feat_channels"], seed=6, pipelined=True, fused_sampling=True, stages=3)
    stack.capture(batches[0])
    _poison(stack)
    for k in range(len(batches)):
        stack.replay(batches[k])
        torch.cuda.synchronize()
        if k >= 2:
            assert bench.verify_scene(stack, batches[k], 0, prev_xyz=batches[k - 1], prev2_xyz=batches[k - 2]) == []


def test_loader_fills_the_input_buffer_in_place(oracle):
    """no copy in replay(): the producer writes the next batch into stack.input_buffer() (stream-ordered behind the last replay)"""
    import bench
    from epnet_amd import sa_stack
    b, n = 1, 16384
    kinds, batches = _batches(oracle, b, n)
    stack = sa_stack.SAStack(b, n=n, device=DEV, seed=5, pipelined=True, fused_sampling=True)
    stack.capture(batches[0], batches[1])
    assert stack.input_buffer(0) is stack.inputs[0] and stack.input_buffer(1) is stack.inputs[1] and stack.input_buffer() is stack.inputs[0]
    _poison(stack)
    stack.replay()                              # samples inputs[0] = batch 0, groups inputs[1] = batch 1 (primed by capture)
    torch.cuda.synchronize()
    assert bench.verify_scene(stack, batches[0], 0, prev_xyz=batches[1]) == []
    stack.input_buffer().copy_(batches[2])      # parity 1's buffer: its last reader (the grouping of replay 0) is behind us on the stream
    stack.replay()
    torch.cuda.synchronize()
    assert bench.verify_scene(stack, batches[2], 0, prev_xyz=batches[0]) == []


def test_unfused_composition_on_a_stream_of_different_batches(oracle):
    """--unfused (the reference's op-by-op grouping: flipped cloud, group_points, no centre subtraction) pipelined: the flipped
    cloud stage S writes is read by stage G one step later, so it is double-buffered like the centres (the bench's flag sweep of
    round 3 found it single-buffered: level-1 grouped_xyz of the wrong batch)"""
    import bench
    from epnet_amd import sa_stack
    b, n = 1, 16384
    kinds, batches = _batches(oracle, b, n)
    stack = sa_stack.SAStack(b, n=n, device=DEV, seed=5, pipelined=True, fused_sampling=True, fused=False)
    stack.capture(batches[0])
    stack.replay(batches[0])
    for k in range(1, 4):
        stack.replay(batches[k])
        torch.cuda.synchronize()
        assert bench.verify_scene(stack, batches[k], 0, prev_xyz=batches[k - 1]) == [], (kinds[k - 1], kinds[k])


@pytest.mark.parametrize("with_fp", [False, True])
def test_eager_pipelined_steps_on_different_batches(oracle, with_fp):
    import bench
    from epnet_amd import sa_stack
    b, n = 2, 16384
    kinds, batches = _batches(oracle, b, n)
    stack = sa_stack.SAStack(b, n=n, device=DEV, with_fp=with_fp, seed=5, pipelined=True, fused_sampling=True)
    stack.step(batches[0])
    _poison(stack)
    for k in range(1, 4):
        stack.step(batches[k])
        torch.cuda.synchronize()
        assert bench.verify_scene(stack, batches[k], 1, prev_xyz=batches[k - 1]) == [], (kinds[k - 1], kinds[k])


def test_unpipelined_graph_takes_a_new_batch_per_replay(oracle):
    import bench
    from epnet_amd import sa_stack
    b, n = 1, 16384
    kinds, batches = _batches(oracle, b, n)
    stack = sa_stack.SAStack(b, n=n, device=DEV, seed=5, pipelined=False, fused_sampling=True, with_fp=True)
    stack.capture(batches[0])
    for k in (1, 3):
        _poison(stack)
        stack.replay(batches[k])
        torch.cuda.synchronize()
        assert bench.verify_scene(stack, batches[k], 0) == [], kinds[k]
