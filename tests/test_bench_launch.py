"""`bench.py --gpus N` / `bench_step.py --gpus N` start N ranks themselves: the decision is a pure
function (scene_shard.launch_plan), and the whole launch is rehearsed on the CPU with the gloo backend."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launch_plan_decisions():
    from epnet_amd import scene_shard as ss
    assert ss.launch_plan(1, "bench.py", [], environ={}) == ("run", None)
    assert ss.launch_plan(8, "bench.py", ["--gpus", "8"], environ={"WORLD_SIZE": "8"}) == ("run", None)
    action, cmd = ss.launch_plan(8, "/x/bench.py", ["--gpus", "8", "--steps", "5"], environ={}, executable="py", port=1234)
    assert action == "spawn"
    assert cmd == ["py", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                   "--master-port", "1234", "/x/bench.py", "--gpus", "8", "--steps", "5"]
    with pytest.raises(ValueError):        # a torchrun world that contradicts the flag is a wrong measurement, not a default
        ss.launch_plan(8, "bench.py", [], environ={"WORLD_SIZE": "2"})
    with pytest.raises(ValueError):
        ss.launch_plan(1, "bench.py", [], environ={"WORLD_SIZE": "2"})
    with pytest.raises(ValueError):
        ss.launch_plan(0, "bench.py", [], environ={})


def _run(script, extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(EPNET_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1")
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, script)] + extra, env=env, capture_output=True, text=True, timeout=280)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("script", ["bench.py", "bench_step.py"])
def test_gpus_flag_starts_that_many_ranks(script):
    r = _run(script, ["--gpus", "2", "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["gpus_flag"] == 2
    # every rank reports the device it computes on; CPU (gloo) ranks never pass for distinct GPUs
    assert len(line["devices"]) == 2 and len(set(line["devices"])) == 2 and line["devices_distinct"] == 0


@pytest.mark.timeout(120)
def test_world_contradicting_flag_is_refused():
    r = _run("bench.py", ["--gpus", "4", "--launch-check"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 4" in (r.stderr + r.stdout)


@pytest.mark.timeout(120)
@pytest.mark.parametrize("script", ["bench.py", "bench_step.py"])
def test_rehearsal_switches_need_the_rehearsal_flag(script):
    """EPNET_BENCH_DEVICE can put every rank on one GPU: a measurement run refuses it (before touching any device)"""
    r = _run(script, [], {"EPNET_BENCH_DEVICE": "0"})
    assert r.returncode != 0 and "--rehearsal" in (r.stderr + r.stdout)


def test_distinct_devices_counts_gpus_not_ranks():
    from epnet_amd import scene_shard as ss
    ids = ["h/gpu/uuid=a", "h/gpu/uuid=a", "h/gpu/uuid=b", "h/cpu/pid7"]
    assert ss.distinct_devices(ids) == 2
    assert ss.gather_over_ranks("x") == ["x"] and "/cpu/" in ss.device_identity("cpu")
