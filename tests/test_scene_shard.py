"""N > 1 path on the CPU: two gloo ranks shard a global batch of scenes with no data-path collective."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from epnet_amd import scene_shard, synth
    scene_shard.init_process_group("gloo")
    ids = scene_shard.scene_ids(total, rank, world)
    clouds = [synth.kitti_like_cloud(256, scene_shard.scene_seed(5, i)) for i in ids]
    checksum = float(sum(c.double().sum() for c in clouds))
    scene_shard.barrier()
    slowest = scene_shard.max_over_ranks(1.0 + rank)           # the bench's timing reduction
    total_points = scene_shard.sum_over_ranks(256 * len(ids))
    all_sum = scene_shard.sum_over_ranks(checksum)
    torch.save({"ids": ids, "slowest": slowest, "total_points": total_points, "all_sum": all_sum},
               os.path.join(out_dir, "r%d.pt" % rank))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_shard_scenes(tmp_path):
    from epnet_amd import scene_shard, synth
    world, total = 2, 7
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, "r%d.pt" % r)) for r in range(world)]
    assert sorted(res[0]["ids"] + res[1]["ids"]) == list(range(total))          # every scene exactly once
    assert not set(res[0]["ids"]) & set(res[1]["ids"])
    assert res[0]["slowest"] == res[1]["slowest"] == 2.0                         # MAX over ranks
    assert res[0]["total_points"] == 256 * total
    # the union of the shards is the same data a single rank would generate
    single = float(sum(synth.kitti_like_cloud(256, scene_shard.scene_seed(5, i)).double().sum() for i in range(total)))
    assert res[0]["all_sum"] == pytest.approx(single, rel=1e-12)


def test_single_process_helpers():
    from epnet_amd import scene_shard
    assert scene_shard.scene_ids(5, 0, 1) == [0, 1, 2, 3, 4]
    assert scene_shard.scene_ids(5, 3, 4) == [3] and scene_shard.scene_ids(2, 3, 4) == []
    with pytest.raises(ValueError):
        scene_shard.scene_ids(5, 4, 4)
    assert scene_shard.max_over_ranks(3.5) == 3.5
