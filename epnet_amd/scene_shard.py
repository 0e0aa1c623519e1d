"""Scene-parallel sharding across ranks (one process per GPU).

Every op on the hot path is independent per scene (the batch index is a grid dimension of every
reference kernel; roipool3d and NMS are per scene), so multi-GPU operation needs NO collective in the
data path: rank r simply owns scenes r, r + world, r + 2*world, ... of the global batch (the assignment
a DistributedSampler makes). The only exchanges are bookkeeping: a barrier around timed regions and a
MAX-reduction of elapsed time; a gradient all-reduce exists only in the end-to-end training step
(DDP over RCCL), outside the op stack.
"""
import os


def env_world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment (1 process: 0, 0, 1)"""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def scene_ids(total_scenes, rank, world):
    """global scene ids owned by `rank`: round-robin, every scene owned exactly once"""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    return list(range(rank, total_scenes, world))


def scene_seed(base_seed, scene_id):
    """seed of a synthetic scene: a function of the GLOBAL scene id, so the union of all ranks' scenes
    does not depend on how many ranks there are"""
    return base_seed + 7919 * scene_id


def init_process_group(backend, device=None):
    import torch.distributed as dist
    rank, _local, world = env_world()
    if world == 1 or dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver
    kw = {"device_id": device} if (device is not None and backend == "nccl") else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    """MAX of a python float over all ranks (identity for a single process)"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
