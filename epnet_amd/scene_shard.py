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


def launch_plan(gpus, script, argv, environ=None, executable=None, port=None):
    """What a bench entry point started as ``python <script> --gpus N ...`` has to do, decided BEFORE anything touches
    the GPU (the reference's multi-GPU entry is one process driving N devices, tools/train_rcnn.py:221-223; here it is one
    process per GPU):

      ("run", None)    this process is a rank: either torch.distributed.run started it (WORLD_SIZE set and equal to
                       --gpus) or a single GPU was asked for;
      ("spawn", cmd)   --gpus N > 1 without a torch.distributed.run environment: start `cmd` (the same script under
                       ``python -m torch.distributed.run --nproc-per-node N``) as a CHILD process and exit with its
                       code -- never exec: a process that has initialised the GPU must not be replaced.

    A torch.distributed.run environment whose WORLD_SIZE contradicts --gpus raises ValueError (a line claiming N GPUs
    measured on another number of ranks would be a wrong measurement)."""
    import sys
    environ = os.environ if environ is None else environ
    if gpus < 1:
        raise ValueError("--gpus must be >= 1, got %d" % gpus)
    if "WORLD_SIZE" in environ:
        world = int(environ["WORLD_SIZE"])
        if world != gpus:
            raise ValueError("--gpus %d but torch.distributed.run started %d ranks (WORLD_SIZE)" % (gpus, world))
        return "run", None
    if gpus == 1:
        return "run", None
    if port is None:
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    cmd = [executable or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(argv)
    return "spawn", cmd


def launch_or_continue(gpus, script, argv):
    """bench entry points call this first. Returns None when this process should go on as a rank; otherwise the ranks ran
    as child processes and the caller exits with the returned code."""
    import subprocess
    action, cmd = launch_plan(gpus, script, argv)
    if action == "run":
        return None
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def assert_world(gpus):
    """every rank: the world torch.distributed.run built is the one --gpus names"""
    _rank, _local, world = env_world()
    if world != gpus:
        raise SystemExit("--gpus %d but this process runs in a world of %d ranks" % (gpus, world))
    return world


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


def device_identity(device=None):
    """a string that names the physical device this rank computes on: host + PCI address / UUID of the torch device (a rank
    without a GPU -- the gloo rehearsal -- names host + process, so that such ranks never pass for distinct GPUs)"""
    import socket
    host = socket.gethostname()
    if device is None or str(device) == "cpu":
        return "%s/cpu/pid%d" % (host, os.getpid())
    import torch
    prop = torch.cuda.get_device_properties(device)
    parts = []
    for name in ("uuid", "pci_domain_id", "pci_bus_id", "pci_device_id"):
        v = getattr(prop, name, None)
        if v is not None:
            parts.append("%s=%s" % (name, v))
    if not parts:   # no identifying property on this build: the device ordinal is all there is
        parts.append("ordinal=%d" % torch.device(device).index)
    return "%s/gpu/%s" % (host, ",".join(parts))


def gather_over_ranks(obj):
    """every rank's python object, in rank order (a list of one for a single process)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def distinct_devices(identities):
    """how many different physical GPUs a list of device_identity() strings names (CPU ranks count as none)"""
    return len({i for i in identities if "/gpu/" in i})
