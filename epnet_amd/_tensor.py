"""Tensor -> raw pointer plumbing shared by the three extension stand-ins."""
import torch


def _bad(name, why):
    raise RuntimeError("%s %s" % (name, why))


def dev_ptr(t, name, dtype):
    """device address of a contiguous GPU tensor of the given dtype (RuntimeError otherwise, as the
    reference's CHECK_CUDA / CHECK_CONTIGUOUS do -- e.g. lib/utils/iou3d/src/iou3d.cpp:7-9 -- plus a
    dtype check the reference lacks: there a wrong dtype is undefined behaviour)."""
    if not isinstance(t, torch.Tensor):
        _bad(name, "must be a torch.Tensor")
    if not t.is_cuda:
        _bad(name, "must be a CUDAtensor (epnet_amd has no CPU fallback)")
    if not t.is_contiguous():
        _bad(name, "must be contiguous")
    if t.dtype != dtype:
        _bad(name, "must have dtype %s, got %s" % (dtype, t.dtype))
    return t.data_ptr()


def host_ptr(t, name, dtype):
    if not isinstance(t, torch.Tensor):
        _bad(name, "must be a torch.Tensor")
    if t.is_cuda:
        _bad(name, "must be a CPU tensor")
    if not t.is_contiguous():
        _bad(name, "must be contiguous")
    if t.dtype != dtype:
        _bad(name, "must have dtype %s, got %s" % (dtype, t.dtype))
    return t.data_ptr()


def need(t, numel, name):
    if t.numel() < numel:
        _bad(name, "has %d elements, the call needs %d" % (t.numel(), numel))


class on_device_of:
    """Makes the tensor's device current for the launch and yields its current HIP stream.

    The ops are called from several host threads with different current devices under
    nn.DataParallel (tools/train_rcnn.py:221-223), so device and stream are always derived from
    the tensor, never from global state."""

    __slots__ = ("dev", "prev")

    def __init__(self, t):
        self.dev = t.device.index
        self.prev = None

    def __enter__(self):
        cur = torch.cuda.current_device()
        if cur != self.dev:
            self.prev = cur
            torch.cuda.set_device(self.dev)
        return torch.cuda.current_stream(self.dev).cuda_stream

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.cuda.set_device(self.prev)
        return False


def writes(*names):
    """decorator of an extension stand-in: the named arguments (tensors, lists of tensors or None) are OUTPUTS the kernels
    fill through ``data_ptr()``. The reference's pybind extensions write the same way, invisibly to autograd; here every
    output's version counter is moved after the launch, so that anything remembered about the old contents (autograd's
    saved-tensor check, pointnet2_utils.scene_index) is known to be stale."""
    import functools
    import inspect

    def deco(fn):
        params = list(inspect.signature(fn).parameters)
        slots = [(params.index(n), n) for n in names]

        @functools.wraps(fn)
        def wrapped(*args, **kwargs):
            result = fn(*args, **kwargs)
            for pos, name in slots:
                out = args[pos] if pos < len(args) else kwargs.get(name)
                if out is None:
                    continue
                torch._C._increment_version(tuple(out) if isinstance(out, (list, tuple)) else (out,))
            return result
        return wrapped
    return deco
