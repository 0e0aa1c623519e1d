import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected with -m gpu; when someone runs the whole suite on a CPU-only machine
    # they are skipped rather than failed.
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def hiplib():
    """builds (if needed) and loads libepnet_hip.so"""
    from epnet_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build_hip()
    return _lib.lib()


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


# ---- canaries around device allocations in every GPU test (EPNET_TEST_GUARD=0 switches them off) ----------------------------------
GUARD_BYTES = 4096     # on either side (a multiple of every alignment the library asks for)
CANARY = 0xA5


class GuardedAlloc:
    """what the extension stand-ins and the operator layer allocate themselves -- scratch, scene indices, outputs -- with a canary
    on either side: a kernel that writes past an allocation (or before it) fails the test it runs in even when the stray bytes
    would have landed in somebody else's live memory unnoticed"""

    def __init__(self):
        self.live = []

    def alloc(self, shape, dtype, device, zero=False):
        import torch
        shape = tuple(int(v) for v in (shape if isinstance(shape, (tuple, list, torch.Size)) else (shape,)))
        nbytes = int(np.prod(shape, dtype=np.int64)) * torch.empty((), dtype=dtype).element_size()
        pad = (-nbytes) % 16
        raw = torch.full((GUARD_BYTES + nbytes + pad + GUARD_BYTES,), CANARY, dtype=torch.uint8, device=device)
        self.live.append((raw, nbytes, shape, dtype))
        # a tensor of its own over the same storage -- not a view of `raw` in autograd's sense, so a custom Function may return it
        # and the caller may modify it in place
        body = torch.empty((0,), dtype=dtype, device=device).set_(raw.untyped_storage(), GUARD_BYTES // torch.empty((), dtype=dtype).element_size(), shape)
        if zero:
            body.zero_()
        return body

    def check(self):
        live, self.live = self.live, []
        for raw, nbytes, shape, dtype in live:
            head, tail = raw[:GUARD_BYTES], raw[GUARD_BYTES + nbytes:]
            assert bool((head == CANARY).all()), ("bytes written BEFORE an allocation", shape, dtype)
            assert bool((tail == CANARY).all()), ("bytes written PAST an allocation", shape, dtype, int((tail != CANARY).nonzero()[0]))


def install_guards(monkeypatch):
    """route the device allocations of the extension stand-ins (`torch.empty` inside epnet_amd.{pointnet2,iou3d,roipool3d}_cuda) and
    of the operator layer (pointnet2_utils._new) through a GuardedAlloc; returns it (call .check() after a synchronize)"""
    import torch
    from epnet_amd import iou3d_cuda, pointnet2_cuda, pointnet2_utils, roipool3d_cuda
    g = GuardedAlloc()

    class _TorchProxy:          # `torch.empty(..., device=cuda)` of these modules; everything else passes through
        def __getattr__(self, name):
            return getattr(torch, name)

        @staticmethod
        def empty(*shape, dtype=torch.float32, device=None, **kw):
            if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
                shape = tuple(shape[0])
            if device is not None and torch.device(device).type == "cuda" and not kw:
                return g.alloc(shape, dtype, device)
            return torch.empty(shape, dtype=dtype, device=device, **kw)

    proxy = _TorchProxy()
    for mod in (pointnet2_cuda, iou3d_cuda, roipool3d_cuda):
        if hasattr(mod, "torch"):
            monkeypatch.setattr(mod, "torch", proxy)
    monkeypatch.setattr(pointnet2_utils, "_new", lambda like, shape, dtype=torch.float32, zero=False: g.alloc(shape, dtype, like.device, zero))
    return g


@pytest.fixture(autouse=True)
def _guard_every_gpu_test(request, monkeypatch):
    """canaries around the library-side allocations of every GPU test (tests/test_gpu_sweep.py installs its own, which also cover
    the outputs it hands to the wrappers); EPNET_TEST_GUARD=0 switches them off"""
    if os.environ.get("EPNET_TEST_GUARD", "1") == "0" or "gpu" not in request.keywords or request.module.__name__ == "test_gpu_sweep":
        yield
        return
    import torch
    g = install_guards(monkeypatch)
    yield
    torch.cuda.synchronize()
    g.check()
