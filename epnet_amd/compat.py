"""Import-path compatibility with the reference tree.

The reference's Python surface does bare top-level imports of its three extensions
(``import pointnet2_cuda as pointnet2`` pointnet2_utils.py:7, ``import iou3d_cuda`` iou3d_utils.py:2,
``import roipool3d_cuda`` roipool3d_utils.py:2) and its model code imports the surface as
``pointnet2_lib.pointnet2.pointnet2_modules`` (lib/net/pointnet2_msg.py:4),
``pointnet2_lib.pointnet2.pytorch_utils`` (lib/net/rpn.py:5), ``lib.utils.iou3d.iou3d_utils``
(lib/rpn/proposal_layer.py:6), ``lib.utils.roipool3d.roipool3d_utils`` (lib/net/rcnn_net.py:4).

``install_extensions()`` makes the three extension names resolve to this package's stand-ins, so an
unmodified reference checkout runs on MI355X (its own Python surface on top of our C ABI).
``install_surface()`` additionally serves this package's surface modules under the reference's
package paths for callers that do not have the reference tree on ``sys.path`` at all.
"""
import importlib
import sys
import types

_EXT = ("pointnet2_cuda", "iou3d_cuda", "roipool3d_cuda")
_SURFACE = {
    "pointnet2_lib.pointnet2.pointnet2_utils": "pointnet2_utils",
    "pointnet2_lib.pointnet2.pointnet2_modules": "pointnet2_modules",
    "pointnet2_lib.pointnet2.pytorch_utils": "pytorch_utils",
    "lib.utils.iou3d.iou3d_utils": "iou3d_utils",
    "lib.utils.roipool3d.roipool3d_utils": "roipool3d_utils",
}


# the "next" rows of SURVEY.md 8(f): the callers either side of the NMS / IoU / pooling ops
_CALLERS = {
    "lib.rpn.proposal_layer": "proposal_layer",                  # lib/net/rpn.py:4 imports ProposalLayer from here
    "lib.rpn.proposal_target_layer": "proposal_target_layer",    # lib/net/rcnn_net.py:5 imports ProposalTargetLayer from here
}


def install_extensions():
    for name in _EXT:
        sys.modules[name] = importlib.import_module("epnet_amd." + name)


def _ensure_package(dotted):
    """create empty namespace packages for the parents of `dotted` unless real ones are importable"""
    parts = dotted.split(".")[:-1]
    for i in range(1, len(parts) + 1):
        pkg = ".".join(parts[:i])
        if pkg in sys.modules:
            continue
        try:
            importlib.import_module(pkg)
        except Exception:
            mod = types.ModuleType(pkg)
            mod.__path__ = []
            sys.modules[pkg] = mod
            if i > 1:
                setattr(sys.modules[".".join(parts[:i - 1])], parts[i - 1], mod)


def install_surface(include_kitti_utils=False):
    install_extensions()
    table = dict(_SURFACE)
    if include_kitti_utils:  # only the 3 helpers exist here; leave the real module alone if present
        table["lib.utils.kitti_utils"] = "kitti_utils"
    for dotted, local in table.items():
        _ensure_package(dotted)
        mod = importlib.import_module("epnet_amd." + local)
        sys.modules[dotted] = mod
        parent, leaf = dotted.rsplit(".", 1)
        setattr(sys.modules[parent], leaf, mod)


def install_callers():
    """serve this package's ProposalLayer / ProposalTargetLayer under the reference's module paths, so that
    lib/net/rpn.py and lib/net/rcnn_net.py construct them unchanged; both read the reference's ``lib.config.cfg`` when
    that module is loaded (same attribute names), their own yaml-valued defaults otherwise"""
    install_extensions()
    for dotted, local in _CALLERS.items():
        _ensure_package(dotted)
        mod = importlib.import_module("epnet_amd." + local)
        sys.modules[dotted] = mod
        parent, leaf = dotted.rsplit(".", 1)
        setattr(sys.modules[parent], leaf, mod)
