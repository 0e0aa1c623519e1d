"""ctypes binding of libepnet_hip.so (C ABI: include/epnet_ops.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C epnet_amd/csrc``. Loading is
lazy and LOUD: if the shared object is missing or lacks a symbol the header declares, a
RuntimeError is raised -- there is no fallback path.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# EPNET_HIP_LIB lets kernel experiments load an alternative build; the default is the in-tree library
LIB_PATH = os.environ.get("EPNET_HIP_LIB") or os.path.join(_HERE, "lib", "libepnet_hip.so")

_vp = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_sz = ctypes.c_size_t
_i64 = ctypes.c_int64

# name -> (restype, argtypes); mirrors include/epnet_ops.h one to one
SIGNATURES = {
    "epnet_abi_version": (_i, []),
    "epnet_strerror": (ctypes.c_char_p, [_i]),
    "epnet_last_hip_error": (ctypes.c_char_p, []),
    "epnet_furthest_point_sampling": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    "epnet_gather_points": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "epnet_gather_points_grad": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "epnet_ball_query": (_i, [_i, _i, _i, _f, _i, _vp, _vp, _vp, _vp]),
    "epnet_ball_query_workspace_bytes": (_sz, [_i, _i, _i]),
    "epnet_ball_query_ws": (_i, [_i, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "epnet_group_points": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "epnet_group_points_grad": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "epnet_group_concat": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "epnet_group_concat_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "epnet_group_concat_ws": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "epnet_group_concat_grad": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp]),
    "epnet_group_points_grad_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "epnet_group_points_grad_ws": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "epnet_group_concat_grad_ws": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "epnet_three_nn": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "epnet_three_nn_workspace_bytes": (_sz, [_i, _i, _i]),
    "epnet_three_nn_ws": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "epnet_three_interpolate": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "epnet_three_interpolate_grad": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "epnet_three_interpolate_grad_workspace_bytes": (_sz, [_i, _i, _i]),
    "epnet_three_interpolate_grad_ws": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "epnet_group_linear": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "epnet_group_linear_grad_w": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "epnet_feature_gather": (_i, [_i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "epnet_feature_gather_grad": (_i, [_i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "epnet_pool_max": (_i, [ctypes.c_longlong, _i, _vp, _vp, _vp, _vp]),
    "epnet_pool_max_grad": (_i, [ctypes.c_longlong, _i, _vp, _vp, _vp, _vp]),
    "epnet_scene_index_bytes": (_sz, [_i, _i]),
    "epnet_scene_index_build": (_i, [_i, _i, _vp, _vp, _sz, _vp]),
    "epnet_scene_index_build_gathered": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "epnet_furthest_point_sampling_indexed": (_i, [_i, _i, _i, _vp, _vp, _sz, _vp, _vp, _vp]),
    "epnet_sample_centres": (_i, [_i, _i, _i, _vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    "epnet_sample_centres_chain": (_i, [_i, _i, _i, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "epnet_group_concat_multi": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "epnet_ball_query_indexed_multi": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "epnet_ball_query_ordered": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _vp, _vp]),
    "epnet_three_nn_indexed": (_i, [_i, _i, _i, _vp, _vp, _vp, _sz, _vp, _sz, _vp, _vp, _vp]),
    "epnet_ball_query_indexed": (_i, [_i, _i, _i, _f, _i, _vp, _vp, _vp, _sz, _vp, _vp]),
    "epnet_boxes_overlap_bev": (_i, [_i, _vp, _i, _vp, _vp, _vp]),
    "epnet_boxes_iou_bev": (_i, [_i, _vp, _i, _vp, _vp, _vp]),
    "epnet_boxes_iou3d": (_i, [_i, _vp, _i, _vp, _vp, _vp]),
    "epnet_boxes_iou3d_pairs": (_i, [_i, _vp, _vp, _vp, _vp]),
    "epnet_aug_roi_by_noise": (_i, [_i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "epnet_rpn_proposals_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "epnet_rpn_proposals": (_i, [_i, _i, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp, _sz, _vp, _vp, _vp, _vp]),
    "epnet_nms_workspace_bytes": (_sz, [_i]),
    "epnet_nms": (_i, [_vp, _i, _f, _vp, _sz, _vp, _vp, _vp]),
    "epnet_nms_normal": (_i, [_vp, _i, _f, _vp, _sz, _vp, _vp, _vp]),
    "epnet_roipool3d_workspace_bytes": (_sz, [_i, _i, _i]),
    "epnet_roipool3d": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "epnet_pts_in_boxes3d_host": (_i, [_vp, _vp, _vp, _i64, _i64]),
    "epnet_roipool3d_host": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64]),
}

_lib = None


class EpnetError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "epnet_amd: %s not found -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C epnet_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError:
                raise RuntimeError("epnet_amd: %s does not export %s (stale build?)" % (LIB_PATH, name))
            fn.restype = res
            fn.argtypes = args
        if handle.epnet_abi_version() != 1:
            raise RuntimeError("epnet_amd: ABI version mismatch in %s" % LIB_PATH)
        _lib = handle
    return _lib


def check(code, what):
    """turn a non-zero C-ABI return code into a Python exception (the reference exit()s instead)."""
    if code != 0:
        l = lib()
        msg = l.epnet_strerror(code).decode()
        hip = l.epnet_last_hip_error().decode()
        raise EpnetError("%s failed: %s%s" % (what, msg, (" [" + hip + "]") if (code == -2 and hip) else ""))
