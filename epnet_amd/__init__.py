"""epnet_amd -- MI355X-native (gfx950) point-cloud geometry operators of EPNet / PointRCNN.

The package holds only what the hot path needs:

* ``csrc/``          hand-written HIP kernels + the C ABI (``include/epnet_ops.h``) -> ``lib/libepnet_hip.so``
* ``_lib``           ctypes loader of that library (fails loudly when it is missing)
* ``pointnet2_cuda``, ``iou3d_cuda``, ``roipool3d_cuda``
                     drop-in stand-ins for the reference's three extension modules (same function
                     names and positional signatures), backed by the C ABI
* ``pointnet2_utils``, ``pointnet2_modules``, ``pytorch_utils``, ``iou3d_utils``,
  ``roipool3d_utils``, ``kitti_utils``
                     the reference's Python operator surface, re-provided with identical names
* ``compat``         registers all of the above under the import paths the reference's callers use
* ``synth``, ``sa_stack``  synthetic KITTI-shaped inputs and the SA/FP op-stack driver used by bench.py

There is no CPU fallback: device ops raise if the HIP library is absent or a tensor is not on a GPU.
"""
__version__ = "0.1.0"
