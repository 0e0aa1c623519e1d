"""TEST INFRASTRUCTURE -- build the reference's own roipool3d CPU ops into oracle/_ref/.

The only part of the reference's hot path that is plain C++ (no nvcc) is
lib/utils/roipool3d/src/roipool3d.cpp: pt_in_box3d_cpu (:82-95), pts_in_boxes3d_cpu (:97-125)
and roipool3d_cpu (:127-195). This recipe compiles that file UNMODIFIED, from where it lies under
/root/reference, with g++ against the installed torch headers, into oracle/_ref/roipool3d_cuda.so.
No reference source is copied and no stand-in source is written: the two GPU launchers the file
merely declares (roipool3dLauncher, roipool3dLauncher_slow -- defined in the .cu, which needs
nvcc) stay undefined symbols of the shared object and are never called; ``load()`` therefore
opens the module with lazy binding. ``-DAT_CHECK=TORCH_CHECK`` is the one-token spelling change
torch made to that macro after the reference was written.

The pointnet2 / iou3d host files include THC/THC.h and cuda_runtime_api.h and need nvcc-built
kernels: unbuildable here, not attempted.

Runs only where /root/reference exists (the build container); oracle/_ref/ is git-ignored.
"""
import importlib.util
import os
import subprocess
import sys
import sysconfig

REF_SRC = "/root/reference/lib/utils/roipool3d/src/roipool3d.cpp"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_ref")
OUT = os.path.join(OUT_DIR, "roipool3d_cuda.so")


def available():
    return os.path.exists(REF_SRC)


def build(force=False):
    if not available():
        return None
    if os.path.exists(OUT) and not force and os.path.getmtime(OUT) >= os.path.getmtime(REF_SRC):
        return OUT
    import torch
    from torch.utils import cpp_extension

    os.makedirs(OUT_DIR, exist_ok=True)
    inc = []
    for p in cpp_extension.include_paths():
        inc += ["-isystem", p]
    inc += ["-isystem", sysconfig.get_paths()["include"]]
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = [
        "g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-w",
        "-DTORCH_EXTENSION_NAME=roipool3d_cuda", "-DAT_CHECK=TORCH_CHECK",
        "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
        *inc, REF_SRC, "-o", OUT,
        "-L" + torch_lib, "-Wl,-rpath," + torch_lib,
        "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python",
    ]
    subprocess.check_call(cmd)
    return OUT


def load():
    """import oracle/_ref/roipool3d_cuda.so (lazy symbol binding: the CUDA launchers are undefined)."""
    if not os.path.exists(OUT):
        if build() is None:
            return None
    import torch  # noqa: F401  (the module links against libtorch)

    old = sys.getdlopenflags()
    sys.setdlopenflags(os.RTLD_LAZY | os.RTLD_LOCAL)
    try:
        spec = importlib.util.spec_from_file_location("roipool3d_cuda", OUT)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.setdlopenflags(old)
    return mod


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
