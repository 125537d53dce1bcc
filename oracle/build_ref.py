"""TEST INFRASTRUCTURE — build the pieces of the REFERENCE that compile from their own few source files,
from the sources where they lie under /root/reference, into oracle/_ref/ (git-ignored; travels to the
GPU box as a prebuilt .so).  Nothing is copied, nothing is stubbed:

  voxel_layer   mmdet3d/ops/voxel/src/voxelization.cpp + voxelization_cpu.cpp  (CPU build: no WITH_CUDA)
                a torch C++ extension; needs only the torch headers/libs of this image.

Not buildable here (documented in DESIGN.md): bev_pool_ext and the CUDA halves of the voxel ops (nvcc),
spconv (cmake + CUDA), the mmcv-dependent ops.

Usage: python -m oracle.build_ref        (no-op when /root/reference is absent or the .so is current)
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
REF = os.environ.get("SF_REFERENCE_ROOT", "/root/reference")


def voxel_layer_path():
    return os.path.join(OUT, "voxel_layer" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build(verbose=False):
    src_dir = os.path.join(REF, "mmdet3d", "ops", "voxel", "src")
    srcs = [os.path.join(src_dir, f) for f in ("voxelization.cpp", "voxelization_cpu.cpp")]
    out = voxel_layer_path()
    if not all(os.path.exists(s) for s in srcs):
        return out if os.path.exists(out) else None
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in srcs):
        return out
    import torch
    from torch.utils import cpp_extension as ce
    os.makedirs(OUT, exist_ok=True)
    inc = ["-I" + p for p in ce.include_paths()] + ["-I" + sysconfig.get_paths()["include"]]
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    abi = "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)
    cmd = ["g++", "-O2", "-shared", "-fPIC", "-std=c++17", abi, "-DTORCH_EXTENSION_NAME=voxel_layer",
           "-DTORCH_API_INCLUDE_EXTENSION_H"] + inc + srcs + \
          ["-L" + libdir, "-Wl,-rpath," + libdir, "-lc10", "-ltorch", "-ltorch_cpu", "-ltorch_python", "-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("reference voxel_layer build failed:\n" + r.stderr[-4000:])
    return out


def load_voxel_layer():
    """Import oracle/_ref/voxel_layer*.so (None if it was never built)."""
    import importlib.util
    import torch  # noqa: F401  (the extension links against libtorch)
    p = voxel_layer_path()
    if not os.path.exists(p):
        return None
    spec = importlib.util.spec_from_file_location("voxel_layer", p)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(verbose=True))
