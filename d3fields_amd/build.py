"""Builds libd3fields_hip.so (gfx950) in-tree with hipcc.

    python -m d3fields_amd.build [--force] [--save-temps]

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels to the GPU
box with the working tree.
"""
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
INCLUDE = os.path.join(os.path.dirname(PKG_DIR), "include")
LIB_PATH = os.path.join(PKG_DIR, "libd3fields_hip.so")
SOURCES = ["fuse_eval.hip", "fuse_backward.hip", "order_kernels.hip", "grid_kernels.hip", "pcd_kernels.hip", "misc_kernels.hip", "corr_kernels.hip", "track_kernels.hip", "d3f_api.hip"]

# -ffp-contract=off: the arithmetic contract says which products are fused; only
# explicit fmaf() may fuse.  No -ffast-math: IEEE division and accurate expf are part of parity.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
               "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(INCLUDE, "d3fields_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, extra_flags=(), verbose=False):
    if not force and not is_stale():
        return LIB_PATH
    cmd = [_hipcc()] + HIPCC_FLAGS + list(extra_flags) + ["-I", INCLUDE]
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB_PATH


if __name__ == "__main__":
    flags = []
    if "--save-temps" in sys.argv:
        flags.append("-save-temps")
    if "--resource-usage" in sys.argv:
        flags.append("-Rpass-analysis=kernel-resource-usage")
    print(build_library(force=("--force" in sys.argv) or bool(flags), extra_flags=flags, verbose=True))
