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
SOURCES = ["fuse_launch.hip", "fuse_direct.hip", "fuse_runs.hip", "fuse_sliced.hip", "fuse_window.hip", "fuse_rows.hip", "fuse_backward.hip", "scan_kernels.hip", "order_kernels.hip", "grid_kernels.hip", "pcd_kernels.hip", "assoc_kernels.hip", "misc_kernels.hip", "corr_kernels.hip", "track_kernels.hip", "d3f_api.hip"]

# -ffp-contract=off: the arithmetic contract says which products are fused; only
# explicit fmaf() may fuse.  No -ffast-math: IEEE division and accurate expf are part of parity.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
               "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


FINGERPRINT_PATH = LIB_PATH + ".fingerprint"


# kernels measured and rejected, kept for reproducing the tuning sessions: compiled ONLY with D3F_BUILD_EXPERIMENTS=1
EXPERIMENT_SOURCES = []


def _sources():
    return SOURCES + (EXPERIMENT_SOURCES if _experiments() else [])


def _experiments():
    """D3F_BUILD_EXPERIMENTS=1 (or --experiments) compiles the environment knobs of tuning sessions in (-DD3F_EXPERIMENTS);
    the product build has none (d3f_api.hip: exp_knob)."""
    return os.environ.get("D3F_BUILD_EXPERIMENTS", "0") not in ("", "0")


def source_fingerprint():
    """sha256 over the kernel sources, the public header and the compiler flags: what the .so was built from.
    Content-based (not mtimes), so a working tree copied to another machine keeps a fresh library fresh."""
    import hashlib
    h = hashlib.sha256(" ".join(HIPCC_FLAGS + _sources() + (["-DD3F_EXPERIMENTS"] if _experiments() else [])).encode())
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if os.path.isfile(os.path.join(CSRC, f))]
    files += [os.path.join(CSRC, s) for s in EXPERIMENT_SOURCES if _experiments()]
    for path in sorted(files) + [os.path.join(INCLUDE, "d3fields_hip.h")]:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def is_stale():
    """True when the library is missing or was built from other sources than the ones in the tree."""
    if not os.path.exists(LIB_PATH) or not os.path.exists(FINGERPRINT_PATH):
        return True
    with open(FINGERPRINT_PATH) as fh:
        return fh.read().strip() != source_fingerprint()


def _object_stale(obj, src, headers):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in [src] + headers)


def build_library(force=False, extra_flags=(), verbose=False):
    """One object per source (compiled in parallel, rebuilt only when the source or a header is newer), then one link."""
    if not force and not is_stale():
        return LIB_PATH
    hipcc = _hipcc()
    import fcntl
    lock = open(LIB_PATH + ".lock", "w")
    fcntl.flock(lock, fcntl.LOCK_EX)            # several ranks / test processes may arrive here together
    try:
        if not force and not is_stale():
            return LIB_PATH
        return _build_locked(hipcc, force, extra_flags, verbose)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _build_locked(hipcc, force, extra_flags, verbose):
    objdir = os.path.join(PKG_DIR, "build_exp" if _experiments() else "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(INCLUDE, "d3fields_hip.h")]
    cflags = [f for f in HIPCC_FLAGS if f != "-shared"] + list(extra_flags) + (["-DD3F_EXPERIMENTS"] if _experiments() else []) + ["-I", INCLUDE, "-I", CSRC, "-c"]
    jobs, objs = [], []
    for s in _sources():
        src, obj = os.path.join(CSRC, s), os.path.join(objdir, os.path.basename(s).replace(".hip", ".o"))
        objs.append(obj)
        if force or extra_flags or _object_stale(obj, src, headers):
            cmd = [hipcc] + cflags + [src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            jobs.append((s, subprocess.Popen(cmd, cwd=CSRC)))
    failed = [s for s, p in jobs if p.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, "hipcc -c " + " ".join(failed))
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link, cwd=CSRC)
    with open(FINGERPRINT_PATH, "w") as fh:
        fh.write(source_fingerprint() + "\n")
    return LIB_PATH


if __name__ == "__main__":
    flags = os.environ.get("D3F_EXTRA_CFLAGS", "").split()          # experiments: what-if macros of a tuning session
    if "--experiments" in sys.argv:
        os.environ["D3F_BUILD_EXPERIMENTS"] = "1"
    if "--save-temps" in sys.argv:
        flags.append("-save-temps")
    if "--resource-usage" in sys.argv:
        flags.append("-Rpass-analysis=kernel-resource-usage")
    print(build_library(force=("--force" in sys.argv) or bool(flags), extra_flags=flags, verbose=True))
