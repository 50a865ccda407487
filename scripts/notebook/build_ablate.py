"""Builds the what-if variants of the window / sliced / cell-run kernels for a tuning session: build_ab/<tag>_<bits>.so.
Apply scripts/notebook/patches/r6_whatif_macros.patch first (the product sources carry no what-if blocks since round 6):
-DD3F_WIN_ABLATE=<bits>: 1 no copies after slice 0, 2 no point loop, 4 rows stored over each other (no HBM writes), 8 corner reads without
arithmetic, 16 arithmetic without corner reads.  Results of these libraries are wrong by construction; only times are read.
    D3F_BUILD_EXPERIMENTS=1 python scripts/notebook/build_ablate.py 0 1 4 8 16"""
import os, shutil, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from d3fields_amd import build
os.makedirs("build_ab", exist_ok=True)
macro, tag = ("D3F_SLICED_WHATIF", "sliced") if "--sliced" in sys.argv else (("D3F_RUNS_PREFETCH", "runs") if "--runs" in sys.argv else ("D3F_WIN_ABLATE", "ablate"))
for ab in [int(a) for a in sys.argv[1:] if not a.startswith("--")]:
    build.build_library(force=True, extra_flags=["-D%s=%d" % (macro, ab)])
    shutil.copy(build.LIB_PATH, "build_ab/%s_%d.so" % (tag, ab))
    print("built", tag, ab, flush=True)
