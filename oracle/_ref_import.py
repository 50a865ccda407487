"""TEST INFRASTRUCTURE ONLY (never imported by the product path).

Imports the *reference* implementation from /root/reference in THIS container so
that (i) the restatements in oracle/ can be validated against it and (ii) golden
vectors can be generated (oracle/gen_golden.py).  /root/reference does not exist
on the GPU box; nothing under tests/ -m gpu, smoke() or bench.py uses this file.

Recipe follows SURVEY.md §8c: heavy third-party imports of fusion.py:15-30 are
replaced by MagicMock modules; Fusion.__init__ (model downloads, fusion.py:223-286)
is bypassed with Fusion.__new__.
"""
import os
import sys
import types
from unittest.mock import MagicMock

REF_ROOT = os.environ.get("D3F_REFERENCE_ROOT", "/root/reference")

_STUBS = [
    "cv2", "mcubes", "trimesh", "open3d", "torchvision", "torchvision.transforms",
    "groundingdino", "groundingdino.util", "groundingdino.util.inference",
    "groundingdino.util.utils", "groundingdino.datasets",
    "groundingdino.datasets.transforms", "groundingdino.models",
    "groundingdino.util.slconfig", "groundingdino.util.box_ops",
    "segment_anything", "dgl", "dgl.geometry", "pytorch3d", "pytorch3d.transforms",
]


def have_reference():
    return os.path.isfile(os.path.join(REF_ROOT, "fusion.py"))


def import_reference():
    """Returns (fusion_module, corr_utils_module)."""
    if not have_reference():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    for name in _STUBS:
        if name not in sys.modules:
            m = MagicMock(name=name)
            m.__path__ = ["/nonexistent"]
            sys.modules[name] = m
    # pytorch3d (pinned 0.7.5, env.yaml:14) is absent: rigid_tracking (fusion.py:1627-1628) gets the restated
    # so3_exp_map / Transform3d of oracle/pytorch3d_restated.py instead of a MagicMock
    from oracle import pytorch3d_restated as p3d
    tr = types.ModuleType("pytorch3d.transforms")
    tr.__path__ = ["/nonexistent"]
    tr.Transform3d = p3d.Transform3d
    so3 = types.ModuleType("pytorch3d.transforms.so3")
    so3.so3_exp_map = p3d.so3_exp_map
    tr.so3 = so3
    sys.modules["pytorch3d.transforms"] = tr
    sys.modules["pytorch3d.transforms.so3"] = so3
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import importlib
    fusion = importlib.import_module("fusion")
    corr = importlib.import_module("utils.corr_utils")
    # opencv-python (env.yaml:17) is absent: cv2 is a MagicMock, except for the one function whose RESULT the path
    # depends on -- cv2.erode with an all-ones kernel (fusion.py:1293,1305,1561) -- which gets the restated
    # definition of oracle/np_pcd.py so that select_features_rand_v2 of the reference runs here
    from oracle import np_pcd
    fusion.cv2.erode = np_pcd.erode_cv2
    return fusion, corr


def make_reference_fusion(fusion_mod, obs_torch, H, W, mu=0.02, device="cpu"):
    """Build a reference Fusion object without running its __init__."""
    import torch
    f = fusion_mod.Fusion.__new__(fusion_mod.Fusion)
    f.device = device
    f.dtype = torch.float32
    f.mu = mu
    f.H = H
    f.W = W
    f.num_cam = obs_torch["depth"].shape[0]
    f.curr_obs_torch = dict(obs_torch)
    return f
