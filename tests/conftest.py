import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


SCENE_CASES = ["scene_patchres_smooth", "scene_patchres_stress", "scene_fullres_smooth",
               "scene_denseK_1view", "scene_wideC_9views"]
SET_NAMES = ["dino_feats", "mask", "color_tensor"]


def rel_err(a, ref):
    """max |a-ref| / max(|ref|_inf, 1): the tolerance definition of DESIGN.md §Parity."""
    a = np.asarray(a, np.float64)
    ref = np.asarray(ref, np.float64)
    if a.size == 0:
        return 0.0
    scale = max(float(np.max(np.abs(ref))), 1.0)
    return float(np.max(np.abs(a - ref))) / scale
