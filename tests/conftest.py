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


ALIGN_CASES = ["align_v3_a", "align_v3_b"]


def align_inputs(g):
    """tests/golden/align_v3_*.npz -> (mask_gs, mask_label, mask_conf) in the reference's per-view format (fusion.py:1141-1143)."""
    V, H, W = int(g["V"]), int(g["H"]), int(g["W"])
    gs, labels, confs = [], [], []
    for v in range(V):
        n = int(g["mask_n_%d" % v])
        gs.append(np.unpackbits(g["mask_gs_%d" % v])[:n * H * W].reshape(n, H, W).astype(bool))
        labels.append([str(x) for x in g["mask_label_%d" % v]])
        confs.append(g["mask_conf_%d" % v])
    return gs, labels, confs


def assert_instances_match(g, prefix, instances, V):
    """instances (list of dicts with the reference's keys) against the summary oracle/gen_golden.py:_instances_summary stored"""
    assert [inst["label"] for inst in instances] == [str(x) for x in g[prefix + "_labels"]], prefix
    want_idx = g[prefix + "_idx"]
    for k, inst in enumerate(instances):
        assert [inst["idx"].get(v, -1) for v in range(V)] == want_idx[k].tolist(), (prefix, k)
        assert sorted(set(int(x) for x in inst["vox_idx"])) == g["%s_%d_voxset" % (prefix, k)].tolist(), (prefix, k)
        assert len(inst["vox_idx"]) == int(g["%s_%d_voxlen" % (prefix, k)]), (prefix, k)
        keys = sorted(int(x) for x in inst["conf_per_pt"])
        assert keys == g["%s_%d_confkeys" % (prefix, k)].tolist(), (prefix, k)
        assert [len(inst["conf_per_pt"][x]) for x in keys] == g["%s_%d_confcount" % (prefix, k)].tolist(), (prefix, k)
        flat = [float(c) for x in keys for c in inst["conf_per_pt"][x]]
        assert flat == g["%s_%d_confvals" % (prefix, k)].tolist(), (prefix, k)
