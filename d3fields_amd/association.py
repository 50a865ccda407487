"""Multi-view instance association: what the reference does between Grounded-SAM's per-view detections and the `mask`
tensor the field query reads (fusion.py:801-849 merge_instances_from_new_view_vox_ver, :860-868 del_partial_vox_idx,
:978-1040 filter_instances_vox_ver, :1042-1050 reorder_instances, :1052-1063 swap_instance_mask, :1065-1098
align_instance_mask_v3).  `Fusion` exposes these under the reference's names; this module holds the work.

Where the work runs.  Everything per PIXEL or per POINT is a device kernel behind the C-ABI: the 2x2 erosion of a detection,
its masked back-projection / world transform / boundary crop, the 1-cm voxel-grid mean, the 3-cm voxel index of every point
(`Fusion.extract_masked_pcd_in_views`, `pcd_to_index`), the set sizes behind every IoU (`Fusion.vox_idx_iou`: one device hash
set) and the painting of the consensus label images (`d3f_compose_labels`).  What is left on the host is the reference's
bookkeeping over a few dozen detections and a few thousand voxels: which instance a detection joins, which of two overlapping
instances keeps a voxel.  The instances are dicts with the reference's keys, so a caller holding `instances_info` sees no
difference:
    'label'        the detection's text label
    'vox_idx'      voxel indices: the detection's raw per-point array for a new instance, the sorted union after a merge
    'conf_per_pt'  voxel -> list of the confidences of the views that saw it
    'idx'          view -> index of the view's detection that belongs to the instance

Behaviour the outputs depend on, kept as the reference has it: a background detection of a later view that matches nothing
joins the LAST instance (Python's index -1); the union is taken before the "voxels new to a view seen again" difference, so
such a view adds voxels but no confidences; two of the three overlap ratios use raw array lengths; an index can enter the
deletion list more than once and then takes its successor with it; an empty-vs-empty comparison divides by zero.
tests/golden/align_v3_*.npz hold the reference's instances and label images for synthetic detections (CPU restatement:
oracle/np_assoc.py, never imported here).
"""
import ctypes

import numpy as np
import torch

from . import _lib

VOXEL_SIZE = 0.03          # fusion.py:1078
MATCH_IOU = 0.20           # fusion.py:823: a detection joins the best same-label instance above this
OVERLAP_IOU, OVERLAP_PART = 0.25, 0.5      # fusion.py:994
TABLE_LIKE = ("table",)    # fusion.py:1024: labels that only serve as background


def prepare_grid(fusion, boundaries):
    """fusion.py:1067-1091: the association grid of 3-cm voxels over the workspace and its closures, stored under the
    reference's attribute names."""
    from . import pcd_utils
    fusion.iou_threshold = 0.005
    lower = np.array([boundaries["x_lower"], boundaries["y_lower"], boundaries["z_lower"]])
    higher = np.array([boundaries["x_upper"], boundaries["y_upper"], boundaries["z_upper"]])
    fusion.voxel_num = ((higher - lower) / VOXEL_SIZE).astype(np.int32)
    (fusion.pcd_to_voxel, fusion.voxel_to_pcd, fusion.voxel_to_index, fusion.index_to_voxel, fusion.pcd_to_index,
     fusion.index_to_pcd) = pcd_utils.init_low_level_memory(lower, higher, VOXEL_SIZE, voxel_num=fusion.voxel_num)


def detection_voxels(fusion, view, det, boundaries):
    """The voxel index of every point of detection `det` of `view` (device: erosion, back-projection, crop, 1-cm means, index)."""
    cloud = fusion.extract_masked_pcd_in_views([det], [view], boundaries)
    return fusion.pcd_to_index(np.asarray(cloud).reshape(-1, 3))


def merge_view(fusion, instances, view, boundaries):
    obs = fusion.curr_obs_torch
    names = obs["mask_label"][view]
    assert names[0] == "background"
    for det, name in enumerate(names):
        vox = detection_voxels(fusion, view, det, boundaries)
        conf = obs["mask_conf"][view][det]
        best_iou, best = 0, -1
        for k, inst in enumerate(instances):
            if inst["label"] == name:
                iou = fusion.vox_idx_iou(vox, inst["vox_idx"])[0]
                if iou > best_iou:
                    best_iou, best = iou, k
        if not best_iou > MATCH_IOU and (name != "background" or view == 0):
            instances.append({"label": name, "vox_idx": vox, "conf_per_pt": {v: [conf] for v in vox}, "idx": {view: det}})
            continue
        home = instances[best]                                  # best == -1 (nothing matched): the last instance
        seen_before = view in home["idx"]
        home["vox_idx"] = np.unique(np.concatenate([home["vox_idx"], vox]))
        if not seen_before:                                     # (a view seen again: every voxel is in the union already)
            for v in set(vox):
                home["conf_per_pt"].setdefault(v, []).append(conf)
        home["idx"][view] = det
    return instances


def drop_voxels(inst, voxels):
    keep = set(inst["vox_idx"])
    for v in voxels:
        inst["conf_per_pt"].pop(v, None)
        keep.discard(v)
    inst["vox_idx"] = np.array(list(keep))
    return inst


def _outvoted(mine, theirs):
    """Shared voxels that `theirs` saw from more views, or from as many with a higher mean confidence."""
    shared = [v for v in mine if v in theirs]
    return [v for v in shared
            if len(mine[v]) < len(theirs[v]) or (len(mine[v]) == len(theirs[v]) and np.mean(mine[v]) < np.mean(theirs[v]))]


def filter_instances(fusion, instances):
    gone = []                                                   # a list on purpose: see the module docstring
    n = len(instances)
    for a in range(n):
        if a in gone:
            continue
        for b in range(a + 1, n):
            if b in gone:
                continue
            A, B = instances[a], instances[b]
            iou, part_a, part_b = fusion.vox_idx_iou(A["vox_idx"], B["vox_idx"])
            if iou > OVERLAP_IOU or part_a > OVERLAP_PART or part_b > OVERLAP_PART:
                lost_a, lost_b = _outvoted(A["conf_per_pt"], B["conf_per_pt"]), _outvoted(B["conf_per_pt"], A["conf_per_pt"])
                drop_voxels(A, lost_a)
                drop_voxels(B, lost_b)
            gone += [k for k in (a, b) if len(instances[k]["vox_idx"]) < 1]
    gone += [a for a in range(n) if a not in gone and instances[a]["label"] in TABLE_LIKE]
    for a in range(n):
        if a not in gone and len(instances[a]["vox_idx"]) < 1:
            gone.append(a)
    for a in sorted(gone, reverse=True):
        del instances[a]
    return instances


def reorder(instances, query_texts):
    return [inst for text in ["background"] + list(query_texts) for inst in instances if inst["label"] == text]


def _device_tensor(x, dev):
    """The detection stack of one view on the device: tensors (CPU or device) are taken as they are -- np.asarray() of a device
    tensor raises -- and everything else goes through numpy once."""
    if isinstance(x, torch.Tensor):
        return x.to(dev)
    return torch.as_tensor(np.asarray(x)).to(dev)


def paint_label_images(fusion, instances):
    """-> curr_obs_torch['mask'] = (V,H,W) uint8 device tensor: per view, detection idx[view] of instance k painted with k."""
    lib = _lib.load()
    dev = torch.device(fusion.device)
    obs = fusion.curr_obs_torch
    out = torch.zeros((fusion.num_cam, fusion.H, fusion.W), dtype=torch.uint8, device=dev)
    for view in range(fusion.num_cam):
        dets = _device_tensor(obs["mask_gs"][view], dev)        # numpy, CPU or device tensor (a GPU SAM producer's natural output)
        dets = (dets != 0).to(torch.uint8).reshape(dets.shape[0], -1).contiguous()
        assert dets.shape[1] == fusion.H * fusion.W
        owner = np.full(dets.shape[0], -1, np.int32)
        for k, inst in enumerate(instances):                    # ascending: a detection listed twice keeps the later index
            if view in inst["idx"]:
                owner[inst["idx"][view]] = k
        owner_d = torch.from_numpy(owner).to(dev)
        with torch.cuda.device(dev):
            _lib.check(lib.d3f_compose_labels(_lib.ptr(dets), ctypes.c_int32(dets.shape[0]), dets.shape[1], _lib.ptr(owner_d),
                                              _lib.ptr(out[view]), _lib.current_stream_handle(dev)))
    obs["mask"] = out
    return out


def align(fusion, queries, boundaries, expected_labels=None):
    prepare_grid(fusion, boundaries)
    instances = []
    for view in range(fusion.num_cam):
        instances = fusion.merge_instances_from_new_view_vox_ver(instances, view, boundaries)
    instances = fusion.filter_instances_vox_ver(instances)
    instances = fusion.reorder_instances(instances, queries)
    fusion.swap_instance_mask(instances)
    consensus = [inst["label"] for inst in instances]
    fusion.curr_obs_torch["consensus_mask_label"] = consensus
    if expected_labels is not None and consensus != expected_labels:
        print("consensus mask label", consensus)
    return instances
