"""TEST INFRASTRUCTURE ONLY -- numpy / plain-Python restatement of the reference's multi-view instance association (CPU).

What Fusion.align_instance_mask_v3 (fusion.py:1065-1098) does with the per-view detections 'mask_gs' / 'mask_label' / 'mask_conf':

  view_cloud_index : extract_masked_pcd_in_views([j], [i], boundaries) (fusion.py:1279-1297: 2x2 cv2.erode of the detection,
                     masked back-projection, world frame, boundary crop, 1-cm open3d voxel_down_sample) followed by pcd_to_index
                     on the 3-cm association grid (fusion.py:159-164, 1078-1091)
  merge_view       : merge_instances_from_new_view_vox_ver (fusion.py:801-849)
  drop_voxels      : del_partial_vox_idx (fusion.py:860-868)
  filter_instances : filter_instances_vox_ver (fusion.py:978-1040)
  reorder          : reorder_instances (fusion.py:1042-1050)
  label_images     : swap_instance_mask (fusion.py:1052-1063)
  align            : the whole call

The instances are dicts with the reference's keys ('label', 'vox_idx', 'conf_per_pt', 'idx').  Every quirk that decides an output
is kept and named where it happens (Python's negative index when no instance matched, the deletion list that may hold an index
twice, the union taken BEFORE the "new voxels of a view seen again" difference, raw lengths in two of the three ratios).
Pinned against tests/golden/align_v3_*.npz (written by running the reference, oracle/gen_golden.py:align_case).
Never imported by d3fields_amd.
"""
import numpy as np

from . import np_pcd

VOXEL = 0.03            # fusion.py:1078
NEW_INSTANCE_IOU = 0.20  # fusion.py:823
BACKGROUND_NAMES = ("table",)   # fusion.py:1024


def association_grid(bounds):
    """bounds = [x_lower, x_upper, y_lower, y_upper, z_lower, z_upper] -> (lower, voxel_num int32), fusion.py:1075-1079"""
    lower = np.array([bounds[0], bounds[2], bounds[4]], np.float64)
    higher = np.array([bounds[1], bounds[3], bounds[5]], np.float64)
    return lower, ((higher - lower) / VOXEL).astype(np.int32)


def view_cloud_index(depth, K, pose, detection, bounds, lower, voxel_num):
    """One detection of one view -> the 3-cm voxel index of every point of its 1-cm down-sampled cloud (duplicates stay)."""
    gate = np_pcd.erode_cv2((detection.astype(bool) * 255).astype(np.uint8), np.ones([2, 2], np.uint8)) // 255      # fusion.py:1293
    pose44 = np.concatenate([np.asarray(pose, np.float64)[:3], np.array([[0.0, 0.0, 0.0, 1.0]])], axis=0)
    K = np.asarray(K, np.float64)
    pts, _ = np_pcd.backproject_view(depth, gate.astype(bool), [K[0, 0], K[1, 1], K[0, 2], K[1, 2]], np.linalg.inv(pose44), bounds)
    pts = np_pcd.voxel_mean(pts, 0.01)                                                                               # draw_utils.py:400-402
    return np_pcd.pcd_to_index(pts.reshape(-1, 3), lower, VOXEL, voxel_num)


def set_iou(a, b):
    """fusion.py:794-799; an empty union divides by zero like the reference"""
    return np_pcd.vox_idx_iou(a, b)


def merge_view(instances, view, labels, confs, indices):
    """indices[j] = view_cloud_index of detection j of this view.  Returns the (mutated) list."""
    assert labels[0] == "background"
    for j, name in enumerate(labels):
        mine = indices[j]
        best, best_k = 0, -1
        for k, inst in enumerate(instances):
            if inst["label"] != name:
                continue
            iou = set_iou(mine, inst["vox_idx"])[0]
            if iou > best:
                best, best_k = iou, k
        unseen = not best > NEW_INSTANCE_IOU
        if unseen and (name != "background" or view == 0):
            instances.append({"label": name, "vox_idx": mine, "conf_per_pt": {v: [confs[j]] for v in mine}, "idx": {view: j}})
            continue
        # a background detection of a later view that matched nothing lands here with best_k == -1: Python's LAST instance
        tgt = instances[best_k]
        tgt["vox_idx"] = np.unique(np.concatenate([tgt["vox_idx"], mine]))
        # the union above already holds every voxel of `mine`: a view seen again adds no confidences
        fresh = set(mine).difference(set(tgt["vox_idx"])) if view in tgt["idx"] else set(mine)
        for v in fresh:
            tgt["conf_per_pt"].setdefault(v, []).append(confs[j])
        tgt["idx"][view] = j
    return instances


def drop_voxels(inst, voxels):
    left = set(inst["vox_idx"])
    for v in voxels:
        inst["conf_per_pt"].pop(v, None)
        if v in left:
            left.remove(v)
    inst["vox_idx"] = np.array(list(left))
    return inst


def _weaker(mine, other):
    """voxels of `mine` that `other` holds with more views, or as many views and a higher mean confidence (fusion.py:1000-1008)"""
    out = []
    for v, c in mine.items():
        if v not in other:
            continue
        if len(c) < len(other[v]) or (len(c) == len(other[v]) and np.mean(c) < np.mean(other[v])):
            out.append(v)
    return out


def filter_instances(instances):
    doomed = []                                           # a LIST: an index can enter it more than once (see the end)
    for a, A in enumerate(instances):
        if a in doomed:
            continue
        for b, B in enumerate(instances):
            if b <= a or b in doomed:
                continue
            iou, part_a, part_b = set_iou(A["vox_idx"], B["vox_idx"])
            if iou > 0.25 or part_a > 0.5 or part_b > 0.5:
                from_a, from_b = _weaker(A["conf_per_pt"], B["conf_per_pt"]), _weaker(B["conf_per_pt"], A["conf_per_pt"])
                drop_voxels(A, from_a)
                drop_voxels(B, from_b)
            if len(A["vox_idx"]) < 1:
                doomed.append(a)
            if len(B["vox_idx"]) < 1:
                doomed.append(b)
    for a, A in enumerate(instances):
        if a not in doomed and A["label"] in BACKGROUND_NAMES:
            doomed.append(a)
    for a, A in enumerate(instances):
        if a not in doomed and len(A["vox_idx"]) < 1:
            doomed.append(a)
    for a in sorted(doomed, reverse=True):                # an index listed twice deletes its successor too (fusion.py:1037-1038)
        del instances[a]
    return instances


def reorder(instances, queries):
    return [inst for q in ["background"] + list(queries) for inst in instances if inst["label"] == q]


def label_images(instances, mask_gs):
    out = []
    for view, dets in enumerate(mask_gs):
        img = np.zeros(np.asarray(dets[0]).shape, np.uint8)
        for k, inst in enumerate(instances):
            if view in inst["idx"]:
                img[np.asarray(dets[inst["idx"][view]]).astype(bool)] = k
        out.append(img)
    return np.stack(out, axis=0)


def align(depth, K, pose, mask_gs, mask_label, mask_conf, queries, bounds, stages=None):
    """-> (label image [V,H,W] uint8, consensus labels).  stages: optional dict that receives deep copies of the instances after
    the merges ('merged') and after the filter ('filtered')."""
    import copy
    lower, voxel_num = association_grid(bounds)
    instances = []
    for view in range(len(mask_gs)):
        indices = [view_cloud_index(depth[view], K[view], pose[view], det, bounds, lower, voxel_num) for det in mask_gs[view]]
        instances = merge_view(instances, view, list(mask_label[view]), mask_conf[view], indices)
        if stages is not None:
            stages.setdefault("count_after_view", []).append(len(instances))
    if stages is not None:
        stages["merged"] = copy.deepcopy(instances)
    instances = filter_instances(instances)
    if stages is not None:
        stages["filtered"] = copy.deepcopy(instances)
    instances = reorder(instances, queries)
    return label_images(instances, mask_gs), [inst["label"] for inst in instances]
