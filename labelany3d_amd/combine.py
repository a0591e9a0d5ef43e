"""Omni3D record writer: drop-in for the reference's ``combine_coco_results``
(src/tools/combine_results.py:146-310) — per-scene ``3dbbox.json`` + ``cam_params.json`` (+ ``bboxes.json``) -> one Omni3D-format
JSON.  Same arguments, files, printed messages, ids, key order and skip rules; the arithmetic runs on the MI355X in batched form:
the 8 corners of EVERY annotation of the split are projected in one ``la3d_project_boxes`` launch (the reference: eight ``np.dot``
per box, :105-108, :233-239), and each scene's IoU matrix comes from ``la3d_iou_matrix`` (the reference: a Python double loop,
:111-136); the assignment is SciPy's ``linear_sum_assignment`` as in the reference (:138).

The category table (COCO names with Omni3D-style ids, :17-99) is data: ``labelany3d_amd/data/omni3d_coco_categories.json``.
"""
from __future__ import annotations

import json
import os

import numpy as np

from .consumers import iou2d_matrix, project_boxes

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "omni3d_coco_categories.json")


def coco_categories():
    with open(_DATA) as f:
        return json.load(f)


def _load(path):
    with open(path, "r") as f:
        return json.load(f)


def combine_coco_results(results_dir, split, output_path, bbox_filename="3dbbox.json"):
    """See the module docstring; reference src/tools/combine_results.py:146-310."""
    from scipy.optimize import linear_sum_assignment

    categories = coco_categories()
    name_to_id = {c["name"]: c["id"] for c in categories}
    scene_dir = os.path.join(results_dir, split)
    if not os.path.exists(scene_dir):
        raise FileNotFoundError(f"Results directory not found: {scene_dir}")
    scene_ids = sorted(d for d in os.listdir(scene_dir) if os.path.isdir(os.path.join(scene_dir, d)))
    print(f"Found {len(scene_ids)} scenes in {scene_dir}")
    val = split == "val"
    dataset_id = 22 if val else 23
    image_id = 1000000 if val else 2000000
    annotation_id = 100000000 if val else 200000000

    # ---- pass 1 (host): which scenes / annotations survive the reference's skip rules, in its order --------------------------
    images, scenes = [], []          # scenes: (image_id, K, H, W, kept annotations, bbox2d list or None)
    for scene_name in scene_ids:
        scene_path = os.path.join(scene_dir, scene_name)
        bbox_path, cam_path = os.path.join(scene_path, bbox_filename), os.path.join(scene_path, "cam_params.json")
        bbox2d_path = os.path.join(scene_path, "bboxes.json")
        if not os.path.exists(bbox_path):
            print(f"Warning: Missing {bbox_filename} in {scene_name}, skipping")
            continue
        if not os.path.exists(cam_path):
            print(f"Warning: Missing cam_params.json in {scene_name}, skipping")
            continue
        cam = _load(cam_path)
        K = np.array(cam["K"])
        H, W = cam["H"], cam["W"]
        bbox_anno = _load(bbox_path)
        if len(bbox_anno) == 0:
            print(f"Warning: Empty bbox in {scene_name}, skipping")
            continue
        bbox2d = None
        if os.path.exists(bbox2d_path):
            bbox2d = _load(bbox2d_path)
        else:
            print(f"Warning: Missing bboxes.json in {scene_name}, using projected bbox as bbox2D_tight")
        images.append({"width": int(W), "height": int(H), "file_path": f"coco/images/{split}2017/{scene_name}.jpg", "K": K.tolist(),
                       "src_90_rotate": 0, "src_flagged": False, "incomplete": False, "id": image_id, "dataset_id": dataset_id})
        kept = []
        for anno in bbox_anno:
            name = anno.get("category_name", "").replace("_", " ")
            cid = name_to_id.get(name, -1)
            if cid == -1:
                print(f"Warning: Unknown category '{name}' in {scene_name}, skipping")
                continue
            kept.append((anno, name, cid))
        scenes.append((image_id, K, H, W, kept, bbox2d))
        image_id += 1

    # ---- pass 2 (GPU): every corner of every kept annotation projected in one launch ---------------------------------------
    n_total = sum(len(s[4]) for s in scenes)
    proj = np.zeros((0, 4))
    if n_total:
        rec = np.zeros((n_total, 39))
        idx = np.zeros(n_total, np.int32)
        Ks = np.stack([s[1] for s in scenes]).astype(np.float64)
        r = 0
        for si, s in enumerate(scenes):
            for anno, _, _ in s[4]:
                rec[r, 15:39] = np.asarray(anno["bbox3D_cam"], dtype=np.float64).reshape(24)
                idx[r] = si
                r += 1
        proj = project_boxes(rec, Ks, (1.0, 1.0), image_index=idx)[:, :4].cpu().numpy()

    # ---- pass 3: records, then the per-scene matching ---------------------------------------------------------------------------
    annotations = []
    r = 0
    for img_id, K, H, W, kept, bbox2d in scenes:
        local = []
        for anno, name, cid in kept:
            min_x, min_y, max_x, max_y = (float(v) for v in proj[r])
            r += 1
            local.append({
                "behind_camera": False, "truncation": 0.0, "visibility": 1, "segmentation_pts": -1, "lidar_pts": -1, "valid3D": True,
                "category_name": name, "category_id": cid, "image_id": img_id, "id": annotation_id, "dataset_id": dataset_id,
                "center_cam": anno.get("center_cam"), "dimensions": anno.get("dimensions"), "R_cam": anno.get("R_cam"),
                "bbox3D_cam": anno.get("bbox3D_cam"),
                "bbox2D_proj": [min_x, min_y, max_x, max_y],
                "bbox2D_trunc": [max(0, min_x), max(0, min_y), min(W, max_x), min(H, max_y)],   # the reference's expressions (:243-248)
                "depth_error": -1,
            })
            annotation_id += 1
        if bbox2d is not None and len(local) > 0 and len(bbox2d) > 0:
            trunc = np.array([a["bbox2D_trunc"] for a in local], dtype=np.float64)
            iou = iou2d_matrix(trunc, np.array(bbox2d, dtype=np.float64)).cpu().numpy()
            rows, cols = linear_sum_assignment(-iou)
            for i, j in zip(rows, cols):
                local[i]["bbox2D_tight"] = bbox2d[j]
        else:
            for a in local:
                a["bbox2D_tight"] = a["bbox2D_trunc"]
        annotations.extend(local)

    output = {
        "info": {"id": dataset_id, "source": "COCO", "name": f"COCO {'Validation' if val else 'Train'}", "split": split.capitalize(),
                 "version": "0.1", "url": "https://cocodataset.org/#home"},
        "categories": categories,
        "images": images,
        "annotations": annotations,
    }
    os.makedirs(os.path.dirname(output_path) if os.path.dirname(output_path) else ".", exist_ok=True)
    with open(output_path, "w") as f:
        json.dump(output, f)
    print(f"Saved {len(images)} images, {len(annotations)} annotations to {output_path}")
    return output
