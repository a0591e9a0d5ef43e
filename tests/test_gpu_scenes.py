"""The real-data entry (labelany3d_amd/fit_scenes.py): a synthetic tree of 64 scene folders (depth_map.npy, cam_params.json,
ground files for some) + a COCO-style annotation file with polygon and run-length segmentations, crowd / tiny / border-touching
annotations -> 3dbbox.json per scene.  Every record is compared with the CPU oracle in REFERENCE-SUBSAMPLE mode (the reference's
semantics for masks above 500 px, drawn from the global NumPy RNG in kept-instance order), the keep decisions and object ids with
the oracle's restatement of the reference's reader (src/util.py:336-383)."""
import json
import os

import numpy as np
import pytest

from oracle import la3d_oracle as O
from oracle import poly_oracle as P

from .test_gpu_parity import assert_records

pytestmark = pytest.mark.gpu


def _oracle_scene(sc, names, subsample, rng_state):
    """the reference's reader + estimate_bbox per kept instance, restated by the oracle; consumes np.random like the reference"""
    H, W = sc["height"], sc["width"]
    depth, K = sc["depth"], np.asarray(sc["K"], dtype=np.float64)
    kept = []
    for a in sc["annotations"]:
        if a["iscrowd"] or "segmentation" not in a:
            continue
        seg = a["segmentation"]
        if isinstance(seg, dict):
            mask = O.rle_decode(np.asarray(seg["counts"]), H, W)
            from_rle = True
        else:
            mask, _ = P.create_boolean_mask_from_polygon((W, H), seg)
            from_rle = False
        if O.keep_instance(O.mask_stats(mask), H, from_rle):
            kept.append((a, mask))
    recs = []
    pts_all = O.depth_to_points(depth[None], K)
    for k, (a, mask) in enumerate(kept):
        pts = pts_all[mask]
        g = sc.get("ground", {}).get(k) if isinstance(sc.get("ground"), dict) else None
        ri = False
        if subsample and len(pts) > 500:
            ri = np.random.randint(0, len(pts), 500)     # the reference's draw (src/util_3dbox.py:124), global RNG
        rec, st, _ = O.fit_points(pts, g, ri)
        if st == 0:
            recs.append((str(k), names.get(a["category_id"], "unknown"), rec))
    return recs


@pytest.mark.parametrize("subsample", [True, False])
def test_fit_scenes_tool_on_a_synthetic_tree(tmp_path, subsample):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from labelany3d_amd import fit_scenes as F

    root = str(tmp_path / "scenes")
    scenes, data = F.synthetic_scenes(64, seed=5, root=root, with_ground=True)
    assert os.path.exists(os.path.join(root, scenes[0]["name"], "depth_map.npy"))
    np.random.seed(123)
    argv = ["--scenes", root, "--batch-images", "24"] + (["--subsample", "--seed", "123"] if subsample else [])
    assert F.main(argv) == 0
    names = F.category_names(data["categories"])
    # the oracle consumes the global RNG in kept-instance order over the scenes IN THE TOOL'S BATCH ORDER (one frame size here:
    # batches are consecutive scenes), like the reference's loop would
    np.random.seed(123)
    n_boxes = 0
    for sc in scenes:
        want = _oracle_scene(sc, names, subsample, None)
        with open(os.path.join(root, sc["name"], "3dbbox.json")) as f:
            got = json.load(f)
        assert [g["obj_id"] for g in got] == [w[0] for w in want], sc["name"]
        assert [g["category_name"] for g in got] == [w[1] for w in want]
        for g, w in zip(got, want):
            assert list(g) == ["obj_id", "category_name", "center_cam", "R_cam", "dimensions", "bbox3D_cam"]   # src/util_3dbox.py:283-290
            rec = np.concatenate([g["center_cam"], g["dimensions"], np.ravel(g["R_cam"]), np.ravel(g["bbox3D_cam"])])
            assert_records(rec[None], w[2][None], f"{sc['name']} obj {g['obj_id']}")
        n_boxes += len(got)
    assert n_boxes > 150


def test_scene_pipeline_in_memory_matches_disk(tmp_path):
    """the same scenes handed over as in-memory dicts (what bench.py --end-to-end times) give the records the tool writes"""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from labelany3d_amd import fit_scenes as F

    root = str(tmp_path / "scenes")
    scenes, data = F.synthetic_scenes(20, seed=9, root=root)
    timings = {}
    mem = {sc["name"]: recs for sc, recs in F.ScenePipeline(batch_images=7, write=False, timings=timings).run(scenes)}
    disk = {sc["name"]: recs for sc, recs in F.ScenePipeline(batch_images=32).run(F.scenes_from_disk(root, os.path.join(root, "annotations.json")))}
    assert set(mem) == set(disk) == {sc["name"] for sc in scenes}
    for k in mem:
        assert mem[k].text == disk[k].text and list(mem[k]) == list(disk[k]), k
        # the file holds exactly the text Python's own writer would produce for the list of dicts (the reference: json.dump(bbox_list, f))
        with open(os.path.join(root, k, "3dbbox.json"), "rb") as f:
            on_disk = f.read()
        assert on_disk == disk[k].text == json.dumps(list(disk[k])).encode(), k
    assert timings["images"] == 20 and timings["h2d_bytes"] > 20 * 480 * 640 * 4 and timings["fit_s"] > 0


def test_scene_pipeline_mixed_frame_sizes_and_rle_strings():
    """images of two frame sizes in one run (batches are formed per size) and COCO run lengths given as COMPRESSED strings (what
    COCONut's annotation files hold): the records of every scene equal those of the scene fitted on its own"""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from labelany3d_amd import fit_scenes as F

    a, _ = F.synthetic_scenes(9, seed=21, H=480, W=640, rle_fraction=0.5)
    b, _ = F.synthetic_scenes(7, seed=22, H=384, W=512, rle_fraction=0.5)
    scenes = [x for pair in zip(a, b) for x in pair] + a[7:]
    for i, sc in enumerate(scenes):
        sc["name"] = f"s{i}"
        for an in sc["annotations"]:
            seg = an["segmentation"]
            if isinstance(seg, dict):
                seg["counts"] = O.rle_to_string(seg["counts"])     # pycocotools' compressed form
    together = {sc["name"]: recs for sc, recs in F.ScenePipeline(batch_images=4, write=False).run(scenes)}
    assert set(together) == {sc["name"] for sc in scenes}
    # staging memory is kept per ring slot and sized by capacity, not per frame size (a COCO run meets hundreds of sizes)
    assert sum(1 for k in F._PINNED if k[0] == "depth") <= 3 and all(len(k) == 2 for k in F._PINNED if k[0] in ("depth", "K"))
    flat = lambda recs: np.array([np.concatenate([r["center_cam"], r["dimensions"], np.ravel(r["R_cam"]), np.ravel(r["bbox3D_cam"])]) for r in recs]).reshape(-1, 39)  # noqa: E731
    for sc in scenes:
        alone = [recs for _, recs in F.ScenePipeline(batch_images=1, write=False).run([sc])][0]
        both = together[sc["name"]]
        assert [(r["obj_id"], r["category_name"]) for r in alone] == [(r["obj_id"], r["category_name"]) for r in both], sc["name"]
        # (small batches of run-length / polygon input take the split engine, whose fp64 partial sums are grouped by the batch's
        # concatenated tile list: the same instance in a different batch agrees to rounding, INTEGRATION.md "Bitwise reproducibility")
        np.testing.assert_allclose(flat(alone)[:, :15], flat(both)[:, :15], rtol=1e-10, atol=1e-10, err_msg=sc["name"])
        np.testing.assert_allclose(flat(alone)[:, 15:], flat(both)[:, 15:], rtol=0, atol=2e-2, err_msg=sc["name"])


def test_category_names_follow_the_reference_table_unless_asked(tmp_path):
    """the reference maps category ids through ITS OWN table whatever the annotation file says
    (replace_categories_with_supercategories, src/util.py:452-462) and writes "unknown" for ids outside it.  A file whose
    `categories` block disagrees with that table must not change 3dbbox.json - unless the caller opts in."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from labelany3d_amd import fit_scenes as F

    root = str(tmp_path / "scenes")
    scenes, data = F.synthetic_scenes(6, seed=4, root=root)
    data["categories"] = [{"id": 1, "name": "human"}, {"id": 3, "name": "automobile"}, {"id": 999, "name": "mystery"}]
    ann = os.path.join(root, "annotations.json")
    with open(ann, "w") as f:
        json.dump(data, f)
    table = F.category_names()
    assert table[1] == "person" and table[3] == "car" and 999 not in table and 12 not in table
    by_ref = {sc["name"]: recs for sc, recs in F.ScenePipeline(batch_images=8, write=False).run(F.scenes_from_disk(root, ann))}
    by_file = {sc["name"]: recs for sc, recs in F.ScenePipeline(batch_images=8, write=False).run(F.scenes_from_disk(root, ann, file_categories=True))}
    seen_ref, seen_file = set(), set()
    for k in by_ref:
        assert len(by_ref[k]) == len(by_file[k])
        for a, b in zip(by_ref[k], by_file[k]):
            assert a["obj_id"] == b["obj_id"] and a["center_cam"] == b["center_cam"]
            seen_ref.add(a["category_name"]); seen_file.add(b["category_name"])
    assert "person" in seen_ref and "unknown" in seen_ref and not ({"human", "automobile", "mystery"} & seen_ref)
    assert {"human", "mystery"} <= seen_file and "person" not in seen_file


def test_scene_pipeline_odd_frame_widths_vs_oracle():
    """COCO frames whose width is not a multiple of 32 (427, 375, 333 ...): the pipeline pads the depth rows on the device and tells
    the fit where the image ends (la3d_fit_args::frame_width); every record must be the oracle's on the UNPADDED frame (masks by the
    reference's decoders), and equal to the scene fitted on its own, two-phase mode (ground vectors) included."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import poly_oracle as PO

    from labelany3d_amd import fit_scenes as F

    a, _ = F.synthetic_scenes(5, seed=31, H=375, W=500, rle_fraction=0.4)
    b, _ = F.synthetic_scenes(5, seed=32, H=640, W=427, rle_fraction=0.4)
    scenes = a + b
    for i, sc in enumerate(scenes):
        sc["name"] = f"o{i}"
    scenes[2]["ground"] = [np.array([0.03, -0.97, 0.1, 1.2])] * 40        # (two-phase: the image's kept objects take a ground vector)
    got = {sc["name"]: recs for sc, recs in F.ScenePipeline(batch_images=4, write=False).run(scenes)}
    n = 0
    for sc in scenes:
        H, W = sc["height"], sc["width"]
        K = np.asarray(sc["K"], dtype=np.float64).reshape(3, 3)
        want = []
        kept = 0
        for an in sc["annotations"]:
            if an.get("iscrowd") or "segmentation" not in an:
                continue
            seg = an["segmentation"]
            if isinstance(seg, dict):
                m = O.rle_decode(seg["counts"] if isinstance(seg["counts"], list) else O.rle_from_string(seg["counts"]), H, W)
                st = O.mask_stats(m, 10)
                keep = O.keep_instance(st, H, True)
            else:
                m = np.logical_or.reduce([PO.create_boolean_mask_from_polygon((W, H), [part])[0] for part in seg])
                st = O.mask_stats(m, 10)
                keep = O.keep_instance(st, H, False)
            if not keep:
                continue
            g = sc["ground"][kept] if "ground" in sc and isinstance(sc["ground"], list) and kept < len(sc["ground"]) else None
            rec, status = O.fit_instance(sc["depth"], m, K, g)[:2]
            if status == 0:
                want.append((str(kept), rec))
            kept += 1
        recs = list(got[sc["name"]])
        assert [r["obj_id"] for r in recs] == [w[0] for w in want], sc["name"]
        for r, (_, rec) in zip(recs, want):
            np.testing.assert_allclose(r["center_cam"], rec[:3], rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(r["dimensions"], rec[3:6], rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(np.ravel(r["R_cam"]), rec[6:15], rtol=0, atol=1e-8)
            np.testing.assert_allclose(np.ravel(r["bbox3D_cam"]), rec[15:], rtol=0, atol=2e-2)
            n += 1
    assert n >= 25
