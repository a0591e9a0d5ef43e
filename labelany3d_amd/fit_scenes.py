"""Real-data entry of the hot path: a directory of per-image scene folders + a COCO / COCONut annotation file -> ``3dbbox.json``
per scene, batched over images, with uploads overlapped with the fit.

    python -m labelany3d_amd.fit_scenes --scenes DIR --annotations coconut.json [--subsample] [--start_index A --end_index B]

What it reads and writes is what the reference's stages exchange on disk:

* per scene ``<DIR>/<image stem, '/' and '-' replaced by '_'>/`` (reference src/batch_scripts/whole.py:45) with ``depth_map.npy``
  (H, W) float32 and ``cam_params.json`` holding ``K`` (whole.py:63-67); optional ``reconstruction/<k>_<category>_canonical_upright.npy``
  (whole.py:121) = the ground vector of kept instance k;
* instances = the image's annotations through the reference's reader (``read_bounding_boxes_segmentations``, src/util.py:336-383:
  crowd annotations skipped, RLE or polygon segmentation, keep rule height / H > 0.0625, < 10 px in the 10-px border strips,
  area >= 100) - decoded / rasterised and filtered INSIDE the fit launch (``fit_instances_ex(filter=...)``), never as (N,H,W) planes;
* output ``3dbbox.json``: a list of ``{obj_id, category_name, center_cam, R_cam, dimensions, bbox3D_cam}`` (src/util_3dbox.py:283-292),
  obj_id = index among the kept instances of the image, category_name from the annotation file's categories (the reference's own
  table, src/util.py:419-462, where the file has none; unknown ids -> "unknown").

The composition per instance is SURVEY.md section 3.3 (mask -> unproject -> fit); ``--subsample`` is the reference's semantics for
masks above 500 px (500 points drawn with replacement from the global NumPy RNG in kept-instance order, src/util_3dbox.py:123-125).

Pipeline (the reference loops image by image in Python): images are grouped by frame size into batches of ``--batch-images``; a
background thread loads the next batch's ``depth_map.npy`` files into PINNED host memory (a thread per file), packs the
segmentations on the host and starts the host-to-device copies on a copy stream, while the current batch is fitted (one launch per
segmentation kind for the whole batch) and its records come back through pinned memory.  ``timings`` collects the split
(pack / H2D / fit / D2H) that ``bench.py --end-to-end`` reports.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import queue
import threading
import time
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, Iterable, Iterator, List, Optional

import numpy as np
import torch

from ._lib import lib
from .batched import _dev, _upload_many, draw_sample_idx
from .jsonout import SceneRecords, format_scenes
from .masks import fit_instances_ex, mask_stats_poly, mask_stats_rle, pack_polygons, pack_rle, pad_depth_rows, padded_width

_HERE = os.path.dirname(os.path.abspath(__file__))
OUT_NAME = "3dbbox.json"


def scene_dir_name(file_name: str) -> str:
    """The scene folder of an image: its stem with '/' and '-' replaced by '_' (reference src/batch_scripts/whole.py:45)."""
    return file_name.split(".")[0].replace("/", "_").replace("-", "_")


_TABLE: Dict[int, str] = {}


def category_names(categories=None) -> Dict[int, str]:
    """id -> name.  Default (``categories`` None): the reference's built-in COCO / COCONut table - its reader maps ids through that
    table whatever the annotation file says and prints "unknown" for ids outside it (``replace_categories_with_supercategories``,
    src/util.py:419-462; data emitted by tests/golden/make_golden_scenes.py).  Passing the file's ``categories`` block is the explicit
    opt-in to the file's own names (``load_annotations(file_categories=True)``, ``--file-categories``)."""
    if categories:
        return {int(c["id"]): c["name"] for c in categories}
    if not _TABLE:
        with open(os.path.join(_HERE, "data", "coco_category_names.json")) as f:
            _TABLE.update({int(k): v for k, v in json.load(f).items()})
    return _TABLE


def load_annotations(path_or_dict, file_categories: bool = False):
    """COCO-format file / dict -> (images in file order, {image_id: [annotations in file order]}, {category_id: name}).  The names are
    the reference's built-in table unless ``file_categories`` asks for the file's own ``categories`` block."""
    if isinstance(path_or_dict, (str, os.PathLike)):
        with open(path_or_dict) as f:
            data = json.load(f)
    else:
        data = path_or_dict
    by_image: Dict[int, list] = {}
    for a in data.get("annotations", []):
        by_image.setdefault(a["image_id"], []).append(a)
    return data.get("images", []), by_image, category_names(data.get("categories") if file_categories else None)


def _ground_files(scene_dir: str) -> Dict[int, str]:
    """{kept-instance id: path} of the scene's ``reconstruction/<id>_<category>_canonical_upright.npy`` files."""
    rec = os.path.join(scene_dir, "reconstruction")
    out = {}
    if os.path.isdir(rec):
        for fn in os.listdir(rec):
            if fn.endswith("_canonical_upright.npy"):
                head = fn.split("_", 1)[0]
                if head.isdigit():
                    out[int(head)] = os.path.join(rec, fn)
    return out


def scenes_from_disk(scenes_dir: str, annotations, start_index: int = 0, end_index: Optional[int] = None, skip_done: bool = False,
                     out_name: str = OUT_NAME, file_categories: bool = False) -> Iterator[dict]:
    """One dict per image of the annotation file in [start_index, end_index) (the reference's --start_index / --end_index sharding,
    whole.py:25-27,42) whose scene folder holds ``depth_map.npy`` and ``cam_params.json``.  Depth and K are loaded later, by the
    pipeline's loader threads."""
    images, by_image, names = load_annotations(annotations, file_categories)
    for im in images[start_index:end_index]:
        d = os.path.join(scenes_dir, scene_dir_name(im["file_name"]))
        if not (os.path.exists(os.path.join(d, "depth_map.npy")) and os.path.exists(os.path.join(d, "cam_params.json"))):
            continue
        if skip_done and os.path.exists(os.path.join(d, out_name)):   # whole.py:61-62
            continue
        yield {"name": scene_dir_name(im["file_name"]), "dir": d, "width": int(im["width"]), "height": int(im["height"]),
               "annotations": by_image.get(im["id"], []), "names": names}


def _load_scene(scene: dict, depth_out: np.ndarray, k_out: np.ndarray) -> None:
    """depth plane and K of one scene into row slots of the batch's pinned buffers (runs on a loader thread)."""
    if "depth" in scene:
        d, K = scene["depth"], scene["K"]
    else:
        d = np.load(os.path.join(scene["dir"], "depth_map.npy"))
        with open(os.path.join(scene["dir"], "cam_params.json")) as f:
            K = json.load(f)["K"]
    if d.shape != depth_out.shape:
        raise ValueError(f"{scene['name']}: depth_map {d.shape} does not match the image size {depth_out.shape} of the annotation file")
    if isinstance(d, np.ndarray) and d.dtype == np.float32 and d.flags.c_contiguous and depth_out.flags.c_contiguous:
        # one memmove through ctypes: foreign calls release the GIL, np.copyto does not for this case - sixteen loader threads then
        # copy in PARALLEL instead of taking turns (round 5: the staging copies ran at 21 GB/s in aggregate, 1.3 GB/s per thread)
        C.memmove(depth_out.ctypes.data, d.ctypes.data, d.nbytes)
    else:
        np.copyto(depth_out, d, casting="same_kind")
    k_out[:] = np.asarray(K, dtype=np.float64).reshape(9)


# Pinned host buffers outlive a pipeline object (pinning costs far more than the copy it serves; a tool run creates one pipeline,
# bench.py several).  All of them are keyed by PURPOSE and ring slot and sized by CAPACITY - grown to the next power of two when a
# batch needs more, viewed / sliced per batch: a COCO run meets hundreds of frame sizes and a different number of annotations in
# every batch, and a buffer per exact shape would pin (and never release) memory without bound.  So the pinned total is three ring
# slots x the largest batch.  One lock: the producer thread and the caller's thread both come here.
_PINNED: Dict[tuple, torch.Tensor] = {}
_PINNED_LOCK = threading.Lock()


def _pinned_bytes(key, nbytes):
    """at least ``nbytes`` bytes of pinned memory under ``key`` (uint8, grown by doubling; the contents are not preserved)"""
    with _PINNED_LOCK:
        t = _PINNED.get(key)
        if t is None or t.numel() < nbytes:
            cap = 1 << 20
            while cap < nbytes:
                cap *= 2
            t = torch.empty((cap,), dtype=torch.uint8, pin_memory=True)
            _PINNED[key] = t
        return t


def _pinned_rows(key, rows, cols, dtype):
    """a (rows, cols) view of a pinned buffer with at least ``rows`` rows (capacity doubles when it has to grow)"""
    with _PINNED_LOCK:
        t = _PINNED.get(key)
        if t is None or t.shape[0] < rows or t.dtype != dtype or tuple(t.shape[1:]) != tuple(cols):
            cap = 256
            while cap < rows:
                cap *= 2
            t = torch.empty((cap,) + tuple(cols), dtype=dtype, pin_memory=True)
            _PINNED[key] = t
        return t[:rows]


class _Prepared:
    __slots__ = ("scenes", "H", "W", "depth", "K", "groups", "ready", "h2d0", "parity", "t_pack", "t_load", "nbytes", "grounds")


class ScenePipeline:
    """See the module docstring.  ``run(scenes)`` yields ``(scene, records)`` with ``records`` a ``SceneRecords``: the text of the scene's
    ``3dbbox.json`` (``.text``, also written to ``scene['dir']`` when the scene has one and ``write=True``) that behaves like the list
    of its dicts (parsed on first access)."""

    def __init__(self, device=None, batch_images: int = 256, subsample: bool = False, boundary_threshold: int = 10,
                 scale_threshold: int = 100, loader_threads: int = 16, write: bool = True, out_name: str = OUT_NAME, rng=None,
                 timings: Optional[dict] = None):
        self.dev = _dev(device)
        self.batch_images = int(batch_images)
        self.subsample = bool(subsample)
        self.flt = {"boundary_threshold": int(boundary_threshold), "scale_threshold": int(scale_threshold)}
        self.write, self.out_name, self.rng = write, out_name, rng
        self.pool = ThreadPoolExecutor(max_workers=max(1, loader_threads))
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        self.t = timings if timings is not None else {}
        for k in ("load_s", "pack_s", "h2d_bytes", "h2d_s", "fit_s", "d2h_s", "write_s", "images", "instances", "boxes", "batches"):
            self.t.setdefault(k, 0.0)
        self._busy: Dict[int, torch.cuda.Event] = {}   # ring slot -> the event behind its last upload
        self._fitters: Dict[tuple, object] = {}

    # ---- stage 1 (background thread): load, pack, start the uploads ------------------------------------------------------------
    def _pinned_depth(self, P, H, W, parity):
        ev = self._busy.get(parity)
        if ev is not None:
            ev.synchronize()   # the upload that last read this ring slot's buffers (three batches ago) has long finished; make it certain
        flat = _pinned_bytes(("depth", parity), P * H * W * 4)
        return flat[:P * H * W * 4].view(torch.float32).view(P, H, W), _pinned_rows(("K", parity), P, (9,), torch.float64)

    def _prepare(self, scenes: List[dict], parity: int) -> _Prepared:
        H, W = scenes[0]["height"], scenes[0]["width"]
        P = len(scenes)
        pr = _Prepared()
        pr.scenes, pr.H, pr.W = scenes, H, W
        t0 = time.perf_counter()
        dpin, kpin = self._pinned_depth(P, H, W, parity)
        dnp, knp = dpin.numpy(), kpin.numpy()
        # one task per loader thread over a contiguous run of images (a task per image spends its time handing the GIL around)
        nthr = max(1, min(self.pool._max_workers, P))
        cuts = [P * t // nthr for t in range(nthr + 1)]

        def load_range(t):
            for i in range(cuts[t], cuts[t + 1]):
                _load_scene(scenes[i], dnp[i], knp[i])
        inmem = all("depth" in sc and isinstance(sc["depth"], np.ndarray) and sc["depth"].dtype == np.float32 and sc["depth"].flags.c_contiguous
                    and sc["depth"].shape == (H, W) for sc in scenes)
        if inmem:
            # planes already in host memory: ONE foreign call copies them into the pinned batch buffer on native threads - no Python
            # task per loader thread, nothing that takes the interpreter lock while this thread packs
            ptrs = (C.c_void_p * P)(*[sc["depth"].ctypes.data for sc in scenes])
            for i, sc in enumerate(scenes):
                knp[i] = np.asarray(sc["K"], dtype=np.float64).reshape(9)
            loads = [self.pool.submit(lib.la3d_gather_planes_host, ptrs, P, H * W * 4, dnp.ctypes.data, nthr)]
        else:
            loads = [self.pool.submit(load_range, t) for t in range(nthr)]   # np.load / memmove release the GIL: they run WHILE this thread packs
        tp0 = time.perf_counter()
        # the reference's reader: crowd annotations and annotations without a segmentation are skipped (src/util.py:355-358)
        groups = {"rle": {"seg": [], "img": [], "ann": [], "area": [], "cat": []}, "poly": {"seg": [], "img": [], "ann": [], "area": [], "cat": []}}
        for p, sc in enumerate(scenes):
            for j, a in enumerate(sc["annotations"]):
                if a.get("iscrowd") or "segmentation" not in a:
                    continue
                seg = a["segmentation"]
                kind = "rle" if isinstance(seg, dict) and "counts" in seg else "poly"
                g = groups[kind]
                g["seg"].append({"size": seg["size"], "counts": seg["counts"]} if kind == "rle" else seg)
                g["img"].append(p); g["ann"].append(j); g["area"].append(a.get("area")); g["cat"].append(int(a["category_id"]))
        pr.grounds = [(_ground_files(sc["dir"]) if "dir" in sc else {}) if "ground" not in sc else sc["ground"] for sc in scenes]
        packed = {}
        for kind, g in groups.items():
            if not g["seg"]:
                continue
            hint = None if any(v is None for v in g["area"]) else np.clip(np.asarray(g["area"], dtype=np.float64), 0, 2**31 - 1).astype(np.int32)
            if kind == "rle":
                counts, offsets, Hh, Ww = pack_rle(g["seg"])
                if (Hh, Ww) != (H, W):
                    raise ValueError(f"RLE size {(Hh, Ww)} does not match the image size {(H, W)}")
                arrays = [(counts, torch.int32), (offsets, torch.int64)]
            else:
                xy, ro, ir, _, _ = pack_polygons(g["seg"], H, W)
                arrays = [(xy, torch.int32), (ro, torch.int64), (ir, torch.int64)]
            arrays += [(np.asarray(g["img"], np.int32), torch.int32), (hint, torch.int32)]
            packed[kind] = (arrays, g)
        pr.t_pack = time.perf_counter() - tp0
        for f in loads:
            f.result()
        pr.t_load = time.perf_counter() - t0      # (wall time of the loads, the packing above included: they overlap)
        # uploads on the copy stream: the depth planes from pinned memory (asynchronous), the small arrays in one copy per kind
        pr.nbytes = dpin.numel() * 4
        with torch.cuda.stream(self.copy_stream):
            pr.h2d0 = torch.cuda.Event(enable_timing=True)
            pr.h2d0.record(self.copy_stream)
            pr.depth = torch.empty((P, H, W), dtype=torch.float32, device=self.dev)
            pr.depth.copy_(dpin, non_blocking=True)
            if W % 32 != 0:
                # a frame of odd width (COCO: 427, 500, 375, 333 ...): the rows are padded to the next multiple of 32 ON THE DEVICE
                # (the upload stays W wide) and the fit is told where the image ends (frame_width): the tiled / single-pass forms
                # instead of the row-linear one, 3-6 x faster (profiles/r05/r05_frame_sizes.txt)
                pr.depth, _ = pad_depth_rows(pr.depth, self.dev)   # (la3d_pad_rows on the current - the copy - stream)
            pr.K = torch.empty((P, 9), dtype=torch.float64, device=self.dev)
            pr.K.copy_(kpin, non_blocking=True)
            pr.groups = {}
            for kind, (arrays, g) in packed.items():
                up = _upload_many(arrays, self.dev, pinned=_pinned_rows(("small", kind, parity), sum((np.asarray(a).nbytes + 15) & ~15 for a, _ in arrays if a is not None) or 16,
                                                                     (), torch.uint8))
                pr.nbytes += sum(int(np.asarray(a).nbytes) for a, _ in arrays if a is not None)
                pr.groups[kind] = (up, g)
            pr.ready = torch.cuda.Event(enable_timing=True)
            pr.ready.record(self.copy_stream)
        self._busy[parity] = pr.ready
        pr.parity = parity
        return pr

    def _fitter(self, kind, parity, B, H, W):
        """output buffers + workspace of a fit call, kept per (kind, ring slot, frame size) and sized by capacity (the number of
        annotations changes with every batch; a fresh InstanceFitter per call cost ~1 ms of allocations)"""
        from .batched import InstanceFitter
        key = (kind, parity, H, W)
        f = self._fitters.pop(key, None)
        if f is None or f.B < B:
            cap = 256
            while cap < B:
                cap *= 2
            f = InstanceFitter(cap, H, W, self.dev)
        self._fitters[key] = f                      # (most recently used last)
        while len(self._fitters) > 24:              # a run over many frame sizes: the least recently used sizes give their memory back
            self._fitters.pop(next(iter(self._fitters)))
        return f

    # ---- stage 2 (caller's thread): fit, download --------------------------------------------------------------------------------
    def _fit(self, pr: _Prepared):
        H, W, P = pr.H, pr.W, len(pr.scenes)
        cur = torch.cuda.current_stream(self.dev)
        cur.wait_event(pr.ready)
        for t in (pr.depth, pr.K):
            t.record_stream(cur)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        K = pr.K.view(P, 3, 3)
        two_phase = self.subsample or any(len(g) for g in pr.grounds)
        results = {}
        for kind, (up, g) in pr.groups.items():
            for t in up:
                if t is not None:
                    t.record_stream(cur)
            masks_kw = dict(rles=(up[0], up[1], H, W)) if kind == "rle" else dict(polys=(up[0], up[1], up[2], H, W))
            ii, hint = up[-2], up[-1]
            if not two_phase:
                res = fit_instances_ex(pr.depth, K, image_index=ii, filter=self.flt, area_hint=hint, device=self.dev, frame_width=W,
                                       _fitter=self._fitter(kind, pr.parity, len(g["seg"]), H, padded_width(W)), **masks_kw)
                results[kind] = (res["boxes"], res["status"], g)
            else:
                # the keep rule first (its statistics also give N for the subsample draw), then the fit of the whole group with
                # per-instance ground / drawn indices; dropped instances leave through the fused filter as before
                stats = (mask_stats_rle(masks_kw["rles"], self.flt["boundary_threshold"], device=self.dev) if kind == "rle"
                         else mask_stats_poly(masks_kw["polys"], self.flt["boundary_threshold"], device=self.dev))
                results[kind] = (stats, None, g, masks_kw, ii, hint)
        if two_phase:
            results = self._second_phase(pr, results, K)
        out = {}
        for kind, (boxes, status, g) in results.items():
            hb = _pinned_rows(("boxes", kind, pr.parity), boxes.shape[0], boxes.shape[1:], boxes.dtype)
            hs = _pinned_rows(("status", kind, pr.parity), status.shape[0], (), status.dtype)
            hb.copy_(boxes, non_blocking=True); hs.copy_(status, non_blocking=True)
            out[kind] = (hb, hs, g)
        e1.record(cur)
        return out, e0, e1

    def _second_phase(self, pr, results, K):
        H = pr.H
        kept_rows = []   # (image, annotation index, kind, row in its group)
        host_stats = {}
        for kind, (stats, _, g, masks_kw, ii, hint) in results.items():
            st = stats.cpu().numpy()
            host_stats[kind] = st
            height = st[:, 1] if kind == "rle" else st[:, 2]
            keep = (16 * height > H) & (st[:, 3] < 10) & (st[:, 0] >= self.flt["scale_threshold"])   # src/util.py:375; analyze_mask :326
            for r in np.nonzero(keep)[0]:
                kept_rows.append((g["img"][r], g["ann"][r], kind, int(r)))
        kept_rows.sort()   # kept-instance order: image, then annotation order - the order the reference's reader appends in
        ground = {k: np.full((len(v[2]["seg"]), 4), np.nan) for k, v in results.items()}
        counts = np.zeros(len(kept_rows), np.int64)
        per_image_k = {}
        for n, (img, _, kind, r) in enumerate(kept_rows):
            k = per_image_k.get(img, 0)
            per_image_k[img] = k + 1
            gsrc = pr.grounds[img]
            if isinstance(gsrc, dict):
                if k in gsrc:
                    v = gsrc[k]
                    ground[kind][r] = np.asarray(np.load(v) if isinstance(v, str) else v, dtype=np.float64).reshape(-1)[:4]
            elif gsrc is not None and len(gsrc) > k:
                ground[kind][r] = np.asarray(gsrc[k], dtype=np.float64)[:4]
            counts[n] = host_stats[kind][r, 0]
        draws = draw_sample_idx(counts, self.rng) if self.subsample and len(kept_rows) else None
        out = {}
        for kind, (stats, _, g, masks_kw, ii, hint) in results.items():
            B = len(g["seg"])
            si = None
            if draws is not None:
                si = np.zeros((B, draws.shape[1]), np.int32)
                for n, (_, _, kd, r) in enumerate(kept_rows):
                    if kd == kind:
                        si[r] = draws[n]
            gr = ground[kind] if np.isfinite(ground[kind][:, 0]).any() else None
            res = fit_instances_ex(pr.depth, K, image_index=ii, filter=self.flt, area_hint=hint, ground=gr, sample_idx=si, device=self.dev,
                                   frame_width=pr.W, **masks_kw)
            out[kind] = (res["boxes"], res["status"], g)
        return out

    # ---- stage 3: records per scene -----------------------------------------------------------------------------------------------
    def _finish(self, pr: _Prepared, out) -> List[tuple]:
        """Per scene the text of its ``3dbbox.json`` (C writer, jsonout.format_scenes: the bytes json.dump writes for the reference's
        list of six-key dicts, src/util_3dbox.py:283-292) wrapped in a lazy ``SceneRecords``.  Everything per record is array work:
        kept rows in (image, annotation) order, the object id = index among the image's kept instances, category names through the
        unique ids.  No Python object per record (round 4: 0.112 s of a 0.223 s run went into building them)."""
        S = len(pr.scenes)
        recs, img, ann, st, cat = [], [], [], [], []
        for kind, (hb, hs, g) in out.items():
            recs.append(hb.numpy()); st.append(hs.numpy())
            img.append(np.asarray(g["img"], np.int64)); ann.append(np.asarray(g["ann"], np.int64)); cat.append(np.asarray(g["cat"], np.int64))
        if not recs:
            return [(sc, SceneRecords(b"[]", 0)) for sc in pr.scenes]
        R = recs[0] if len(recs) == 1 else np.concatenate(recs)
        img, ann, st, cat = (v[0] if len(v) == 1 else np.concatenate(v) for v in (img, ann, st, cat))
        kept = np.nonzero(st != 6)[0]                                  # 6 = dropped by the instance filter
        kept = kept[np.lexsort((ann[kept], img[kept]))]                # kept-instance order: image, then annotation order
        first = np.searchsorted(img[kept], np.arange(S + 1))           # the image's slice of the kept list
        obj = np.arange(len(kept)) - first[img[kept]]                  # index among the image's kept instances = the reference's object id
        ok = st[kept] == 0       # a failed fit is skipped but keeps its id (the reference prints the exception and goes on, :279-281)
        rows, obj = kept[ok], obj[ok].astype(np.int32)
        scene_off = np.searchsorted(img[rows], np.arange(S + 1)).astype(np.int64)
        # category names: one lookup per distinct (names table, id) - normally ONE table, the annotation file's / the reference's
        tables = [sc.get("names") or category_names() for sc in pr.scenes]
        names: List[str] = []
        name_ids = np.zeros(len(rows), np.int32)
        if all(t is tables[0] for t in tables):
            u, inv = np.unique(cat[rows], return_inverse=True)
            names = [tables[0].get(int(c), "unknown") for c in u]
            name_ids = inv.astype(np.int32)
        else:
            index: Dict[str, int] = {}
            ci, ii = cat[rows], img[rows]
            for n in range(len(rows)):
                nm = tables[ii[n]].get(int(ci[n]), "unknown")
                name_ids[n] = index.setdefault(nm, len(index))
            names = list(index)
        texts = format_scenes(R, rows, obj, name_ids, scene_off, names)
        counts = np.diff(scene_off)
        return [(sc, SceneRecords(texts[i], int(counts[i]))) for i, sc in enumerate(pr.scenes)]

    def _write(self, sc, recs):
        if self.write and "dir" in sc:
            with open(os.path.join(sc["dir"], self.out_name), "wb") as f:
                f.write(recs.text)

    def _batches(self, scenes: Iterable[dict]) -> Iterator[List[dict]]:
        pending: Dict[tuple, list] = {}
        for sc in scenes:
            key = (sc["height"], sc["width"])
            pending.setdefault(key, []).append(sc)
            if len(pending[key]) >= self.batch_images:
                yield pending.pop(key)
        for v in pending.values():
            yield v

    def run(self, scenes: Iterable[dict]) -> Iterator[tuple]:
        q: "queue.Queue" = queue.Queue(maxsize=1)   # one batch prepared ahead of the one being fitted

        def producer():
            try:
                with torch.cuda.device(self.dev):
                    for n, batch in enumerate(self._batches(scenes)):
                        q.put(self._prepare(batch, n % 3))
                q.put(None)
            except BaseException as e:  # noqa: BLE001 - handed to the consumer
                q.put(e)

        th = threading.Thread(target=producer, daemon=True)
        th.start()
        writers = []
        inflight = None
        while True:
            pr = q.get()
            if isinstance(pr, BaseException):
                raise pr
            nxt = None
            if pr is not None:
                t0 = time.perf_counter()
                nxt = (pr,) + self._fit(pr)
                self.t["fit_issue_s"] = self.t.get("fit_issue_s", 0.0) + time.perf_counter() - t0
            if inflight is not None:       # finish the previous batch while this one runs
                ppr, out, e0, e1 = inflight
                e1.synchronize()
                self.t["fit_s"] += e0.elapsed_time(e1) * 1e-3          # fit launches + the record download, on the compute stream
                self.t["h2d_s"] += ppr.h2d0.elapsed_time(ppr.ready) * 1e-3   # the batch's uploads, on the copy stream
                self.t["load_s"] += ppr.t_load; self.t["pack_s"] += ppr.t_pack; self.t["h2d_bytes"] += ppr.nbytes
                self.t["images"] += len(ppr.scenes); self.t["batches"] += 1
                self.t["instances"] += sum(len(g["seg"]) for _, _, g in out.values())
                t0 = time.perf_counter()
                for sc, recs in self._finish(ppr, out):
                    self.t["boxes"] += len(recs)
                    if self.write and "dir" in sc:
                        writers.append(self.pool.submit(self._write, sc, recs))
                    yield sc, recs
                self.t["write_s"] += time.perf_counter() - t0
            inflight = nxt
            if pr is None:
                break
        for w in writers:
            w.result()
        th.join()


# ---------------------------------------------------------------------------------------------------------------------------------
# synthetic scene trees (tests, bench.py --end-to-end): the reference's real tensors are pipeline products that are not in the repo
# ---------------------------------------------------------------------------------------------------------------------------------
def synthetic_scenes(n_scenes: int, seed: int = 0, H: int = 480, W: int = 640, mean_instances: float = 7.0, rle_fraction: float = 0.25,
                     root: Optional[str] = None, with_ground: bool = False):
    """COCO-like scenes: one smooth random depth plane per image, ~Poisson(mean_instances) instances with log-uniform area, each as a
    polygon (ellipse outline, 24 vertices, half-pixel coordinates like the COCONut converter writes) or an uncompressed COCO RLE;
    some crowd / tiny / border-touching annotations for the filter.  ``root``: also write the tree (depth_map.npy, cam_params.json,
    optional reconstruction/*_canonical_upright.npy) and ``annotations.json`` there.  Returns (scenes, annotation dict)."""
    rs = np.random.RandomState(seed)
    vv, uu = np.mgrid[0:H, 0:W]
    images, annos, scenes = [], [], []
    aid = 0
    cats = [1, 3, 17, 18, 44, 62, 63, 67, 999]
    for i in range(n_scenes):
        fn = f"val/{i:06d}-img.jpg"
        images.append({"id": 1000 + i, "file_name": fn, "width": W, "height": H})
        depth = (rs.uniform(2, 6) + rs.uniform(-1e-3, 1e-3) * uu + rs.uniform(0, 3e-3) * vv + 0.02 * rs.randn(H, W)).astype(np.float32)
        f = rs.uniform(450, 650)
        K = [[f, 0.0, W / 2 + rs.uniform(-5, 5)], [0.0, f, H / 2 + rs.uniform(-5, 5)], [0.0, 0.0, 1.0]]
        anns = []
        for _ in range(max(1, rs.poisson(mean_instances))):
            area = np.exp(rs.uniform(np.log(60), np.log(90000)))
            asp = np.exp(rs.uniform(-0.6, 0.6))
            hh, ww = np.sqrt(area * asp), np.sqrt(area / asp)
            cy, cx = rs.uniform(0, H), rs.uniform(0, W)
            a = {"id": aid, "image_id": 1000 + i, "category_id": int(cats[rs.randint(len(cats))]), "iscrowd": int(rs.rand() < 0.05),
                 "bbox": [float(cx - ww / 2), float(cy - hh / 2), float(ww), float(hh)]}
            aid += 1
            if rs.rand() < rle_fraction:
                m = (((vv - cy) / (hh / 2)) ** 2 + ((uu - cx) / (ww / 2)) ** 2) <= 1.0
                flat = m.T.reshape(-1)   # column-major runs, zeros first (COCO)
                change = np.flatnonzero(np.diff(flat.astype(np.int8))) + 1
                runs = np.diff(np.concatenate([[0], change, [flat.size]]))
                counts = runs.tolist() if not flat[0] else [0] + runs.tolist()
                a["segmentation"] = {"size": [H, W], "counts": counts}
                a["area"] = float(m.sum())
            else:
                ang = np.linspace(0, 2 * np.pi, 24, endpoint=False)
                px = np.round((cx + ww / 2 * np.cos(ang)) * 2) / 2
                py = np.round((cy + hh / 2 * np.sin(ang)) * 2) / 2
                a["segmentation"] = [np.stack([px, py], 1).reshape(-1).tolist()]
                if rs.rand() < 0.3:   # a second part (an object seen in two pieces)
                    a["segmentation"].append((np.stack([px, py], 1) + [ww * 0.7, 0.0]).reshape(-1).tolist())
                if rs.rand() < 0.5:
                    a["area"] = float(np.pi * hh * ww / 4)
            anns.append(a)
        annos += anns
        sc = {"name": scene_dir_name(fn), "width": W, "height": H, "annotations": anns, "depth": depth, "K": K, "names": None}
        if with_ground and i % 2 == 0:
            sc["ground"] = {k: np.array([0.05, -0.97, 0.1, 1.2]) + 0.03 * rs.randn(4) for k in range(0, 12, 2)}
        scenes.append(sc)
    data = {"images": images, "annotations": annos,
            "categories": [{"id": 1, "name": "person"}, {"id": 3, "name": "car"}, {"id": 17, "name": "cat"}, {"id": 18, "name": "dog"},
                           {"id": 44, "name": "bottle"}, {"id": 62, "name": "chair"}, {"id": 63, "name": "couch"}, {"id": 67, "name": "dining table"}]}
    names = category_names(data["categories"])
    for sc in scenes:
        sc["names"] = names
    if root is not None:
        os.makedirs(root, exist_ok=True)
        with open(os.path.join(root, "annotations.json"), "w") as f:
            json.dump(data, f)
        for sc in scenes:
            d = os.path.join(root, sc["name"])
            os.makedirs(d, exist_ok=True)
            np.save(os.path.join(d, "depth_map.npy"), sc["depth"])
            with open(os.path.join(d, "cam_params.json"), "w") as f:
                json.dump({"K": sc["K"], "c2w": np.eye(4).tolist()}, f)
            if "ground" in sc:
                os.makedirs(os.path.join(d, "reconstruction"), exist_ok=True)
                cat_of = {}
                for k, v in sc["ground"].items():
                    np.save(os.path.join(d, "reconstruction", f"{k}_{cat_of.get(k, 'object')}_canonical_upright.npy"), v)
    return scenes, data


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--scenes", required=True, help="directory holding one folder per image (depth_map.npy, cam_params.json)")
    ap.add_argument("--annotations", help="COCO / COCONut annotation JSON (default: <scenes>/annotations.json)")
    ap.add_argument("--start_index", type=int, default=0)
    ap.add_argument("--end_index", type=int, default=None)
    ap.add_argument("--gpu_idx", type=int, default=0)
    ap.add_argument("--batch-images", type=int, default=256)
    ap.add_argument("--subsample", action="store_true", help="the reference's 500-point subsample for masks above 500 px (global NumPy RNG)")
    ap.add_argument("--seed", type=int, default=None, help="np.random.seed before the first draw (--subsample)")
    ap.add_argument("--skip-done", action="store_true", help="skip scenes that already hold the output file (whole.py:61-62)")
    ap.add_argument("--file-categories", action="store_true", help="category names from the annotation file's own `categories` block "
                                                                   "(default: the reference's built-in COCO / COCONut table, src/util.py:419-462)")
    ap.add_argument("--make-synthetic", type=int, default=0, metavar="N", help="first write a synthetic tree of N scenes into --scenes")
    args = ap.parse_args(argv)
    if args.make_synthetic:
        synthetic_scenes(args.make_synthetic, seed=0, root=args.scenes)
    ann = args.annotations or os.path.join(args.scenes, "annotations.json")
    if args.seed is not None:
        np.random.seed(args.seed)
    torch.cuda.set_device(args.gpu_idx)
    timings: dict = {}
    pipe = ScenePipeline(device=torch.device("cuda", args.gpu_idx), batch_images=args.batch_images, subsample=args.subsample, timings=timings)
    t0 = time.perf_counter()
    n_scenes = n_boxes = 0
    for sc, recs in pipe.run(scenes_from_disk(args.scenes, ann, args.start_index, args.end_index, args.skip_done,
                                              file_categories=args.file_categories)):
        n_scenes += 1
        n_boxes += len(recs)
    dt = time.perf_counter() - t0
    print(json.dumps({"scenes": n_scenes, "boxes": n_boxes, "seconds": dt, "boxes_per_s": n_boxes / dt if dt else None,
                      "split_s": {k: timings[k] for k in ("load_s", "pack_s", "fit_s", "write_s")}, "h2d_bytes": timings["h2d_bytes"]}))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
