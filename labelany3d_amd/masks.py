"""Mask ingestion on the device: the data format immediately upstream of the box fit.

The reference turns COCO / COCONut annotations into ``(N,H,W)`` bool arrays on the CPU with pycocotools
(``mask_utils.decode``, reference src/util.py:367,401-402; encoder src/download_coconut.py:167-175) and keeps
an instance when ``height/H > 0.0625 and not truncated and area >= 100`` (src/util.py:291-335, :375).  Here the
run lengths go to the GPU as they are (a few hundred bytes per instance instead of H*W bytes):

    counts, offsets, H, W = pack_rle(annotation_segmentations)        # host: list of COCO RLE dicts
    boxes, status, aux = fit_instances_rle(depth, (counts, offsets, H, W), K, ...)   # no dense mask at all
    masks = rle_decode((counts, offsets, H, W))                         # the reference's mask array, on the GPU
    keep  = keep_instances(mask_stats(masks), H, from_rle=True)         # the reference's three filters

Polygon segmentations — what every kept COCONut instance carries (the converter writes polygons for all non-crowd
instances, src/download_coconut.py:275-280) — take ``create_boolean_mask_from_polygon`` in the reference
(``cv2.fillPoly`` per part, src/util.py:386-400); here the parts go to the GPU as int32 vertex lists:

    polys = pack_polygons([a["segmentation"] for a in annos], H, W)    # host: truncation to int32 like :398
    boxes, status, aux = fit_instances_poly(depth, polys, K, ...)       # rasterised inside the fit kernel
    masks = poly_decode(polys)                                          # the reference's boolean masks, on the GPU
    keep  = keep_instances(mask_stats_poly(polys), H, from_rle=False)   # height = last row - first row + 1 (:328-335)
"""
from __future__ import annotations

import ctypes as C
import itertools
import threading

import numpy as np
import torch

from . import options
from ._lib import AUX, BOX_FILTERED, REC, check, lib
from .batched import InstanceFitter, _as_dev, _bulk, _dev, _ptr, _record, _stream, _upload_many, pad_rows_f32


def rle_from_string(s) -> np.ndarray:
    """COCO compressed RLE string -> run lengths (pycocotools rleFrString), via the library's host helper."""
    if isinstance(s, str):
        s = s.encode("ascii")
    cap = len(s) + 1
    buf = (C.c_int32 * cap)()
    n = lib.la3d_rle_from_string_host(s, len(s), buf, cap)
    if n < 0:
        raise ValueError("malformed COCO RLE string")
    return np.frombuffer(buf, dtype=np.int32, count=n).copy()


def pack_rle(rles):
    """List of COCO RLE objects ({'size': [h, w], 'counts': list | str | bytes}, as in the annotation JSON the
    reference reads at src/util.py:360-368) -> (counts int32 (T,), offsets int64 (B+1,), H, W) NumPy arrays."""
    if isinstance(rles, tuple) and len(rles) == 4:
        return rles
    sizes = {tuple(r["size"]) for r in rles}
    if len(sizes) > 1:
        raise ValueError("all masks of one batch must share the frame size")
    H, W = sizes.pop() if sizes else (0, 0)
    cs = [r["counts"] for r in rles]
    if cs and all(type(c) is list for c in cs):   # uncompressed run lengths as the JSON holds them: one conversion loop for the batch
        lens = np.fromiter(map(len, cs), np.int64, len(cs))
        counts = np.fromiter(itertools.chain.from_iterable(cs), np.int64, int(lens.sum())).astype(np.int32)
        offsets = np.zeros(len(cs) + 1, np.int64)
        np.cumsum(lens, out=offsets[1:])
    else:
        parts = [rle_from_string(c) if isinstance(c, (str, bytes)) else np.asarray(c, dtype=np.int32) for c in cs]
        offsets = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int64)
        counts = np.concatenate(parts).astype(np.int32) if parts and offsets[-1] else np.zeros(1, np.int32)
    if not len(counts):
        counts = np.zeros(1, np.int32)
    return counts, offsets, int(H), int(W)


def rle_decode(rles, device=None, stream=None) -> torch.Tensor:
    """``mask_utils.decode`` for a batch: (B,H,W) bool tensor on the GPU."""
    counts, offsets, H, W = pack_rle(rles)
    dev = _dev(device)
    c, o = _as_dev(counts, torch.int32, dev), _as_dev(offsets, torch.int64, dev)
    B = o.numel() - 1
    out = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(lib.la3d_rle_decode(_ptr(c), _ptr(o), B, H, W, _ptr(out), _stream(stream)), "la3d_rle_decode")
    return out.view(torch.bool)


def mask_stats(masks, boundary_threshold: int = 10, stream=None) -> torch.Tensor:
    """(B,4) int32 on the GPU: area, rows holding a pixel, last-first+1 rows, boundary-strip pixels — what the
    reference's analyze_mask / get_maximum_height / RLE-height look at (src/util.py:291-335, :368-369)."""
    dev = masks.device if isinstance(masks, torch.Tensor) and masks.is_cuda else _dev()
    m = _as_dev(masks, torch.uint8, dev)
    B, H, W = m.shape
    out = torch.empty((B, 4), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(lib.la3d_mask_stats(_ptr(m), B, H, W, int(boundary_threshold), _ptr(out), _stream(stream)), "la3d_mask_stats")
    return out


def mask_stats_rle(rles, boundary_threshold: int = 10, stream=None, device=None) -> torch.Tensor:
    """``mask_stats`` straight from COCO run lengths (a list of RLE objects or the tuple from ``pack_rle``): the four
    filter quantities by interval arithmetic on the runs, no mask plane is decoded.  With ``keep_instances(...,
    from_rle=True)`` this is the reference's RLE branch of read_bounding_boxes_segmentations (src/util.py:364-376)."""
    counts, offsets, H, W = pack_rle(rles)
    dev = _dev(device)
    c, o = _as_dev(counts, torch.int32, dev), _as_dev(offsets, torch.int64, dev)
    B = o.numel() - 1
    out = torch.empty((B, 4), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(lib.la3d_mask_stats_rle(_ptr(c), _ptr(o), B, H, W, int(boundary_threshold), _ptr(out), _stream(stream)),
              "la3d_mask_stats_rle")
    return out


def keep_instances(stats: torch.Tensor, image_height: int, from_rle: bool = True, scale_threshold: int = 100) -> torch.Tensor:
    """The reference's keep rule (src/util.py:375): height/H > 0.0625 and truncation < 10 and area >= 100, with
    height = rows holding a pixel for RLE annotations (:368-369) and last-first+1 for polygons (:328-335)."""
    height = stats[:, 1] if from_rle else stats[:, 2]
    return (height.double() / image_height > 0.0625) & (stats[:, 3] < 10) & (stats[:, 0] >= scale_threshold)


def pack_polygons(segmentations, H=None, W=None):
    """List of polygon segmentations (each a list of parts, each part a flat [x0, y0, x1, y1, ...] list as in the COCO /
    COCONut JSON) -> ``(xy int32 (T,2), ring_offsets int64 (R+1,), inst_rings int64 (B+1,), H, W)``.  Vertices are
    truncated exactly like the reference: ``np.array(polygon).reshape(-1, 2).astype(np.int32)`` (src/util.py:398) — an odd
    number of coordinates raises the same ``ValueError`` from ``reshape``."""
    if isinstance(segmentations, tuple) and len(segmentations) == 5:
        return segmentations
    if H is None or W is None:
        raise ValueError("pack_polygons needs the frame size (H, W)")
    for seg in segmentations:
        if not isinstance(seg, (list, tuple)):
            raise TypeError("polygon segmentation must be a list of parts")
    parts = [polygon for seg in segmentations for polygon in seg]
    inst_rings = np.zeros(len(segmentations) + 1, np.int64)
    np.cumsum([len(seg) for seg in segmentations], out=inst_rings[1:])
    # (a part given as nested pairs [[x, y], ...] is a list too: the flat fast path is for lists of scalars only)
    plain = all(type(q) is list and not (q and isinstance(q[0], (list, tuple, np.ndarray))) for q in parts)
    lens = np.fromiter(map(len, parts), np.int64, len(parts)) if plain else None
    if plain and len(parts) and not (lens & 1).any():
        # flat Python lists (what the annotation JSON holds): ONE conversion loop over all coordinates of the batch instead of one
        # np.array per part (a third of the packing time of a 256-image batch); the truncation is the same astype
        flat = np.fromiter(itertools.chain.from_iterable(parts), np.float64, int(lens.sum()))
        xy = flat.reshape(-1, 2).astype(np.int32)
        ring_off = np.zeros(len(parts) + 1, np.int64)
        np.cumsum(lens >> 1, out=ring_off[1:])
    else:   # arrays, nested pairs, or an odd count (np.reshape raises the reference's ValueError)
        pts, ring = [], [0]
        for polygon in parts:
            q = np.array(polygon).reshape(-1, 2).astype(np.int32)
            pts.append(q)
            ring.append(ring[-1] + len(q))
        xy = np.concatenate(pts).astype(np.int32) if pts else np.zeros((0, 2), np.int32)
        ring_off = np.asarray(ring, np.int64)
    if not len(xy):
        xy = np.zeros((1, 2), np.int32)
    return xy, ring_off, inst_rings, int(H), int(W)


def _poly_dev(polys, dev):
    xy, ro, ir, H, W = polys
    return (_as_dev(xy, torch.int32, dev), _as_dev(ro, torch.int64, dev), _as_dev(ir, torch.int64, dev), H, W)


def poly_decode(polys, device=None, stream=None) -> torch.Tensor:
    """``create_boolean_mask_from_polygon`` (reference src/util.py:386-400) for a batch: (B,H,W) bool tensor on the GPU.
    ``polys`` is the tuple from ``pack_polygons``."""
    dev = _dev(device)
    xy, ro, ir, H, W = _poly_dev(polys, dev)
    B = ir.numel() - 1
    out = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(lib.la3d_poly_decode(_ptr(xy), _ptr(ro), _ptr(ir), B, H, W, _ptr(out), _stream(stream)), "la3d_poly_decode")
    return out.view(torch.bool)


def mask_stats_poly(polys, boundary_threshold: int = 10, stream=None, device=None) -> torch.Tensor:
    """``mask_stats`` for polygon annotations, rasterised in LDS (no plane is written).  With ``keep_instances(...,
    from_rle=False)`` this is the polygon branch of read_bounding_boxes_segmentations (src/util.py:371-376)."""
    dev = _dev(device)
    xy, ro, ir, H, W = _poly_dev(polys, dev)
    B = ir.numel() - 1
    out = torch.empty((B, 4), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(lib.la3d_mask_stats_poly(_ptr(xy), _ptr(ro), _ptr(ir), B, H, W, int(boundary_threshold), _ptr(out), _stream(stream)),
              "la3d_mask_stats_poly")
    return out


def filter_annotations(annotations, image_size, boundary_threshold: int = 10, scale_threshold: int = 100, device=None):
    """The reference's ``read_bounding_boxes_segmentations(annotations, image_size)`` (src/util.py:336-383) without ever
    materialising a mask plane: crowd annotations are skipped (:355-357); RLE segmentations take ``mask_stats_rle`` with
    height = rows holding a pixel (:364-369), polygon segmentations take ``mask_stats_poly`` with height = last row - first
    row + 1 (``create_boolean_mask_from_polygon`` -> ``get_maximum_height``, :371-372, :328-335); the keep rule is :375
    (height/H > 0.0625, fewer than 10 boundary pixels, area >= 100).

    annotations: list of COCO annotation dicts with 'iscrowd', 'bbox', 'category_id' and a 'segmentation' that is an RLE
    ({'size': [h, w], 'counts': list | str | bytes}) or a list of polygon parts; image_size = (width, height) as in the
    reference.  Returns ``(bboxes, segmentations, kept_index, category_ids)``: the kept annotations' boxes, their
    segmentations in annotation order (RLE dicts -> ``fit_instances_rle`` / ``rle_decode``; part lists ->
    ``fit_instances_poly`` / ``poly_decode``; a mixed list -> ``segmentations_to_masks``), their positions in
    ``annotations`` and their raw COCO category ids (the reference maps those to super-category names with a table that is
    not part of this path)."""
    W_img, H_img = int(image_size[0]), int(image_size[1])
    rle_c, rle_i, poly_c, poly_i = [], [], [], []
    for i, a in enumerate(annotations):
        if a.get("iscrowd"):
            continue
        if "segmentation" not in a:
            continue
        seg = a["segmentation"]
        if isinstance(seg, dict) and "counts" in seg:
            rle_c.append({"size": seg["size"], "counts": seg["counts"]})
            rle_i.append(i)
        else:
            poly_c.append(seg)
            poly_i.append(i)
    keep = {}
    if rle_c:
        k = keep_instances(mask_stats_rle(rle_c, boundary_threshold, device=device), H_img, from_rle=True,
                           scale_threshold=scale_threshold).cpu().numpy()
        keep.update({i: (bool(f), c) for i, f, c in zip(rle_i, k, rle_c)})
    if poly_c:
        k = keep_instances(mask_stats_poly(pack_polygons(poly_c, H_img, W_img), boundary_threshold, device=device), H_img,
                           from_rle=False, scale_threshold=scale_threshold).cpu().numpy()
        keep.update({i: (bool(f), c) for i, f, c in zip(poly_i, k, poly_c)})
    kept = [i for i in sorted(keep) if keep[i][0]]
    return ([annotations[i]["bbox"] for i in kept], [keep[i][1] for i in kept], np.asarray(kept, np.int64),
            [annotations[i]["category_id"] for i in kept])


def padded_width(W: int) -> int:
    """The row length the tiled / single-pass forms of the fit want: the next multiple of 32."""
    return (int(W) + 31) // 32 * 32


def pad_depth_rows(depth, device=None):
    """Depth plane(s) (..., H, W) -> ((..., H, padded_width(W)) float32 on the device, W): rows padded on the right with zeros.
    COCO frames come in widths like 427, 500, 375, 333; with run-length / polygon masks such a frame is fitted as a frame of the
    padded width whose first W columns are image (C-ABI ``la3d_fit_args::frame_width``): 4-5 x faster than the row-linear form
    that odd widths otherwise take (profiles/r05/r05_frame_sizes.txt).  ``fit_instances_ex`` / ``fit_instances_rle`` /
    ``fit_instances_poly`` / ``fit_annotations*`` do this themselves; a caller that fits the same planes many times pads once and
    passes ``frame_width=W``."""
    dev = _dev(device)
    d = _as_dev(depth, torch.float32, dev)
    W = int(d.shape[-1])
    Wp = padded_width(W)
    if Wp == W:
        return d, W
    return pad_rows_f32(d, Wp), W


def fit_instances_ex(depth, K, masks=None, rles=None, polys=None, ground=None, sample_idx=None, image_index=None, filter=None,
                     image_size=None, area_hint=None, stream=None, device=None, frame_width=None, _fitter=None, _stats=None):
    """Every option of the fit in one call (C-ABI ``la3d_fit_instances_ex``): exactly one of ``masks`` (B,H,W) u8 / bool,
    ``rles`` (COCO RLE list or ``pack_rle`` tuple), ``polys`` (``pack_polygons`` tuple) gives the masks; ``filter`` as in
    ``fit_instances_poly`` (run-length / polygon masks only); ``image_size=(width, height)`` adds the 2-D boxes the reference's
    Omni3D writer derives from every record - ``bbox2D_proj | bbox2D_trunc`` (B,8), src/tools/combine_results.py:105-108, :238-252 -
    written by the same kernel epilogue that writes the record.  ``area_hint`` (B,) int: mask areas the caller already knows (the
    annotation's ``area``, a preceding filter's statistics) - the size-balanced launch order then skips its estimate pass over the
    masks; a hint only orders the work.  ``frame_width`` (run-length / polygon masks): None - a frame whose width is not a multiple
    of 32 has its depth rows padded here (``pad_depth_rows``); an int - ``depth`` already IS padded (its last dimension is the padded
    width) and the masks' frame is ``frame_width`` columns wide.  Returns a dict: boxes, status, aux, and stats / boxes2d when asked."""
    import ctypes as C

    from ._lib import FitArgs

    dev = _dev(device)
    if (masks is not None) + (rles is not None) + (polys is not None) != 1:
        raise ValueError("give exactly one of masks / rles / polys")
    a = FitArgs()
    keep = []
    if masks is not None:
        m = _as_dev(masks, torch.uint8, dev)
        B, H, W = m.shape
        a.mask = _ptr(m); keep.append(m)
        what = "mask"
    elif rles is not None:
        counts, offsets, H, W = pack_rle(rles)
        counts, offsets, ground, image_index, sample_idx, area_hint = _bulk(
            dev, (counts, torch.int32), (offsets, torch.int64), (ground, torch.float64), (image_index, torch.int32), (sample_idx, torch.int32),
            (area_hint, torch.int32))
        c, o = _as_dev(counts, torch.int32, dev), _as_dev(offsets, torch.int64, dev)
        B = o.numel() - 1
        a.rle_counts, a.rle_offsets = _ptr(c), _ptr(o); keep += [c, o]
        what = "RLE"
    else:
        pxy, pro, pir, H, W = polys
        pxy, pro, pir, ground, image_index, sample_idx, area_hint = _bulk(
            dev, (pxy, torch.int32), (pro, torch.int64), (pir, torch.int64), (ground, torch.float64), (image_index, torch.int32),
            (sample_idx, torch.int32), (area_hint, torch.int32))
        xy, ro, ir, H, W = _poly_dev((pxy, pro, pir, H, W), dev)
        B = ir.numel() - 1
        a.poly_xy, a.ring_offsets, a.inst_rings = _ptr(xy), _ptr(ro), _ptr(ir); keep += [xy, ro, ir]
        what = "polygon"
    if filter and masks is not None:
        raise ValueError("the fused filter needs run-length or polygon masks")
    fw = 0
    if masks is None:
        if frame_width is None:
            if W % 32 != 0:
                with torch.cuda.device(dev):
                    depth, fw = pad_depth_rows(depth, dev)
                W = padded_width(W)
        else:
            if int(frame_width) != W:
                raise ValueError(f"frame_width {frame_width} does not match the {what} frame width {W}")
            Wd = int(depth.shape[-1])
            if Wd != W:
                if Wd < W or Wd % 32 != 0:
                    raise ValueError("padded depth rows must be a multiple of 32 wide and at least frame_width")
                fw, W = W, Wd
    d, k, P, ii, g, si = _fit_common(depth, K, H, W, B, ground, sample_idx, image_index, dev, what)
    out = {}
    with torch.cuda.device(dev):
        # (_fitter / _stats: buffers a per-image caller keeps between calls - fit_annotations - instead of allocating them per call)
        f = _fitter if _fitter is not None else InstanceFitter(B, H, W, dev)
        if f.B < B or (f.H, f.W) != (H, W):
            raise ValueError("_fitter too small for this call")
        out.update(boxes=f.boxes[0][:B], status=f.status[0][:B], aux=f.aux[0][:B])   # (a kept fitter may have more rows than this call)
        if filter:
            out["stats"] = _stats if _stats is not None else torch.zeros((B, 4), dtype=torch.int32, device=dev)
        if image_size is not None:
            out["boxes2d"] = torch.full((B, 8), float("nan"), dtype=torch.float64, device=dev)
        if B == 0:
            return out
        a.struct_size = C.sizeof(FitArgs)
        a.B, a.H, a.W = B, H, W
        a.depth, a.depth_plane_stride, a.image_index = _ptr(d), (H * W if P > 1 else 0), _ptr(ii)
        a.K, a.k_stride = _ptr(k), (9 if k.shape[0] > 1 else 0)
        a.ground, a.sample_idx = _ptr(g), _ptr(si)
        if filter:
            a.filter_boundary, a.filter_min_area, a.filter_max_edge = _filter_args(filter)
            a.stats = _ptr(out["stats"])
        if image_size is not None:
            a.proj, a.image_width, a.image_height = _ptr(out["boxes2d"]), float(image_size[0]), float(image_size[1])
        if area_hint is not None:
            ah = _as_dev(area_hint, torch.int32, dev).reshape(-1)
            if ah.numel() != B:
                raise ValueError("area_hint must have one entry per instance")
            a.area_hint = _ptr(ah); keep.append(ah)
        a.out, a.status, a.aux = _ptr(f.boxes[0]), _ptr(f.status[0]), _ptr(f.aux[0])
        a.workspace, a.stream = _ptr(f.workspace[0]), _stream(stream)
        a.opt_engine, a.opt_launch_order, a.opt_build = options.codes()
        a.frame_width = fw
        check(lib.la3d_fit_instances_ex(C.byref(a)), "la3d_fit_instances_ex")
    _record(stream, d, k, ii, g, si, *keep, f.workspace, *out.values())
    return out


_ANN_CACHE: dict = {}   # (B, H, W, device) -> (InstanceFitter, stats, pinned read-back buffer): the per-image pattern repeats a few shapes


_ANN_LOCK = threading.Lock()


def _ann_buffers(B, H, W, dev, kind):
    # per THREAD: two threads calling fit_annotations with equal shapes must not share a fitter, its statistics
    # buffer and its pinned read-back buffer; the dict itself is guarded by a lock
    key = (B, H, W, kind, dev.index if dev.index is not None else torch.cuda.current_device(), threading.get_ident())
    with _ANN_LOCK:
        ent = _ANN_CACHE.get(key)
        if ent is None and len(_ANN_CACHE) >= 64:
            _ANN_CACHE.pop(next(iter(_ANN_CACHE)))
    if ent is None:
        f = InstanceFitter(B, H, W, dev)
        # the pinned buffer mirrors the head of the fitter's arena: boxes | aux | status (256-aligned pieces), without the workspace
        ent = (f, torch.empty((B, 4), dtype=torch.int32, device=dev), torch.empty(f._arena.numel() - f.workspace.numel(), dtype=torch.uint8, pin_memory=True))
        with _ANN_LOCK:
            _ANN_CACHE[key] = ent
    return ent


def fit_annotations(annotations, image_size, depth, K, ground=None, boundary_threshold: int = 10, scale_threshold: int = 100,
                    image_index=None, device=None, to_host: bool = False):
    """``read_bounding_boxes_segmentations`` (reference src/util.py:336-383) and the box fit in ONE pass over the annotations:
    crowd annotations are skipped (:355-357); every other annotation's segmentation is decoded / rasterised once, inside the
    fit launch, which also evaluates the keep rule (:375) on the bit image and fits only the kept instances (``filter=`` of
    ``fit_instances_rle`` / ``fit_instances_poly``).  One launch per segmentation kind present.

    depth / K / image_index as in ``fit_instances`` (image_index per ANNOTATION when depth holds several planes); ground: None
    or (len(annotations), 4).  Returns ``(bboxes, kept_index, category_ids, boxes (n,39) f64, status (n,) i32)`` for the kept
    annotations in annotation order, the last two on the GPU - or, with ``to_host=True``, as NumPy arrays taken from the ONE
    packed read-back (records | aux | status in a single copy into pinned memory) that the keep decision needs anyway: no
    further device work, which is what a caller writing JSON wants.  Output buffers, workspace and the pinned buffer are kept per
    (B, H, W) between calls (the per-image pattern repeats a handful of shapes)."""
    W_img, H_img = int(image_size[0]), int(image_size[1])
    dev = _dev(device)
    groups = split_annotations(annotations)
    flt = {"boundary_threshold": boundary_threshold, "scale_threshold": scale_threshold}
    if (to_host and isinstance(depth, torch.Tensor) and depth.is_cuda and depth.dtype == torch.float32 and depth.is_contiguous()
            and not isinstance(K, torch.Tensor) and not isinstance(ground, torch.Tensor) and not isinstance(image_index, torch.Tensor)):
        return _fit_annotations_host(annotations, groups, W_img, H_img, depth, K, ground, image_index, flt)
    Wp = padded_width(W_img)
    if Wp != W_img and any(idx for idx, _ in groups.values()):   # a frame of odd width: rows padded once for both segmentation kinds
        with torch.cuda.device(dev):
            depth, _ = pad_depth_rows(depth, dev)
    sels, box_all, st_all, pins = [], [], [], []
    for kind, (idx, segs) in groups.items():
        if not idx:
            continue
        sel = np.asarray(idx, np.int64)
        g = None if ground is None else (ground[torch.as_tensor(sel, device=ground.device)] if isinstance(ground, torch.Tensor)
                                         else np.asarray(ground, dtype=np.float64)[sel])
        ii = None if image_index is None else (image_index[torch.as_tensor(sel, device=image_index.device)]
                                               if isinstance(image_index, torch.Tensor) else np.asarray(image_index)[sel])
        # the annotation's own "area" (COCO: the mask area in pixels), when every annotation of the group has one, spares the launch
        # order its estimate pass
        ar = [annotations[i].get("area") for i in idx]
        hint = None if any(v is None for v in ar) else np.clip(np.asarray(ar, dtype=np.float64), 0, 2**31 - 1).astype(np.int32)
        # every small host array of the group goes up in ONE copy (six to eight separate uploads cost ~100 us per image)
        host = lambda v: None if isinstance(v, torch.Tensor) else v     # noqa: E731   (device tensors pass through as they are)
        if kind == "rle":
            counts, offsets, Hh, Ww = pack_rle(segs)
            up = _upload_many([(counts, torch.int32), (offsets, torch.int64), (host(g), torch.float64), (host(ii), torch.int32),
                               (hint, torch.int32), (host(K), torch.float64)], dev)
            kw = dict(rles=(up[0], up[1], Hh, Ww))
        else:
            xy, ro, ir, Hh, Ww = pack_polygons(segs, H_img, W_img)
            up = _upload_many([(xy, torch.int32), (ro, torch.int64), (ir, torch.int64), (host(g), torch.float64), (host(ii), torch.int32),
                               (hint, torch.int32), (host(K), torch.float64)], dev)
            kw = dict(polys=(up[0], up[1], up[2], Hh, Ww))
        g_d = up[-4] if up[-4] is not None else g
        ii_d = up[-3] if up[-3] is not None else ii
        K_d = up[-1] if up[-1] is not None else K
        fitter, stats_buf, pin = _ann_buffers(len(idx), Hh, padded_width(Ww), dev, kind)
        res = fit_instances_ex(depth, K_d, ground=g_d, image_index=ii_d, device=dev, filter=flt, area_hint=up[-2], _fitter=fitter,
                               _stats=stats_buf, frame_width=Ww, **kw)
        head = fitter._arena[:pin.numel()]
        pin.copy_(head, non_blocking=True)      # records | aux | status of this group: one copy, read after the one synchronisation
        sels.append(sel); box_all.append(res["boxes"]); st_all.append(res["status"]); pins.append((pin, len(idx)))
    if not sels:
        return [], np.zeros(0, np.int64), [], torch.zeros((0, 39), dtype=torch.float64, device=dev), torch.zeros(0, dtype=torch.int32, device=dev)
    torch.cuda.current_stream(dev).synchronize()          # the one synchronisation of the call
    up8 = lambda v: (v + 255) // 256 * 256  # noqa: E731  (InstanceFitter's arena layout)
    hb, hs = [], []
    for pin, n in pins:
        raw = pin.numpy()
        nb, na = n * REC * 8, n * AUX * 8
        hb.append(raw[:nb].view(np.float64).reshape(n, REC))
        hs.append(raw[up8(nb) + up8(na):up8(nb) + up8(na) + n * 4].view(np.int32))
    st_host = hs[0] if len(hs) == 1 else np.concatenate(hs)
    sel_c = np.concatenate(sels)
    pos = np.nonzero(st_host != 6)[0]                     # kept rows of the concatenated results ...
    pos = pos[np.argsort(sel_c[pos], kind="stable")]      # ... in annotation order
    kept = sel_c[pos]
    if to_host:
        bh = hb[0] if len(hb) == 1 else np.concatenate(hb)
        return ([annotations[i]["bbox"] for i in kept], kept, [annotations[i]["category_id"] for i in kept], bh[pos].copy(), st_host[pos].copy())
    boxes_c = box_all[0] if len(box_all) == 1 else torch.cat(box_all)
    status_c = st_all[0] if len(st_all) == 1 else torch.cat(st_all)
    pt = torch.as_tensor(pos, device=dev)
    return ([annotations[i]["bbox"] for i in kept], kept, [annotations[i]["category_id"] for i in kept],
            boxes_c.index_select(0, pt), status_c.index_select(0, pt))


def split_annotations(annotations):
    """Annotations by segmentation kind, as ``read_bounding_boxes_segmentations`` treats them (reference src/util.py:355-367):
    crowd annotations and annotations without a segmentation are skipped; a dict with ``counts`` is a run-length mask, anything
    else a list of polygon parts.  Returns {"rle": (indices, segmentations), "poly": (indices, segmentations)}."""
    groups = {"rle": ([], []), "poly": ([], [])}
    for i, a in enumerate(annotations):
        if a.get("iscrowd") or "segmentation" not in a:
            continue
        seg = a["segmentation"]
        kind = "rle" if isinstance(seg, dict) and "counts" in seg else "poly"
        groups[kind][0].append(i)
        groups[kind][1].append({"size": seg["size"], "counts": seg["counts"]} if kind == "rle" else seg)
    return groups


def annotation_areas(annotations, default: float = 0.0) -> np.ndarray:
    """The ``area`` field of every annotation (COCO: the mask area in pixels) - the per-instance cost ``plan_shards`` balances by,
    known from the annotation file alone."""
    return np.asarray([float(a.get("area", default) or default) for a in annotations], dtype=np.float64)


def fit_annotations_all(annotations, image_size, depth, K, ground=None, image_index=None, filter=None, device=None):
    """The fit of ``fit_annotations`` with one record per ANNOTATION, in annotation order, on the GPU: ``(boxes (n,39) f64,
    status (n,) i32)``.  Skipped annotations (crowd, no segmentation) and - with ``filter`` (a dict of ``boundary_threshold`` /
    ``scale_threshold``, or True for the reference's 10 / 100) - the ones the keep rule drops carry status 6 and a NaN record.
    One launch per segmentation kind present; the segmentations are decoded / rasterised inside the fit kernel (no u8 plane
    exists anywhere).  This is the rank-local step of ``shard.fit_annotations_sharded``."""
    W_img, H_img = int(image_size[0]), int(image_size[1])
    dev = _dev(device)
    n = len(annotations)
    boxes = torch.full((n, REC), float("nan"), dtype=torch.float64, device=dev)
    status = torch.full((n,), int(BOX_FILTERED), dtype=torch.int32, device=dev)
    flt = None
    if filter:
        flt = {"boundary_threshold": 10, "scale_threshold": 100} if filter is True else dict(filter)
    groups = split_annotations(annotations)
    if W_img % 32 != 0 and any(idx for idx, _ in groups.values()):   # a frame of odd width: rows padded once for both kinds
        with torch.cuda.device(dev):
            depth, _ = pad_depth_rows(depth, dev)
    for kind, (idx, segs) in groups.items():
        if not idx:
            continue
        sel = np.asarray(idx, np.int64)
        sel_t = torch.as_tensor(sel, device=dev)
        take = lambda v: None if v is None else (v.to(dev)[sel_t] if isinstance(v, torch.Tensor) else np.asarray(v)[sel])  # noqa: E731
        ar = [annotations[i].get("area") for i in idx]
        hint = None if any(v is None for v in ar) else np.clip(np.asarray(ar, dtype=np.float64), 0, 2**31 - 1).astype(np.int32)
        kw = dict(rles=segs) if kind == "rle" else dict(polys=pack_polygons(segs, H_img, W_img))
        res = fit_instances_ex(depth, K, ground=take(ground), image_index=take(image_index), device=dev, filter=flt, area_hint=hint,
                               frame_width=W_img, **kw)
        boxes.index_copy_(0, sel_t, res["boxes"])
        status.index_copy_(0, sel_t, res["status"])
    return boxes, status


def _raw_stream(dev_index: int) -> int:
    """torch's current stream of a device as a hipStream_t value (the private accessor is ~10 x cheaper than building a Stream object:
    this sits on a ~100 us per-image path)."""
    try:
        return torch._C._cuda_getCurrentRawStream(dev_index)
    except AttributeError:
        return torch.cuda.current_stream(dev_index).cuda_stream


def _fit_annotations_host(annotations, groups, W_img, H_img, depth, K, ground, image_index, flt):
    """``fit_annotations(to_host=True)`` with the depth plane(s) resident and everything else on the host: ONE foreign call per
    segmentation kind (``la3d_fit_annotations_host``: the small arrays go up through the library's pinned block, the records come back
    through it, the call polls a completion flag) - no torch tensor, no wrapper layers in between."""
    from ._lib import FitArgs

    P = depth.shape[0] if depth.dim() == 3 else 1
    dev_index = depth.device.index if depth.device.index is not None else torch.cuda.current_device()
    Wp = padded_width(W_img)
    if Wp != W_img:   # a frame of odd width: rows padded to the next multiple of 32, frame_width says where the image ends
        depth, _ = pad_depth_rows(depth, depth.device)
    Kh = np.ascontiguousarray(np.asarray(K, dtype=np.float64).reshape(-1, 9))
    if Kh.shape[0] not in (1, P):
        raise ValueError("K must be (3,3) or (P,3,3)")
    sels, recs, sts = [], [], []
    for kind, (idx, segs) in groups.items():
        if not idx:
            continue
        sel = np.asarray(idx, np.int64)
        B = len(idx)
        a = FitArgs()
        a.struct_size = C.sizeof(FitArgs)
        keep = []
        if kind == "rle":
            counts, offsets, Hh, Ww = pack_rle(segs)
            a.rle_counts, a.rle_offsets = counts.ctypes.data, offsets.ctypes.data
            keep += [counts, offsets]
        else:
            xy, ro, ir, Hh, Ww = pack_polygons(segs, H_img, W_img)
            a.poly_xy, a.ring_offsets, a.inst_rings = xy.ctypes.data, ro.ctypes.data, ir.ctypes.data
            keep += [xy, ro, ir]
        if (Hh, Ww) != (H_img, W_img) or depth.shape[-2:] != (Hh, Wp):
            raise ValueError(f"depth {tuple(depth.shape[-2:])} / image size {(H_img, W_img)} do not match the mask size {(Hh, Ww)}")
        a.B, a.H, a.W = B, Hh, Wp
        a.frame_width = Ww if Wp != Ww else 0
        a.depth, a.depth_plane_stride = depth.data_ptr(), (Hh * Wp if P > 1 else 0)
        if image_index is not None:
            ii = np.ascontiguousarray(np.asarray(image_index)[sel], np.int32)
            if ii.size and (ii.min() < 0 or ii.max() >= P):
                raise ValueError("image_index out of range")
            a.image_index = ii.ctypes.data; keep.append(ii)
        elif P > 1:
            # one plane per ANNOTATION (crowd / unsegmented ones included): this kind's instances pick theirs by annotation index
            if P != len(annotations):
                raise ValueError("several depth planes need image_index (or one plane per annotation)")
            ii = sel.astype(np.int32)
            a.image_index = ii.ctypes.data; keep.append(ii)
        a.K, a.k_stride = Kh.ctypes.data, (9 if Kh.shape[0] > 1 else 0)
        if ground is not None:
            g = np.ascontiguousarray(np.asarray(ground, dtype=np.float64).reshape(-1, 4)[sel])
            a.ground = g.ctypes.data; keep.append(g)
        ar = [annotations[i].get("area") for i in idx]
        if not any(v is None for v in ar):
            hint = np.clip(np.asarray(ar, dtype=np.float64), 0, 2**31 - 1).astype(np.int32)
            a.area_hint = hint.ctypes.data; keep.append(hint)
        a.filter_boundary, a.filter_min_area, a.filter_max_edge = _filter_args(flt)
        out = np.empty((B, REC), np.float64)
        st = np.empty(B, np.int32)
        a.out, a.status = out.ctypes.data, st.ctypes.data
        # the C entry enqueues on the stream it is handed, on the thread's CURRENT device: the depth's device, and the stream the
        # depth (and the padding above) was produced on - torch's current stream of that device
        a.stream = _raw_stream(dev_index)
        if dev_index == torch.cuda.current_device():
            check(lib.la3d_fit_annotations_host(C.byref(a)), "la3d_fit_annotations_host")
        else:
            with torch.cuda.device(dev_index):
                check(lib.la3d_fit_annotations_host(C.byref(a)), "la3d_fit_annotations_host")
        sels.append(sel); recs.append(out); sts.append(st)
    if not sels:
        return [], np.zeros(0, np.int64), [], np.zeros((0, REC)), np.zeros(0, np.int32)
    sel_c = sels[0] if len(sels) == 1 else np.concatenate(sels)
    st_c = sts[0] if len(sts) == 1 else np.concatenate(sts)
    rec_c = recs[0] if len(recs) == 1 else np.concatenate(recs)
    pos = np.nonzero(st_c != BOX_FILTERED)[0]
    pos = pos[np.argsort(sel_c[pos], kind="stable")]
    kept = sel_c[pos]
    return ([annotations[i]["bbox"] for i in kept], kept, [annotations[i]["category_id"] for i in kept], rec_c[pos], st_c[pos])


def segmentations_to_masks(segmentations, H: int, W: int, device=None) -> torch.Tensor:
    """A list mixing RLE dicts and polygon part lists (what ``filter_annotations`` returns) -> (B,H,W) bool masks on the
    GPU in list order: the reference's ``np.array(segmentation_mask)`` (src/util.py:382)."""
    dev = _dev(device)
    out = torch.zeros((len(segmentations), H, W), dtype=torch.bool, device=dev)
    ri = [i for i, s in enumerate(segmentations) if isinstance(s, dict)]
    pi = [i for i, s in enumerate(segmentations) if not isinstance(s, dict)]
    if ri:
        out[torch.as_tensor(ri, device=dev)] = rle_decode([segmentations[i] for i in ri], device=dev)
    if pi:
        out[torch.as_tensor(pi, device=dev)] = poly_decode(pack_polygons([segmentations[i] for i in pi], H, W), device=dev)
    return out


def _fit_block(f, d, P, ii, k, g, si, B, H, W, stream, filt, stats, opts, rle=None, poly=None):
    """The argument block of la3d_fit_instances_ex for the run-length / polygon wrappers: used when a scheduling option
    (labelany3d_amd.options) is active for the call - the plain entry points carry none."""
    import ctypes as C

    from ._lib import FitArgs

    a = FitArgs()
    a.struct_size = C.sizeof(FitArgs)
    a.B, a.H, a.W = B, H, W
    a.depth, a.depth_plane_stride, a.image_index = _ptr(d), (H * W if P > 1 else 0), _ptr(ii)
    if rle is not None:
        a.rle_counts, a.rle_offsets = _ptr(rle[0]), _ptr(rle[1])
    else:
        a.poly_xy, a.ring_offsets, a.inst_rings = _ptr(poly[0]), _ptr(poly[1]), _ptr(poly[2])
    a.K, a.k_stride = _ptr(k), (9 if k.shape[0] > 1 else 0)
    a.ground, a.sample_idx = _ptr(g), _ptr(si)
    a.filter_boundary = -1
    if filt is not None:
        a.filter_boundary, a.filter_min_area, a.filter_max_edge = filt
        a.stats = _ptr(stats)
    a.out, a.status, a.aux = _ptr(f.boxes[0]), _ptr(f.status[0]), _ptr(f.aux[0])
    a.workspace, a.stream = _ptr(f.workspace[0]), _stream(stream)
    a.opt_engine, a.opt_launch_order, a.opt_build = opts
    return lib.la3d_fit_instances_ex(C.byref(a))


def _filter_args(filter):
    """``filter`` of fit_instances_rle / fit_instances_poly: True or a dict with the reference's three thresholds
    (src/util.py:291-326, :375): boundary strip width (10), minimum area (100), boundary pixels that make a mask "truncated" (10)."""
    f = {} if filter is True else dict(filter)
    unknown = set(f) - {"boundary_threshold", "scale_threshold", "truncation_pixels"}
    if unknown:
        raise ValueError(f"unknown filter keys: {sorted(unknown)}")
    b, a, e = int(f.get("boundary_threshold", 10)), int(f.get("scale_threshold", 100)), int(f.get("truncation_pixels", 10))
    if b < 0 or e <= 0:
        # the C-ABI reads filter_boundary < 0 or filter_max_edge <= 0 as "no filter" (a zero-initialised la3d_fit_args means none);
        # truncation_pixels <= 0 would mean "reject every instance" in the reference's rule (edge < 0 never holds): refuse it here
        # instead of silently fitting everything
        raise ValueError("filter: boundary_threshold must be >= 0 and truncation_pixels >= 1")
    return b, a, e


def fit_instances_poly(depth, polys, K, ground=None, sample_idx=None, image_index=None, stream=None, device=None, filter=None):
    """fit_instances with polygon masks: the parts are rasterised inside the fit kernel, straight into its LDS bit image
    (cv2.fillPoly semantics, reference src/util.py:386-400).  Arguments and returns as ``labelany3d_amd.fit_instances``;
    ``polys`` is the tuple from ``pack_polygons``.

    PARITY UNPINNED until ``tests/golden/g16_fillpoly.npz`` exists: the rasteriser follows a restatement of OpenCV 4.x's fillPoly
    (oracle/poly_oracle.py), not outputs of cv2 itself - OpenCV cannot be installed in the build image.  Where it can:
    ``pip install opencv-python==4.10.0.84 && python tests/golden/make_golden_fillpoly.py``, then run the polygon tests.

    ``filter=True`` (or a dict of thresholds, see ``_filter_args``) fuses the reference's instance filter (src/util.py:375, polygon
    branch: height = last row - first row + 1) into the same launch: dropped instances get status 6 and a NaN record and cost no
    passes; a fourth return value holds the (B,4) statistics (area, rows, span, edge pixels) as ``mask_stats_poly`` gives them."""
    dev = _dev(device)
    pxy, pro, pir, H, W = polys
    if W % 32 != 0:   # a frame of odd width: depth rows padded to the next multiple of 32 (fit_instances_ex: la3d_fit_args::frame_width)
        r = fit_instances_ex(depth, K, polys=polys, ground=ground, sample_idx=sample_idx, image_index=image_index, filter=filter,
                             stream=stream, device=dev)
        return (r["boxes"], r["status"], r["aux"]) + ((r["stats"],) if filter else ())
    pxy, pro, pir, ground, image_index, sample_idx = _bulk(dev, (pxy, torch.int32), (pro, torch.int64), (pir, torch.int64), (ground, torch.float64),
                                                           (image_index, torch.int32), (sample_idx, torch.int32))
    xy, ro, ir, H, W = _poly_dev((pxy, pro, pir, H, W), dev)
    B = ir.numel() - 1
    d, k, P, ii, g, si = _fit_common(depth, K, H, W, B, ground, sample_idx, image_index, dev, "polygon")
    stats = None
    with torch.cuda.device(dev):
        f = InstanceFitter(B, H, W, dev)
        if filter:
            stats = torch.zeros((B, 4), dtype=torch.int32, device=dev)
        if B == 0:
            return (f.boxes[0], f.status[0], f.aux[0]) + ((stats,) if filter else ())
        opts = options.codes()
        if any(opts):
            rc = _fit_block(f, d, P, ii, k, g, si, B, H, W, stream, _filter_args(filter) if filter else None, stats, opts, poly=(xy, ro, ir))
        elif filter:
            b, a, e = _filter_args(filter)
            rc = lib.la3d_fit_instances_poly_filtered(_ptr(d), H * W if P > 1 else 0, _ptr(ii), _ptr(xy), _ptr(ro), _ptr(ir), _ptr(k),
                                                      9 if k.shape[0] > 1 else 0, _ptr(g), _ptr(si), B, H, W, b, a, e,
                                                      _ptr(f.boxes[0]), _ptr(f.status[0]), _ptr(f.aux[0]), _ptr(stats),
                                                      _ptr(f.workspace[0]), _stream(stream))
        else:
            rc = lib.la3d_fit_instances_poly(_ptr(d), H * W if P > 1 else 0, _ptr(ii), _ptr(xy), _ptr(ro), _ptr(ir), _ptr(k),
                                             9 if k.shape[0] > 1 else 0, _ptr(g), _ptr(si), B, H, W, _ptr(f.boxes[0]),
                                             _ptr(f.status[0]), _ptr(f.aux[0]), _ptr(f.workspace[0]), _stream(stream))
        check(rc, "la3d_fit_instances_poly")
    _record(stream, d, k, ii, g, si, xy, ro, ir, stats, f.workspace, f.boxes, f.status, f.aux)
    return (f.boxes[0], f.status[0], f.aux[0]) + ((stats,) if filter else ())


def _fit_common(depth, K, H, W, B, ground, sample_idx, image_index, dev, what):
    d = _as_dev(depth, torch.float32, dev)
    if d.dim() == 2:
        d = d[None]
    if d.shape[1:] != (H, W):
        raise ValueError(f"depth planes {tuple(d.shape[1:])} do not match the {what} frame {(H, W)}")
    k = _as_dev(K, torch.float64, dev, cache=True)
    if k.dim() == 2:
        k = k[None]
    P = d.shape[0]
    if k.shape[0] == 1 and P > 1:
        k = k.expand(P, 3, 3).contiguous()
    ii = None if image_index is None else _as_dev(image_index, torch.int32, dev)
    if ii is None and P not in (1, B):
        raise ValueError("without image_index, depth must have 1 or B planes")
    g = None if ground is None else _as_dev(ground, torch.float64, dev)
    si = None if sample_idx is None else _as_dev(sample_idx, torch.int32, dev)
    return d, k, P, ii, g, si


def fit_instances_rle(depth, rles, K, ground=None, sample_idx=None, image_index=None, stream=None, device=None, filter=None):
    """fit_instances with run-length masks: the runs are decoded inside the fit kernel, straight into its LDS
    bit image.  Arguments and returns as ``labelany3d_amd.fit_instances``; ``rles`` is a list of COCO RLE
    objects or the tuple from ``pack_rle``.  ``filter``: as in ``fit_instances_poly`` (RLE branch of the rule: height = rows
    holding a pixel, src/util.py:368-369)."""
    counts, offsets, H, W = pack_rle(rles)
    dev = _dev(device)
    if W % 32 != 0:   # (as in fit_instances_poly)
        r = fit_instances_ex(depth, K, rles=(counts, offsets, H, W), ground=ground, sample_idx=sample_idx, image_index=image_index,
                             filter=filter, stream=stream, device=dev)
        return (r["boxes"], r["status"], r["aux"]) + ((r["stats"],) if filter else ())
    counts, offsets, ground, image_index, sample_idx = _bulk(dev, (counts, torch.int32), (offsets, torch.int64), (ground, torch.float64),
                                                             (image_index, torch.int32), (sample_idx, torch.int32))
    c, o = _as_dev(counts, torch.int32, dev), _as_dev(offsets, torch.int64, dev)
    B = o.numel() - 1
    d, k, P, ii, g, si = _fit_common(depth, K, H, W, B, ground, sample_idx, image_index, dev, "RLE")
    stats = None
    with torch.cuda.device(dev):
        f = InstanceFitter(B, H, W, dev)
        if filter:
            stats = torch.zeros((B, 4), dtype=torch.int32, device=dev)
        if B == 0:
            return (f.boxes[0], f.status[0], f.aux[0]) + ((stats,) if filter else ())
        opts = options.codes()
        if any(opts):
            rc = _fit_block(f, d, P, ii, k, g, si, B, H, W, stream, _filter_args(filter) if filter else None, stats, opts, rle=(c, o))
        elif filter:
            b, a, e = _filter_args(filter)
            rc = lib.la3d_fit_instances_rle_filtered(_ptr(d), H * W if P > 1 else 0, _ptr(ii), _ptr(c), _ptr(o), _ptr(k),
                                                     9 if k.shape[0] > 1 else 0, _ptr(g), _ptr(si), B, H, W, b, a, e, _ptr(f.boxes[0]),
                                                     _ptr(f.status[0]), _ptr(f.aux[0]), _ptr(stats), _ptr(f.workspace[0]), _stream(stream))
        else:
            rc = lib.la3d_fit_instances_rle(_ptr(d), H * W if P > 1 else 0, _ptr(ii), _ptr(c), _ptr(o), _ptr(k),
                                            9 if k.shape[0] > 1 else 0, _ptr(g), _ptr(si), B, H, W, _ptr(f.boxes[0]),
                                            _ptr(f.status[0]), _ptr(f.aux[0]), _ptr(f.workspace[0]), _stream(stream))
        check(rc, "la3d_fit_instances_rle")
    _record(stream, d, k, ii, g, si, c, o, stats, f.workspace, f.boxes, f.status, f.aux)
    return (f.boxes[0], f.status[0], f.aux[0]) + ((stats,) if filter else ())


def masked_ratio_median(depth_map, depth_render, mask, render_mask=None, image_index=None, stream=None):
    """Per instance ``np.median(depth_map[overlap] / depth_render[overlap])`` with ``overlap = mask & render_mask``
    — the scale estimate of the reference's align_to_depth_match (src/util.py:476-486), exact (radix select on
    the float32 ratios).  depth_map (P,H,W) or (H,W); depth_render, mask, render_mask (B,H,W).
    Returns (median float32 (B,), count int32 (B,)) on the GPU; an empty overlap gives count 0 and NaN (the
    reference returns the identity transform there, :476-478)."""
    dev = mask.device if isinstance(mask, torch.Tensor) and mask.is_cuda else _dev()
    m = _as_dev(mask, torch.uint8, dev)
    B, H, W = m.shape
    rm = None if render_mask is None else _as_dev(render_mask, torch.uint8, dev)
    d = _as_dev(depth_map, torch.float32, dev)
    if d.dim() == 2:
        d = d[None]
    r = _as_dev(depth_render, torch.float32, dev)
    ii = None if image_index is None else _as_dev(image_index, torch.int32, dev)
    if ii is None and d.shape[0] not in (1, B):
        raise ValueError("without image_index, depth_map must have 1 or B planes")
    med = torch.empty(B, dtype=torch.float32, device=dev)
    cnt = torch.empty(B, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(lib.la3d_masked_ratio_median(_ptr(d), H * W if d.shape[0] > 1 else 0, _ptr(ii), _ptr(r), _ptr(m), _ptr(rm), B, H, W,
                                           _ptr(med), _ptr(cnt), _stream(stream)), "la3d_masked_ratio_median")
    return med, cnt
