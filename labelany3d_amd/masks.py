"""Mask ingestion on the device: the data format immediately upstream of the box fit.

The reference turns COCO / COCONut annotations into ``(N,H,W)`` bool arrays on the CPU with pycocotools
(``mask_utils.decode``, reference src/util.py:367,401-402; encoder src/download_coconut.py:167-175) and keeps
an instance when ``height/H > 0.0625 and not truncated and area >= 100`` (src/util.py:291-335, :375).  Here the
run lengths go to the GPU as they are (a few hundred bytes per instance instead of H*W bytes):

    counts, offsets, H, W = pack_rle(annotation_segmentations)        # host: list of COCO RLE dicts
    boxes, status, aux = fit_instances_rle(depth, (counts, offsets, H, W), K, ...)   # no dense mask at all
    masks = rle_decode((counts, offsets, H, W))                         # the reference's mask array, on the GPU
    keep  = keep_instances(mask_stats(masks), H, from_rle=True)         # the reference's three filters
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ._lib import AUX, REC, check, lib
from .batched import InstanceFitter, _as_dev, _dev, _ptr, _stream


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
    parts = []
    for r in rles:
        c = r["counts"]
        parts.append(rle_from_string(c) if isinstance(c, (str, bytes)) else np.asarray(c, dtype=np.int32))
    offsets = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int64)
    counts = np.concatenate(parts).astype(np.int32) if parts and offsets[-1] else np.zeros(1, np.int32)
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


def filter_annotations(annotations, image_size, boundary_threshold: int = 10, scale_threshold: int = 100, device=None):
    """The RLE branch of the reference's ``read_bounding_boxes_segmentations(annotations, image_size)``
    (src/util.py:336-383) without ever decoding a mask plane: crowd annotations are skipped (:355-357), the four filter
    quantities come from ``mask_stats_rle`` and the keep rule is :375 (height/H > 0.0625, not truncated, area >= 100).

    annotations: list of COCO annotation dicts with 'iscrowd', 'bbox', 'category_id' and an RLE 'segmentation'
    ({'size': [h, w], 'counts': list | str | bytes}); image_size = (width, height) as in the reference.
    Returns ``(bboxes, rles, kept_index, category_ids)``: the kept annotations' boxes, their RLE objects (feed them to
    ``fit_instances_rle`` or ``rle_decode``), their positions in ``annotations`` and their raw COCO category ids (the
    reference maps those to super-category names with a table that is not part of this path).  Polygon segmentations
    need ``cv2.fillPoly`` (src/util.py:386-392) and raise NotImplementedError."""
    cand, idx = [], []
    for i, a in enumerate(annotations):
        if a.get("iscrowd"):
            continue
        seg = a.get("segmentation")
        if seg is None:
            continue
        if not (isinstance(seg, dict) and "counts" in seg):
            raise NotImplementedError("polygon segmentation: rasterise with cv2.fillPoly (reference src/util.py:386-392)")
        cand.append({"size": seg["size"], "counts": seg["counts"]})
        idx.append(i)
    if not cand:
        return [], [], np.zeros(0, np.int64), []
    H = int(image_size[1])
    keep = keep_instances(mask_stats_rle(cand, boundary_threshold, device=device), H, from_rle=True,
                          scale_threshold=scale_threshold).cpu().numpy()
    kept = [j for j in range(len(cand)) if keep[j]]
    return ([annotations[idx[j]]["bbox"] for j in kept], [cand[j] for j in kept], np.asarray([idx[j] for j in kept], np.int64),
            [annotations[idx[j]]["category_id"] for j in kept])


def fit_instances_rle(depth, rles, K, ground=None, sample_idx=None, image_index=None, stream=None, device=None):
    """fit_instances with run-length masks: the runs are decoded inside the fit kernel, straight into its LDS
    bit image.  Arguments and returns as ``labelany3d_amd.fit_instances``; ``rles`` is a list of COCO RLE
    objects or the tuple from ``pack_rle``."""
    counts, offsets, H, W = pack_rle(rles)
    dev = _dev(device)
    c, o = _as_dev(counts, torch.int32, dev), _as_dev(offsets, torch.int64, dev)
    B = o.numel() - 1
    d = _as_dev(depth, torch.float32, dev)
    if d.dim() == 2:
        d = d[None]
    if d.shape[1:] != (H, W):
        raise ValueError(f"depth planes {tuple(d.shape[1:])} do not match the RLE frame {(H, W)}")
    k = _as_dev(K, torch.float64, dev)
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
    with torch.cuda.device(dev):
        f = InstanceFitter(B, H, W, dev)
        if B == 0:
            return f.boxes[0], f.status[0], f.aux[0]
        rc = lib.la3d_fit_instances_rle(_ptr(d), H * W if P > 1 else 0, _ptr(ii), _ptr(c), _ptr(o), _ptr(k),
                                        9 if k.shape[0] > 1 else 0, _ptr(g), _ptr(si), B, H, W, _ptr(f.boxes[0]),
                                        _ptr(f.status[0]), _ptr(f.aux[0]), _ptr(f.workspace[0]), _stream(stream))
        check(rc, "la3d_fit_instances_rle")
    return f.boxes[0], f.status[0], f.aux[0]


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
