"""Masked depth statistics next to the box fit: the data-parallel parts of the reference's two depth
alignment helpers, on the MI355X.

* ``align_depth`` — drop-in for reference src/batch_scripts/depth.py:52-92 (stage 1: MoGe depth aligned to DepthPro's
  metric scale).  The valid-pixel selection / compaction (:67-78) and the prediction scatter (:82-90) are HIP kernels
  (``la3d_align_select`` / ``la3d_align_apply``); the estimator in between is the reference's own third-party call,
  ``sklearn.linear_model.RANSACRegressor(LinearRegression(fit_intercept=False), min_samples=...)``, fed the same samples in
  the same (row-major) order and drawing from the same global NumPy stream, so a seeded run reproduces the reference.
* ``depth_match_transform`` — the arithmetic of ``align_to_depth_match`` (src/util.py:464-494) after its renderer call:
  overlap = mask & render alpha; scale = median(depth_map[overlap] / depth_render[overlap]) (``la3d_masked_ratio_median``,
  exact radix select on the GPU); transform = [inv(R[:3,:3]) * scale | T * scale].
"""
from __future__ import annotations

import numpy as np
import torch

from ._lib import check, lib
from .batched import _as_dev, _dev, _ptr, _stream
from .masks import masked_ratio_median


def align_select(relative_depth, metric_depth, mask=None, max_valid_depth: float = 400.0, device=None, stream=None):
    """``relative[valid], metric[valid]`` with ``valid = ~isinf(relative) & (metric < max_valid_depth) [& mask]`` in row-major
    order (reference depth.py:67-78), on the GPU.  Returns two float32 tensors of equal length."""
    dev = _dev(device)
    rel = _as_dev(relative_depth, torch.float32, dev).reshape(-1)
    met = _as_dev(metric_depth, torch.float32, dev).reshape(-1)
    if rel.numel() != met.numel():
        raise ValueError("relative_depth and metric_depth differ in size")
    m = None if mask is None else _as_dev(mask, torch.uint8, dev).reshape(-1)
    n = rel.numel()
    ro, mo = torch.empty(n, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.float32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    ws = torch.empty(int(lib.la3d_align_workspace_bytes(n)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(lib.la3d_align_select(_ptr(rel), _ptr(met), _ptr(m), n, float(max_valid_depth), _ptr(ro), _ptr(mo), _ptr(cnt),
                                    _ptr(ws), _stream(stream)), "la3d_align_select")
    k = int(cnt.item())
    return ro[:k], mo[:k]


def align_select_batch(relative_depth, metric_depth, mask=None, max_valid_depth: float = 400.0, device=None, stream=None):
    """``align_select`` for a stack of P frames in one call (C-ABI ``la3d_align_select_batch``; the reference loops over images,
    src/batch_scripts/depth.py:138-160): ``relative_depth`` / ``metric_depth`` (P,H,W) [``mask`` (P,H,W)] ->
    ``(rel_sel (P,n), met_sel (P,n), counts (P,) int64)``, all on the GPU and without any host synchronisation; frame p's selection
    is ``rel_sel[p, :counts[p]]`` in row-major order."""
    dev = _dev(device)
    rel = _as_dev(relative_depth, torch.float32, dev)
    met = _as_dev(metric_depth, torch.float32, dev)
    if rel.dim() < 2 or rel.shape != met.shape:
        raise ValueError("relative_depth and metric_depth must be stacks of equal shape (P, ...)")
    P = rel.shape[0]
    rel, met = rel.reshape(P, -1), met.reshape(P, -1)
    n = rel.shape[1]
    m = None
    if mask is not None:
        m = _as_dev(mask, torch.uint8, dev).reshape(P, -1)
        if m.shape != rel.shape:
            raise ValueError("mask must have the shape of the depth stack")
    ro, mo = torch.empty((P, n), dtype=torch.float32, device=dev), torch.empty((P, n), dtype=torch.float32, device=dev)
    cnt = torch.zeros(P, dtype=torch.int64, device=dev)
    ws = torch.empty(max(P, 1) * int(lib.la3d_align_workspace_bytes(n)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(lib.la3d_align_select_batch(_ptr(rel), _ptr(met), _ptr(m), P, n, float(max_valid_depth), _ptr(ro), _ptr(mo), _ptr(cnt),
                                          _ptr(ws), _stream(stream)), "la3d_align_select_batch")
    return ro, mo, cnt


def align_apply(relative_depth, coef, intercept=0.0, mask=None, fill: float = 10000.0, device=None, stream=None):
    """``depth = full(fill); depth[sel] = relative[sel] * coef + intercept`` in float32, ``sel`` = mask if given else
    ``~isinf(relative)`` (reference depth.py:82-90).  Returns a float32 tensor shaped like ``relative_depth``."""
    dev = _dev(device)
    rel = _as_dev(relative_depth, torch.float32, dev)
    m = None if mask is None else _as_dev(mask, torch.uint8, dev)
    out = torch.empty_like(rel)
    with torch.cuda.device(dev):
        check(lib.la3d_align_apply(_ptr(rel), _ptr(m), rel.numel(), float(np.float32(coef)), float(np.float32(intercept)), float(fill),
                                   _ptr(out), _stream(stream)), "la3d_align_apply")
    return out


def align_depth(relative_depth, metric_depth, mask=None, min_samples=0.2, max_valid_depth=400.0):
    """Drop-in for the reference's ``align_depth`` (src/batch_scripts/depth.py:52-92): same arguments, same return (a float32
    array shaped like the input; ``metric_depth`` itself when nothing is valid or the fit fails), same printed messages,
    same consumption of the global NumPy stream by RANSAC."""
    from sklearn.linear_model import LinearRegression, RANSACRegressor   # the reference's estimator (depth.py:30, :66)

    regressor = RANSACRegressor(estimator=LinearRegression(fit_intercept=False), min_samples=min_samples)
    rel_sel, met_sel = align_select(relative_depth, metric_depth, mask, max_valid_depth)
    if rel_sel.numel() == 0:
        print("Warning: No valid points for alignment. Returning metric depth.")
        return metric_depth
    try:
        regressor.fit(rel_sel.cpu().numpy().reshape(-1, 1), met_sel.cpu().numpy().reshape(-1, 1))
    except Exception as e:  # noqa: BLE001 - the reference catches everything here (:76-78)
        print(f"Error fitting RANSACRegressor: {e}, using metric depth directly")
        return metric_depth
    est = regressor.estimator_
    coef = np.asarray(est.coef_, dtype=np.float32).reshape(-1)[0]
    intercept = np.asarray(est.intercept_, dtype=np.float32).reshape(-1)[0]
    out = align_apply(relative_depth, coef, intercept, mask)
    if isinstance(relative_depth, torch.Tensor):
        return out
    return out.cpu().numpy().reshape(np.asarray(relative_depth).shape)


def _as_batch(x):
    return x[None] if isinstance(x, torch.Tensor) else np.asarray(x)[None]


def depth_match_transform(mask, depth_map, R, T, render_alpha_mask, depth_render):
    """``align_to_depth_match`` (reference src/util.py:464-494) after its ``process_object`` call: returns the 4x4 transform
    ``[inv(R[:3,:3]) * scale | T[:3] * scale]`` with ``scale`` the median depth ratio over ``mask & render_alpha_mask``, or
    ``np.eye(4)`` (and the reference's message) when the two masks do not overlap."""
    med, cnt = masked_ratio_median(_as_batch(depth_map), _as_batch(depth_render), _as_batch(mask), _as_batch(render_alpha_mask))
    if int(cnt[0]) == 0:
        print("No overlap between masks found")
        return np.eye(4)
    scale = np.float32(med[0].item())   # a float32 value, as np.median of a float32 array
    transform = np.eye(4)
    transform[:3, :3] = np.linalg.inv(np.asarray(R)[:3, :3]) * scale
    transform[:3, -1] = np.asarray(T)[:3] * scale
    return transform
