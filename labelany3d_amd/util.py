"""Drop-in for the hot-path function of the reference's ``util`` module: ``depth_to_points``
(reference src/util.py:52-75).  Runs on the MI355X through ``la3d_unproject`` (include/la3d.h)."""
from __future__ import annotations

import numpy as np

from ._lib import check, lib
from .batched import unproject


def depth_to_points(depth, K=None, R=None, t=None):
    """Pinhole back-projection of a depth batch; returns the points of batch element 0 as
    ``(H, W, 3)`` float64, like the reference (:75).  ``u`` = column, ``v`` = row, no half-pixel
    offset (:62-69).  NumPy in -> NumPy out; a CUDA tensor in -> a CUDA tensor out (no copy back).

    As in the reference, ``K=None`` is not usable (``np.linalg.inv(None)`` raises there, :56).
    """
    if K is None:
        raise TypeError("depth_to_points: K is required (the reference fails in np.linalg.inv(None))")
    try:
        import torch

        is_t = isinstance(depth, torch.Tensor)
    except ImportError:  # pragma: no cover
        is_t = False
    if is_t:
        if depth.dim() != 3:
            raise ValueError("depth must be (B, H, W)")
        return unproject(depth[0], K, R, t)
    d = np.asarray(depth)
    if d.ndim != 3:
        raise ValueError("depth must be (B, H, W)")
    # host frame in, host points out: ONE C call (la3d_unproject_host: upload, kernel, download on the library's private stream)
    d0 = np.ascontiguousarray(d[0], dtype=np.float32)
    k = np.ascontiguousarray(np.asarray(K, dtype=np.float64).reshape(3, 3))
    rt = None
    if R is not None or t is not None:
        rt = np.concatenate([(np.eye(3) if R is None else np.asarray(R, dtype=np.float64)).reshape(-1),
                             (np.zeros(3) if t is None else np.asarray(t, dtype=np.float64)).reshape(-1)])
    out = np.empty(d0.shape + (3,), np.float64)
    check(lib.la3d_unproject_host(d0.ctypes.data, k.ctypes.data, None if rt is None else rt.ctypes.data, d0.shape[0], d0.shape[1],
                                  out.ctypes.data, 1), "la3d_unproject_host")
    return out
