"""Drop-in for the hot-path function of the reference's ``util`` module: ``depth_to_points``
(reference src/util.py:52-75).  Runs on the MI355X through ``la3d_unproject`` (include/la3d.h)."""
from __future__ import annotations

import numpy as np

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
    return unproject(np.ascontiguousarray(d[0], dtype=np.float32), K, R, t).cpu().numpy()
