"""Drop-in for the reference's ``cam_utils`` module (reference src/cam_utils.py:4-52): orbit-camera
pose helpers.  Not on the data path (no importer in the reference's ``src/``); host-side NumPy, kept
for signature compatibility.  One deliberate deviation: ``length`` on a torch tensor works here (the
reference calls an undefined ``dot`` there and raises NameError, :8)."""
from __future__ import annotations

import numpy as np

_UP = np.array([0, 1, 0], dtype=np.float32)


def length(x, eps=1e-20):
    """Row norms clamped below by sqrt(eps), keepdims (reference :4-8)."""
    if isinstance(x, np.ndarray):
        return np.sqrt(np.maximum((x * x).sum(axis=-1, keepdims=True), eps))
    import torch

    return torch.sqrt(torch.clamp((x * x).sum(-1, keepdim=True), min=eps))


def safe_normalize(x, eps=1e-20):
    """x / length(x) (reference :10-11)."""
    return x / length(x, eps)


def look_at(campos, target, opengl=True):
    """[N,3] eye and target -> [N,3,3] rotation whose columns are (right, up, forward)
    (reference :14-31).  OpenGL: forward = eye - target (camera looks down -z)."""
    sign = 1.0 if opengl else -1.0
    forward = safe_normalize(sign * (campos - target))
    # right-handed basis: right = up x forward (OpenGL) / forward x up (otherwise)
    right = safe_normalize(np.cross(_UP, forward) if opengl else np.cross(forward, _UP))
    up = safe_normalize(np.cross(forward, right) if opengl else np.cross(right, forward))
    return np.stack([right, up, forward], axis=1)


def orbit_camera(elevation, azimuth, radius=1, is_degree=True, target=None, opengl=True):
    """Elevation / azimuth (degrees by default) -> 4x4 float32 cam2world pose on a sphere of
    ``radius`` around ``target`` (reference :35-52).  +elevation moves the eye towards -y."""
    if is_degree:
        elevation, azimuth = np.deg2rad(elevation), np.deg2rad(azimuth)
    ce = np.cos(elevation)
    eye = np.array([radius * ce * np.sin(azimuth), -radius * np.sin(elevation), radius * ce * np.cos(azimuth)])
    if target is None:
        target = np.zeros([3], dtype=np.float32)
    eye = eye + target
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = look_at(eye, target, opengl)
    pose[:3, 3] = eye
    return pose
