"""``3dbbox.json`` text straight from packed (n, 39) records (C writer ``la3d_format_3dbbox_json``, csrc/la3d_json.cpp): the bytes
``json.dump`` writes for the reference's list of six-key dicts (src/util_3dbox.py:283-292), for many scenes per call and without a
Python object per record.  ``SceneRecords`` is the lazy list a caller gets: ``len()`` is free, the dicts appear when somebody looks."""
from __future__ import annotations

import ctypes as C
import json
from typing import List, Sequence

import numpy as np

from ._lib import REC, lib


def format_scenes(records: np.ndarray, rows: np.ndarray, obj_ids: np.ndarray, name_ids: np.ndarray, scene_off: np.ndarray,
                  names: Sequence[str]) -> List[bytes]:
    """One JSON text per scene.  records (m, 39) f64 host array; scene s owns entries scene_off[s]:scene_off[s+1] of ``rows`` (row of
    ``records``), ``obj_ids`` (the object's id: index among the kept instances of the image) and ``name_ids`` (index into ``names``)."""
    records = np.ascontiguousarray(records, np.float64).reshape(-1, REC)
    rows = np.ascontiguousarray(rows, np.int64)
    obj_ids = np.ascontiguousarray(obj_ids, np.int32)
    name_ids = np.ascontiguousarray(name_ids, np.int32)
    scene_off = np.ascontiguousarray(scene_off, np.int64)
    S = len(scene_off) - 1
    n = int(scene_off[-1]) if S >= 0 and len(scene_off) else 0
    if S <= 0:
        return []
    if rows.shape[0] < n or obj_ids.shape[0] < n or name_ids.shape[0] < n:
        raise ValueError("rows / obj_ids / name_ids shorter than scene_off[-1]")
    if n and (rows[:n].min() < 0 or rows[:n].max() >= records.shape[0] or name_ids[:n].min() < 0 or name_ids[:n].max() >= len(names)):
        raise ValueError("row or name index out of range")
    enc = [json.dumps(s).encode() for s in names]      # escaped + quoted, as json.dump writes a string value
    arr = (C.c_char_p * max(len(enc), 1))(*enc)
    per = np.asarray([len(e) for e in enc], np.int64)
    cap = int(lib.la3d_3dbbox_json_bound(n, int(per[name_ids[:n]].sum()) if n else 0, S))
    out = np.empty(cap, np.uint8)
    toff = np.empty(S + 1, np.int64)
    wrote = lib.la3d_format_3dbbox_json(records.ctypes.data, rows.ctypes.data, obj_ids.ctypes.data, name_ids.ctypes.data,
                                        scene_off.ctypes.data, S, arr, out.ctypes.data, cap, toff.ctypes.data)
    if wrote < 0:
        raise RuntimeError("la3d_format_3dbbox_json: buffer too small")
    buf = out[:wrote].tobytes()
    return [buf[toff[s]:toff[s + 1]] for s in range(S)]


class SceneRecords(Sequence):
    """The records of one scene: the JSON text of its ``3dbbox.json`` plus a count; behaves like the list of dicts (parsed on first
    access - what the file would read back as)."""
    __slots__ = ("text", "_n", "_items")

    def __init__(self, text: bytes, n: int):
        self.text, self._n, self._items = text, int(n), None

    def _load(self):
        if self._items is None:
            self._items = json.loads(self.text)
        return self._items

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        return self._load()[i]

    def __iter__(self):
        return iter(self._load())

    def __eq__(self, other):
        return list(self) == list(other)

    def __repr__(self):
        return f"SceneRecords({self._n} records, {len(self.text)} bytes)"
