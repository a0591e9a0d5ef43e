"""The C writer of the reference's per-scene box file (la3d_format_3dbbox_json, csrc/la3d_json.cpp): the text must be what
``json.dump`` writes for the reference's list of six-key dicts (src/util_3dbox.py:283-292), byte for byte - floats as
float.__repr__ prints them, default separators, escaped names.  Host code only: runs without a GPU."""
import json

import numpy as np
import pytest

from labelany3d_amd.jsonout import SceneRecords, format_scenes


def _want(rec, rows, obj, nid, names, lo, hi):
    out = []
    for k in range(lo, hi):
        r = rec[rows[k]]
        out.append(dict(obj_id=str(int(obj[k])), category_name=names[nid[k]], center_cam=r[0:3].tolist(), R_cam=r[6:15].reshape(3, 3).tolist(),
                        dimensions=[float(x) for x in r[3:6]], bbox3D_cam=r[15:39].reshape(8, 3).tolist()))
    return out


def test_text_is_byte_identical_to_json_dumps():
    rs = np.random.RandomState(0)
    n = 600
    rec = rs.randn(n, 39) * np.exp(rs.uniform(-14, 14, (n, 39)))
    special = [0.0, -0.0, 1e5, 1e16, 1e15, 9999999999999998.0, 1e-4, 9.999e-5, 1e-5, 5e-324, 1.7976931348623157e308, 123456789012345678.0,
               0.1, 1 / 3, 100.0, 1.0, -1.0, 2.5e-7, float("nan"), float("inf"), -float("inf"), 65504.0, 0.5, 2.0 ** -14, 1e22, 1e23]
    rec[0, :len(special)] = special
    rec[1] = np.round(rs.randn(39) * 1000)                       # whole numbers print with '.0'
    rec[2] = (rs.randn(39) * 8).astype(np.float16)              # fp16-quantised values, like bbox3D_cam
    names = ["chair", "dining table", "unknown", 'we"ird\\näm€\t']
    rows = rs.permutation(n)[:400].astype(np.int64)
    scene_off = np.array([0, 0, 5, 5, 150, 400], np.int64)       # empty scenes write '[]'
    obj = (np.arange(400) % 23).astype(np.int32)
    nid = rs.randint(0, len(names), 400).astype(np.int32)
    texts = format_scenes(rec, rows, obj, nid, scene_off, names)
    assert len(texts) == 5 and texts[0] == b"[]" and texts[2] == b"[]"
    for s in range(5):
        assert texts[s] == json.dumps(_want(rec, rows, obj, nid, names, scene_off[s], scene_off[s + 1])).encode(), s
    sr = SceneRecords(texts[4], 250)
    assert len(sr) == 250 and sr[0]["obj_id"] == str(int(obj[150])) and list(sr[3]) == ["obj_id", "category_name", "center_cam", "R_cam", "dimensions", "bbox3D_cam"]
    assert sr == json.loads(texts[4])


def test_bad_indices_are_refused():
    rec = np.zeros((3, 39))
    with pytest.raises(ValueError):
        format_scenes(rec, np.array([0, 5]), np.array([0, 1]), np.array([0, 0]), np.array([0, 2]), ["a"])
    with pytest.raises(ValueError):
        format_scenes(rec, np.array([0, 1]), np.array([0, 1]), np.array([0, 3]), np.array([0, 2]), ["a"])
    assert format_scenes(rec, np.zeros(0, np.int64), np.zeros(0, np.int32), np.zeros(0, np.int32), np.array([0, 0, 0]), []) == [b"[]", b"[]"]


def test_float_text_over_every_exponent():
    """float.__repr__ (shortest round-trip digits, exponent form below 1e-4 and from 1e16 on) reproduced for random BIT PATTERNS
    (every exponent, subnormals, NaN / inf payloads), for decimal-looking values and their neighbours, for powers of ten and their
    neighbours: 350 000 doubles, byte for byte (a 1.4 M-value run of the same generator in round 6: no difference)."""
    rs = np.random.RandomState(123)
    n = 3000
    for it in range(3):
        if it == 0:
            with np.errstate(invalid="ignore"):   # (signalling-NaN bit patterns among the random ones)
                rec = rs.randint(0, 2 ** 63, size=(n, 39), dtype=np.int64).view(np.float64) * np.where(rs.rand(n, 39) < 0.5, -1, 1)
        elif it == 1:
            base = np.round(rs.randn(n, 39) * 10.0 ** rs.randint(-8, 18, (n, 39)), 3)
            rec = np.nextafter(base, np.where(rs.rand(n, 39) < 0.5, np.inf, -np.inf))
        else:
            with np.errstate(over="ignore", under="ignore"):
                rec = 10.0 ** rs.randint(-320, 309, (n, 39)).astype(np.float64)
            k = rs.randint(-2, 3, (n, 39))
            for _ in range(2):
                rec = np.where(k > 0, np.nextafter(rec, np.inf), np.where(k < 0, np.nextafter(rec, -np.inf), rec))
        rows = np.arange(n, dtype=np.int64)
        zeros = np.zeros(n, np.int32)
        txt = format_scenes(rec, rows, zeros, zeros, np.array([0, n], np.int64), ["x"])[0]
        assert txt == json.dumps(_want(rec, rows, zeros, zeros, ["x"], 0, n)).encode(), it
