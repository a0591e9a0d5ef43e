"""Realistic polygon fixtures: the shapes the reference's OWN converter writes.  ``binary_mask_to_polygon`` of
/root/reference/src/download_coconut.py:178-199 (scikit-image ``find_contours`` at 0.5 on the padded mask, ``approximate_polygon``
with tolerance 2, x/y flip, negative -> 0) is RUN on synthetic binary masks: blobs, ellipses, shapes with holes, several
components, shapes touching the frame.  Written (data only): every mask as column-major run lengths + the polygon list the
converter returned (half-pixel float coordinates, exactly what lands in the annotation JSON).

scikit-image exists only in this container's second interpreter, so run it with that one:

    /opt/conda/bin/python3.9 tests/golden/make_golden_polygons.py

(`datasets`, `pycocotools`, `tqdm` — imported at the top of download_coconut.py but not used by this function — are replaced by
empty stand-in modules for the import.)  No-op where /root/reference is absent.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def masks():
    rs = np.random.RandomState(2024)
    out = []
    for H, W in ((48, 64), (96, 128), (240, 320), (480, 640)):
        yy, xx = np.mgrid[0:H, 0:W]
        # ellipse
        out.append(((yy - H * 0.45) / (H * 0.3)) ** 2 + ((xx - W * 0.55) / (W * 0.22)) ** 2 < 1)
        # smooth random blob(s): low-pass noise thresholded -> several components, ragged outlines
        f = rs.randn(H // 8 + 2, W // 8 + 2)
        f = np.kron(f, np.ones((8, 8)))[:H, :W]
        for _ in range(3):
            f = (f + np.roll(f, 3, 0) + np.roll(f, -3, 0) + np.roll(f, 3, 1) + np.roll(f, -3, 1)) / 5
        out.append(f > 0.25)
        # ring (hole) + island inside the hole
        r2 = ((yy - H / 2) / (H * 0.4)) ** 2 + ((xx - W / 2) / (W * 0.4)) ** 2
        out.append(((r2 < 1) & (r2 > 0.35)) | (r2 < 0.05))
        # touching the frame on two sides, with a notch
        m = np.zeros((H, W), bool)
        m[: H // 2, : W // 3] = True
        m[H // 8: H // 4, W // 6: W // 3] = False
        out.append(m)
        # thin diagonal band
        out.append(np.abs((yy - H / 2) - 0.6 * (xx - W / 2)) < 3.2)
    return out


def rle_counts(mask):
    """uncompressed COCO run lengths (column-major, zeros first)"""
    flat = np.asarray(mask, bool).ravel(order="F")
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    edges = np.concatenate([[0], change, [flat.size]])
    counts = np.diff(edges).tolist()
    return ([0] if flat[0] else []) + counts


def main():
    path = "/root/reference/src/download_coconut.py"
    if not os.path.exists(path):
        print("reference not present; nothing generated")
        return 0
    for name in ("datasets", "pycocotools", "pycocotools.mask", "tqdm"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:  # noqa: BLE001
                sys.modules[name] = types.ModuleType(name)
    for name, attr, val in (("datasets", "load_dataset", None), ("tqdm", "tqdm", lambda x, *a, **k: x)):
        if not hasattr(sys.modules[name], attr):
            setattr(sys.modules[name], attr, val)
    if not hasattr(sys.modules["pycocotools"], "mask"):
        sys.modules["pycocotools"].mask = sys.modules["pycocotools.mask"]
    spec = importlib.util.spec_from_file_location("_la3d_ref_download_coconut", path)
    dc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dc)
    cases = []
    for m in masks():
        polys = dc.binary_mask_to_polygon(m.astype(np.uint8), tolerance=2)
        cases.append({"size": list(m.shape), "counts": rle_counts(m), "polygons": polys})
    import gzip

    with gzip.open(os.path.join(HERE, "g13_polygons.json.gz"), "wt", compresslevel=9) as f:
        json.dump({"cases": cases}, f)
    print("wrote g13_polygons.json.gz:", len(cases), "masks,", sum(len(c["polygons"]) for c in cases), "polygon parts,",
          os.path.getsize(os.path.join(HERE, "g13_polygons.json.gz")), "bytes")
    return 0


if __name__ == "__main__":
    sys.exit(main())
