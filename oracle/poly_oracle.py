"""CPU ORACLE for polygon mask ingestion — test infrastructure, NOT product code.

The reference turns polygon segmentations into boolean masks with OpenCV
(``create_boolean_mask_from_polygon``, /root/reference/src/util.py:386-415):

    for polygon in segmentation:
        points = np.array(polygon).reshape(-1, 2).astype(np.int32)
        cv2.fillPoly(mask, [points], color=1)            # one call per part, default LINE_8, shift 0

and every kept COCONut instance takes this branch (the converter writes polygons for all non-crowd
instances, src/download_coconut.py:275-280; crowds are skipped at src/util.py:355-357).

PARITY UNPINNED: ``cv2`` (reference pin opencv-python==4.10.0.84, requirements.txt:6) is not installed in
the build container and there is no network, so no golden vector could be generated from OpenCV itself.
This file restates the published algorithm of OpenCV 4.x ``modules/imgproc/src/drawing.cpp`` for the case
the reference uses (8-bit single channel, ``line_type = LINE_8``, ``shift = 0``, ``offset = (0,0)``):

  fillPoly            = CollectPolyEdges (per contour) + FillEdgeCollection
  CollectPolyEdges    : every polygon side is DRAWN with ``Line`` (8-connected Bresenham of
                        ``LineIterator``, left-to-right, clipped by ``clipLine``) and, unless horizontal,
                        becomes a PolyEdge {y0 < y1, x (16.16 fixed point at y0), dx per scanline};
                        x carries +0.5 (XY_ONE/2) when both end points lie inside the image, and the
                        ``clipLine``-clipped end points without the half when one lies outside.
  FillEdgeCollection  : for y in [min y0, min(max y1, rows)): the edges with y0 <= y < y1, ordered by their
                        current x, are paired (even-odd); each pair fills columns x_left >> 16 ... x_right >> 16
                        inclusive (clipped to the row); then x += dx for both.  The bottom scanline of an
                        edge (y == y1) is not swept — the outline drawn by ``Line`` covers it.

What IS pinned (tests/test_oracle_poly.py): axis-aligned integer rectangles fill exactly
[x0..x1] x [y0..y1]; strictly interior pixel centres of convex integer polygons are always set and nothing
farther than one pixel from the polygon is; the Bresenham closed form equals the iterator; the result does
not depend on the starting vertex or orientation of a ring; Pillow's independent polygon rasteriser agrees
on the interior of convex polygons.

Only ``tests/`` may import this module.
"""
from __future__ import annotations

import numpy as np

XY_SHIFT = 16
XY_ONE = 1 << XY_SHIFT


def _cdiv(a: int, b: int) -> int:
    """C++ integer division (truncation toward zero) on Python ints."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b > 0) else -q


def _ctrunc(x: float) -> int:
    """(int64)(double) conversion: truncation toward zero."""
    return int(x)


def clip_line(width: int, height: int, p1, p2):
    """cv::clipLine(Size2l, Point2l&, Point2l&) — returns (inside, p1, p2) with the clipped end points."""
    x1, y1 = p1
    x2, y2 = p2
    right, bottom = width - 1, height - 1
    if width <= 0 or height <= 0:
        return False, (x1, y1), (x2, y2)
    c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8
    c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8
    if (c1 & c2) == 0 and (c1 | c2) != 0:
        if c1 & 12:
            a = 0 if c1 < 8 else bottom
            x1 += _ctrunc(float(a - y1) * (x2 - x1) / (y2 - y1))
            y1 = a
            c1 = (x1 < 0) + (x1 > right) * 2
        if c2 & 12:
            a = 0 if c2 < 8 else bottom
            x2 += _ctrunc(float(a - y2) * (x2 - x1) / (y2 - y1))
            y2 = a
            c2 = (x2 < 0) + (x2 > right) * 2
        if (c1 & c2) == 0 and (c1 | c2) != 0:
            if c1:
                a = 0 if c1 == 1 else right
                y1 += _ctrunc(float(a - x1) * (y2 - y1) / (x2 - x1))
                x1 = a
                c1 = 0
            if c2:
                a = 0 if c2 == 1 else right
                y2 += _ctrunc(float(a - x2) * (y2 - y1) / (x2 - x1))
                x2 = a
                c2 = 0
    return (c1 | c2) == 0, (x1, y1), (x2, y2)


def line_pixels(width: int, height: int, p1, p2):
    """Pixels of cv::Line(img, p1, p2, color, 8): LineIterator(connectivity 8, leftToRight = true)."""
    x1, y1 = p1
    x2, y2 = p2
    if not (0 <= x1 < width and 0 <= x2 < width and 0 <= y1 < height and 0 <= y2 < height):
        ok, (x1, y1), (x2, y2) = clip_line(width, height, (x1, y1), (x2, y2))
        if not ok:
            return []
    dx, dy = x2 - x1, y2 - y1
    delta_x = delta_y = 1
    px, py = x1, y1
    if dx < 0:  # leftToRight: start from the left end point
        dx, dy = -dx, -dy
        px, py = x2, y2
    if dy < 0:
        dy, delta_y = -dy, -1
    vert = dy > dx
    if vert:
        dx, dy = dy, dx
        delta_x, delta_y = delta_y, delta_x
    err = dx - (dy + dy)
    plus_delta, minus_delta = dx + dx, -(dy + dy)
    # non-vertical: every step moves x by delta_x; a "plus" step also moves y by delta_y (swapped when vert)
    out = []
    for _ in range(dx + 1):
        out.append((px, py))
        neg = err < 0
        err += minus_delta + (plus_delta if neg else 0)
        if vert:
            py += delta_x
            if neg:
                px += delta_y
        else:
            px += delta_x
            if neg:
                py += delta_y
    return out


def line_pixels_closed_form(width: int, height: int, p1, p2):
    """The same pixel set without the iterator: pixel i of the major axis sits at minor offset
    floor((2*i*dmin + dmaj - 1) / (2*dmaj)) (the err < 0 test rounds exact halves down).  This is the form the HIP
    kernel uses (one lane per pixel)."""
    x1, y1 = p1
    x2, y2 = p2
    if not (0 <= x1 < width and 0 <= x2 < width and 0 <= y1 < height and 0 <= y2 < height):
        ok, (x1, y1), (x2, y2) = clip_line(width, height, (x1, y1), (x2, y2))
        if not ok:
            return []
    if x2 < x1:
        x1, y1, x2, y2 = x2, y2, x1, y1
    dx, dy = x2 - x1, y2 - y1
    sy = 1 if dy >= 0 else -1
    ady = abs(dy)
    out = []
    if ady > dx:  # y is the major axis
        for i in range(ady + 1):
            k = (2 * i * dx + ady - 1) // (2 * ady) if ady else 0
            out.append((x1 + k, y1 + sy * i))
    else:
        for i in range(dx + 1):
            k = (2 * i * ady + dx - 1) // (2 * dx) if dx else 0
            out.append((x1 + i, y1 + sy * k))
    return out


def collect_poly_edges(width: int, height: int, pts):
    """CollectPolyEdges for line_type = 8, shift = 0, offset = 0: returns (outline pixels, edge list) with
    edges as dicts {y0, y1, x, dx} (x, dx in 16.16 fixed point)."""
    pts = [(int(x), int(y)) for x, y in pts]
    outline, edges = [], []
    count = len(pts)
    if count == 0:
        return outline, edges
    pt0 = (pts[-1][0] << XY_SHIFT, pts[-1][1])
    for i in range(count):
        pt1 = (pts[i][0] << XY_SHIFT, pts[i][1])
        t0 = ((pt0[0] + (XY_ONE >> 1)) >> XY_SHIFT, pt0[1])
        t1 = ((pt1[0] + (XY_ONE >> 1)) >> XY_SHIFT, pt1[1])
        outline += line_pixels(width, height, t0, t1)
        pt0c, pt1c = list(pt0), list(pt1)
        if not (0 <= t0[0] < width and 0 <= t1[0] < width and 0 <= t0[1] < height and 0 <= t1[1] < height):
            _, c0, c1 = clip_line(width, height, t0, t1)
            if c0[1] != c1[1]:
                pt0c = [c0[0] << XY_SHIFT, c0[1]]
                pt1c = [c1[0] << XY_SHIFT, c1[1]]
        else:
            pt0c[0] += XY_ONE >> 1
            pt1c[0] += XY_ONE >> 1
        if pt0[1] != pt1[1]:
            dx = _cdiv(pt1c[0] - pt0c[0], pt1c[1] - pt0c[1])
            if pt0[1] < pt1[1]:
                e = dict(y0=pt0[1], y1=pt1[1], x=pt0c[0] + (pt0[1] - pt0c[1]) * dx, dx=dx)
            else:
                e = dict(y0=pt1[1], y1=pt0[1], x=pt1c[0] + (pt1[1] - pt1c[1]) * dx, dx=dx)
            edges.append(e)
        pt0 = pt1
    return outline, edges


def fill_edge_collection(mask: np.ndarray, edges) -> None:
    """FillEdgeCollection for line_type < LINE_AA (delta = 0).  The active-edge list of OpenCV is kept sorted by x
    (merge insertion + bubble sort), so each scanline pairs the sorted crossings; ties do not change the spans."""
    height, width = mask.shape
    if len(edges) < 2:
        return
    y_min = min(e["y0"] for e in edges)
    y_max = max(e["y1"] for e in edges)
    xs = [e["x"] for e in edges] + [e["x"] + (e["y1"] - e["y0"]) * e["dx"] for e in edges]
    if y_max < 0 or y_min >= height or max(xs) < 0 or min(xs) >= (width << XY_SHIFT):
        return
    y_max = min(y_max, height)
    for y in range(y_min, y_max):
        cross = sorted(e["x"] + (y - e["y0"]) * e["dx"] for e in edges if e["y0"] <= y < e["y1"])
        if y < 0:
            continue
        for a, b in zip(cross[0::2], cross[1::2]):
            x1, x2 = a >> XY_SHIFT, b >> XY_SHIFT
            if x1 < width and x2 >= 0:
                mask[y, max(x1, 0):min(x2, width - 1) + 1] = 1


def fill_poly(mask: np.ndarray, pts) -> np.ndarray:
    """cv2.fillPoly(mask, [pts], color=1) on a uint8 (H, W) array, in place."""
    height, width = mask.shape
    outline, edges = collect_poly_edges(width, height, pts)
    for x, y in outline:
        mask[y, x] = 1
    fill_edge_collection(mask, edges)
    return mask


def create_boolean_mask_from_polygon(image_shape, segmentation):
    """/root/reference/src/util.py:386-415, polygon branch: image_shape = (width, height); every part is truncated
    to int32 and filled on its own.  Returns (bool mask (H, W), get_maximum_height(mask))."""
    mask = np.zeros((image_shape[1], image_shape[0]), dtype=np.uint8)
    for polygon in segmentation:
        points = np.array(polygon).reshape(-1, 2).astype(np.int32)
        fill_poly(mask, points)
    m = mask.astype(bool)
    rows = np.where(np.any(m, axis=1))[0]          # get_maximum_height, src/util.py:328-335
    return m, (0 if rows.size == 0 else int(rows[-1] - rows[0] + 1))
