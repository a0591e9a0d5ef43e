"""Round 6: the r03 stress of the convex-hull yaw with clouds up to the new cap of 2048 points: lattice clouds (many duplicate and collinear points), tiny clouds,
clouds on a circle (every point a hull vertex), random clouds - against the oracle; ties between equally minimal edges are accepted
(same rectangle: centre, dy, {dx, dz} as a set)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import labelany3d_amd as la
from oracle import la3d_oracle as O
rs = np.random.RandomState(77)
clouds = []
for k in range(400):
    kind = k % 5
    n = int(rs.choice([3, 5, 17, 64, 100, 257, 500, 512, 513, 700, 1024, 1025, 1500, 2047, 2048]))
    if kind == 0:      # integer lattice: duplicates + collinear runs
        pc = np.stack([rs.randint(0, 6, n), rs.rand(n), rs.randint(0, 5, n)], 1).astype(float)
    elif kind == 1:    # circle: all vertices
        a = np.sort(rs.uniform(0, 2 * np.pi, n)); pc = np.stack([2 * np.cos(a), rs.rand(n), 1.5 * np.sin(a)], 1)
    elif kind == 2:    # random box rotated
        pc = (rs.rand(n, 3) * [3, 1, 1]) @ O.rotate_y(rs.uniform(-3, 3)).T
    elif kind == 3:    # gaussian blob
        pc = rs.randn(n, 3) * [2, 0.3, 0.7]
    else:              # few distinct x values (vertical stacks in the sorted order)
        pc = np.stack([rs.randint(0, 3, n) * 1.0, rs.rand(n), rs.rand(n)], 1)
    clouds.append(pc + [0, 0, 6])
boxes, status, aux = la.fit_points(clouds, None, None, "convex_hull")
boxes, status, aux = boxes.cpu().numpy(), status.cpu().numpy(), aux.cpu().numpy()
bad = 0
for i, c in enumerate(clouds):
    rec, st, a = O.fit_points(c, None, False, "convex_hull")
    assert st == status[i], (i, st, status[i])
    if st:
        continue
    ok = np.allclose(boxes[i, [0, 1, 2, 4]], rec[[0, 1, 2, 4]], rtol=0, atol=1e-9) and np.allclose(sorted(boxes[i, [3, 5]]), sorted(rec[[3, 5]]), rtol=0, atol=1e-9)
    if not ok:   # a different edge of (nearly) equal area: compare the areas
        ar_g, ar_r = boxes[i, 3] * boxes[i, 5], rec[3] * rec[5]
        if abs(ar_g - ar_r) > 1e-9 * max(1, ar_r):
            bad += 1; print("MISMATCH", i, len(c), boxes[i, :6], rec[:6])
print(f"{len(clouds)} clouds, {int((status == 0).sum())} fitted, {bad} mismatches")
assert bad == 0
