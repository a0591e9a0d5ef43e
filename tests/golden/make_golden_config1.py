"""Generate tests/golden/g14_config1.npz and g15_near_ties.npz by RUNNING THE REFERENCE in the build container.

    python tests/golden/make_golden_config1.py

G14 = BASELINE config 1: ONE 640x480 image - depth RandomState(0).uniform(0.5, 10) f32, K = [[500,0,320],[0,500,240],
[0,0,1]] - and one instance mask, through the literal composition of the two reference functions
    estimate_bbox(depth_to_points(depth[None], K)[mask], None, ground)          (src/util.py:52-75, src/util_3dbox.py:106-178)
for a mask above 500 px (the reference subsamples: seeded global RNG, the drawn indices are recorded), one of <= 500 px, an
irregular one, each with and without a ground vector.  Only seeds, mask parameters and the reference's outputs are stored.

G15 = yaw near ties: clouds whose footprint covariance has a relative eigen-gap of 1e-7 .. 1e-2 (n >= 20: scikit-learn's
covariance_eigh branch), so that the GPU tests' "skip R_cam below gap 1e-6" gate is itself checked against reference output.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refimport  # noqa: E402
from make_golden import K640, rec39, run_bbox  # noqa: E402

H, W = 480, 640


def config1_masks():
    vv, uu = np.mgrid[0:H, 0:W]
    m = np.zeros((3, H, W), bool)
    m[0, 150:150 + 154, 200:200 + 169] = True                            # 26 026 px: subsample branch
    m[1, 300:300 + 20, 411:411 + 24] = True                              # 480 px: full-mask branch
    m[2] = ((uu - 250.5) ** 2 / 90.0 ** 2 + (vv - 230.2) ** 2 / 55.0 ** 2) < 1.0   # ellipse, 15.5k px
    return m


def near_tie_cloud(rs, n, gap, yaw, y_scale=0.3):
    """n points whose (x, z) sample covariance is EXACTLY diag-like with eigenvalues (1, 1 - gap) rotated by yaw: whiten a random
    sample, scale, rotate.  The realised gap differs from the request only by rounding."""
    p = rs.randn(n, 2)
    p -= p.mean(0)
    c = p.T @ p / (n - 1)
    w, v = np.linalg.eigh(c)
    p = p @ v / np.sqrt(w)                       # unit sample covariance
    p = p * np.sqrt([1.0, 1.0 - gap])
    cy, sy = np.cos(yaw), np.sin(yaw)
    xz = p @ np.array([[cy, sy], [-sy, cy]])     # major axis along (cos yaw, sin yaw)
    out = np.empty((n, 3))
    out[:, 0] = xz[:, 0] + 0.4
    out[:, 1] = y_scale * rs.randn(n) - 0.2
    out[:, 2] = xz[:, 1] + 6.0
    return out


def main():
    ref = _refimport.load()
    if ref is None:
        print("reference not present; nothing generated")
        return 0
    depth = np.random.RandomState(0).uniform(0.5, 10, (H, W)).astype(np.float32)
    masks = config1_masks()
    ground = np.array([0.05, -0.97, 0.1, 1.2])
    pts = ref.util.depth_to_points(depth[None], K640)
    assert pts.shape == (H, W, 3) and pts.dtype == np.float64
    out, idx, seeds = [], [], []
    for i in range(3):
        n = int(masks[i].sum())
        for j, g in enumerate((None, ground)):
            seed = 4100 + 10 * i + j
            np.random.seed(seed)
            idx.append(np.random.randint(0, n, 500) if n > 500 else np.zeros(500, np.int64))
            np.random.seed(seed)
            r, en, _ = run_bbox(ref, pts[masks[i]], g)
            assert en == "", (i, j, en)
            out.append(r)
            seeds.append(seed)
    np.savez_compressed(os.path.join(HERE, "g14_config1.npz"), depth_seed=np.array([0]), K=K640, ground=ground,
                        n_masked=masks.reshape(3, -1).sum(1), mask_bits=np.packbits(masks.reshape(3, -1), axis=1),
                        out=np.array(out).reshape(3, 2, 39), sample_idx=np.array(idx).reshape(3, 2, 500),
                        rng_seed=np.array(seeds).reshape(3, 2),
                        pts_pick=np.array([0, 12345, 153600, 307199]), pts_out=pts.reshape(-1, 3)[[0, 12345, 153600, 307199]])

    rs = np.random.RandomState(15)
    clouds, outs, gaps, yaws = [], [], [], []
    for gap in (1e-7, 3e-7, 1e-6, 3e-6, 1e-5, 1e-4, 1e-3, 1e-2):
        for yaw in (0.2, 1.1, -0.6):
            pc = near_tie_cloud(rs, 240, gap, yaw)
            r, en, _ = run_bbox(ref, pc)
            assert en == ""
            clouds.append(pc); outs.append(r); gaps.append(gap); yaws.append(yaw)
    np.savez_compressed(os.path.join(HERE, "g15_near_ties.npz"), pcs=np.array(clouds), out=np.array(outs), gap=np.array(gaps),
                        yaw=np.array(yaws))
    print("wrote g14_config1.npz, g15_near_ties.npz")
    return 0


if __name__ == "__main__":
    sys.exit(main())
