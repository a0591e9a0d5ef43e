"""The C-ABI from a plain host program (examples/fit_from_c.cpp: hipMalloc'd buffers, no Python, no torch): its printed
records are checked against the oracle on inputs regenerated here with the same LCG."""
import os
import subprocess

import numpy as np
import pytest

from oracle import la3d_oracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "labelany3d_amd", "lib", "fit_from_c")


class LCG:
    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFF

    def next(self):
        self.s = (self.s * 1664525 + 1013904223) & 0xFFFFFFFF
        return self.s >> 8

    def unit(self):
        return self.next() / 16777216.0


def make_inputs(B, H, W, seed):
    g = LCG(seed)
    depth = np.empty((B, H, W), np.float32)
    masks = np.zeros((B, H, W), bool)
    ground = np.empty((B, 4))
    for i in range(B):
        depth[i] = np.array([np.float32(0.5 + 9.5 * g.unit()) for _ in range(H * W)], np.float32).reshape(H, W)
        h, w = 1 + g.next() % H, 1 + g.next() % W
        r0, c0 = g.next() % (H - h + 1), g.next() % (W - w + 1)
        masks[i, r0:r0 + h, c0:c0 + w] = True
        ground[i] = [0.02 + 0.1 * (g.unit() - 0.5), -0.98 + 0.1 * (g.unit() - 0.5), 0.1 + 0.1 * (g.unit() - 0.5), 1.5]
    K = np.array([[0.8 * W, 0, 0.5 * W], [0, 0.8 * W, 0.5 * H], [0, 0, 1]])
    return depth, masks, K, ground


@pytest.mark.parametrize("B,H,W,seed", [(5, 48, 64, 7), (12, 128, 256, 11), (3, 37, 53, 3)])
def test_plain_host_program_matches_oracle(B, H, W, seed):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    assert os.path.exists(BIN), "run __graft_entry__.build() first (labelany3d_amd/lib/fit_from_c)"
    out = subprocess.run([BIN, str(B), str(H), str(W), str(seed)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    rows = [l.split() for l in lines if not l.startswith("P")]
    prows = [l.split()[1:] for l in lines if l.startswith("P")]     # la3d_fit_instances_ex: bbox2D_proj | bbox2D_trunc per record
    assert len(rows) == B and len(prows) == B
    status = np.array([int(r[0]) for r in rows])
    rec = np.array([[float.fromhex(x) if x not in ("nan", "-nan") else np.nan for x in r[1:]] for r in rows])
    depth, masks, K, ground = make_inputs(B, H, W, seed)
    ref = [O.fit_instance(depth[i], masks[i], K, ground[i]) for i in range(B)]
    assert status.tolist() == [r[1] for r in ref]
    for i, (r, st, aux) in enumerate(ref):
        if st:
            assert np.isnan(rec[i]).all()
            continue
        scale = max(1.0, np.abs(r[:6]).max())
        np.testing.assert_allclose(rec[i, :15], r[:15], rtol=0, atol=1e-9 * scale, err_msg=f"instance {i}")
        np.testing.assert_allclose(rec[i, 15:], r[15:], rtol=0, atol=max(np.abs(r[15:]).max(), 1.0) * 2.0 ** -10)
    # the 2-D boxes against the reference's arithmetic on the oracle's corners (src/tools/combine_results.py:105-108, :238-252)
    p2 = np.array([[float.fromhex(x) if x not in ("nan", "-nan") else np.nan for x in r] for r in prows])
    for i, (r, st, aux) in enumerate(ref):
        if st:
            assert np.isnan(p2[i]).all()
            continue
        want = O.project_boxes(rec[i][None], K, (W, H))[0]
        np.testing.assert_allclose(p2[i], want, rtol=1e-12, atol=1e-9)
