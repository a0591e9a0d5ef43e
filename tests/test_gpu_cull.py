"""Pass-B tile culling of the instance engine's plain build (round 4): the extents are min / max
over the points, so skipping tiles that provably cannot move them must leave the records BIT-identical.  Checked two ways on
480x640 frames whose masks have enough active tiles for the culling plan to run: against the CPU oracle (the stated 1e-9), and
bit for bit against the no-cull build of the same kernel (opt_build = LA3D_BUILD_NOCULL), which has no culling plan and walks every tile."""
import numpy as np
import pytest

from .conftest import SCHED

from oracle import la3d_oracle as O

from .test_gpu_parity import K640, assert_records, np_

pytestmark = pytest.mark.gpu

H, W = 480, 640


@pytest.fixture(scope="module")
def la():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import labelany3d_amd

    return labelany3d_amd


def _shapes(rs, B):
    """rectangles, ellipses, rings and a mask with a hole band - all large enough for the plan (>= 96 active tiles mostly)"""
    vv, uu = np.mgrid[0:H, 0:W]
    masks = np.zeros((B, H, W), bool)
    for i in range(B):
        h, w = rs.randint(120, 420), rs.randint(160, 600)
        r0, c0 = rs.randint(0, H - h + 1), rs.randint(0, W - w + 1)
        kind = i % 4
        e = ((vv + 0.5 - r0 - h / 2) / (h / 2)) ** 2 + ((uu + 0.5 - c0 - w / 2) / (w / 2)) ** 2
        if kind == 0:
            masks[i, r0:r0 + h, c0:c0 + w] = True
        elif kind == 1:
            masks[i] = e <= 1.0
        elif kind == 2:
            masks[i] = (e <= 1.0) & (e >= 0.45)
        else:
            masks[i, r0:r0 + h, c0:c0 + w] = True
            masks[i, r0 + h // 3:r0 + h // 2, :] = False
    return masks


def _depths(rs, B, kind):
    vv, uu = np.mgrid[0:H, 0:W]
    if kind == "random":
        return rs.uniform(0.5, 10, (B, H, W)).astype(np.float32)
    if kind == "sphere":   # nearest point in the interior of the mask
        d = np.empty((B, H, W), np.float32)
        for i in range(B):
            cy, cx = rs.uniform(100, 380), rs.uniform(150, 490)
            rr = np.clip(1 - ((vv - cy) / 260.0) ** 2 - ((uu - cx) / 340.0) ** 2, 0, 1)
            d[i] = (6.0 - 2.5 * np.sqrt(rr) + 0.002 * rs.randn(H, W)).astype(np.float32)
        return d
    if kind == "plane":
        return (3.0 + 0.004 * uu[None] + 0.002 * vv[None] + 0.003 * rs.randn(B, H, W)).astype(np.float32)
    if kind == "constant":
        return np.full((B, H, W), 2.5, np.float32)
    if kind == "signed":   # negative and zero depths under the mask: tiles holding them are never culled
        d = rs.uniform(0.5, 10, (B, H, W)).astype(np.float32)
        d[:, 100:140, 200:330] *= -1.0
        d[:, 300:310, :] = 0.0
        return d
    if kind == "tiny":     # huge dynamic range: the slack must scale with the tile, not the frame
        return (10.0 ** rs.uniform(-6, 3, (B, H, W))).astype(np.float32)
    raise ValueError(kind)


def _fit(la, monkeypatch, depth, masks, K, ground, retain):
    monkeypatch.setattr(SCHED(), "engine", "instance")
    monkeypatch.setattr(SCHED(), "build", {"0": "plain", "1": "nocull"}[retain])
    b, s, a = la.fit_instances(depth, masks, K, ground=ground)
    return np_(b), np_(s), np_(a)


@pytest.mark.parametrize("kind", ["random", "sphere", "plane", "constant", "signed", "tiny"])
def test_culled_pass_b_is_exact(la, monkeypatch, kind):
    rs = np.random.RandomState({"random": 1, "sphere": 2, "plane": 3, "constant": 4, "signed": 5, "tiny": 6}[kind])
    B = 12
    masks = _shapes(rs, B)
    depth = _depths(rs, B, kind)
    ground = np.array([[0.05, -0.97, 0.1, 1.2]] * B) + 0.03 * rs.randn(B, 4)
    ground[::3, 0] = np.nan   # every third instance without a ground plane
    K = K640 + np.array([[0, 0.7, 0], [0, 0, 0], [0, 0, 0]]) if kind == "plane" else K640
    got = _fit(la, monkeypatch, depth, masks, K, ground, "0")      # plain build: culling on
    full = _fit(la, monkeypatch, depth, masks, K, ground, "1")     # no-cull build: every tile walked
    assert (got[1] == full[1]).all()
    np.testing.assert_array_equal(got[0], full[0], err_msg=f"{kind}: culled pass B changed a record")
    np.testing.assert_array_equal(got[2], full[2])
    ground_o = [None if np.isnan(g[0]) else g for g in ground]   # (the batched API reads a NaN first entry as "no ground")
    ref, rst, _, _ = O.fit_instances(depth, masks, np.broadcast_to(K, (B, 3, 3)), ground=ground_o)
    assert got[1].tolist() == list(rst)
    ok = got[1] == 0
    if kind != "constant":   # (constant depth: exact eigen-ties are the documented don't-care for R_cam)
        assert_records(got[0][ok], ref[ok], f"cull/{kind}", gap=got[2][ok, 3])


def test_culling_with_nonfinite_depth(la, monkeypatch):
    """inf / NaN under the mask: the optimistic pass A is re-run in its checked form, the depth ranges with it; a tile whose
    only valid pixels are gone must not poison the bounds."""
    rs = np.random.RandomState(11)
    B = 8
    masks = _shapes(rs, B)
    depth = rs.uniform(0.5, 10, (B, H, W)).astype(np.float32)
    depth[0, 200:208, 320:352] = np.inf      # one whole tile without a valid pixel
    depth[1, 100:300, 100:500:7] = np.nan
    depth[2, 240, 320] = -np.inf
    depth[3, :, :] = np.where(rs.rand(H, W) < 0.5, np.nan, depth[3])
    got = _fit(la, monkeypatch, depth, masks, K640, None, "0")
    full = _fit(la, monkeypatch, depth, masks, K640, None, "1")
    np.testing.assert_array_equal(got[0], full[0])
    ref, rst, _, _ = O.fit_instances(depth, masks, np.broadcast_to(K640, (B, 3, 3)))
    assert got[1].tolist() == list(rst)
    assert_records(got[0], ref, "cull/nonfinite", gap=got[2][:, 3])


def test_culling_rle_and_polygon_input_match_planes(la, monkeypatch):
    """run-length and polygon input take the plain build by default: same records as the u8 planes, bit for bit"""
    from labelany3d_amd.masks import fit_instances_rle

    rs = np.random.RandomState(5)
    B = 10
    masks = _shapes(rs, B)
    depth = _depths(rs, B, "sphere")
    monkeypatch.setattr(SCHED(), "engine", "instance")
    monkeypatch.setattr(SCHED(), "build", "plain")
    b0, s0, _ = la.fit_instances(depth, masks, K640)
    b1, s1, _ = fit_instances_rle(depth, [O.rle_encode(m) for m in masks], K640)
    np.testing.assert_array_equal(np_(b0), np_(b1))
    assert np_(s0).tolist() == np_(s1).tolist()
    ref, rst, _, _ = O.fit_instances(depth, masks, np.broadcast_to(K640, (B, 3, 3)))
    assert_records(np_(b1), ref, "cull/rle")


# ------------------------------------------------------------------------------------------
# Round 4, late: the speed knobs of the plain build - the staggered start of the resident groups, the culling threshold, the
# no-cull build - are read once per process (config()), so each setting runs in its own interpreter; records must not move by a bit.
# ------------------------------------------------------------------------------------------
_KNOB_SCRIPT = r"""
import hashlib, sys
import numpy as np, torch
sys.path.insert(0, %r)
from labelany3d_amd import InstanceFitter
B, H, W = 700, 480, 640
rs = np.random.RandomState(77)
dev = torch.device("cuda", 0)
depth = torch.as_tensor(rs.uniform(0.5, 10, (8, H, W)).astype(np.float32), device=dev)
ii = torch.as_tensor(rs.randint(0, 8, B).astype(np.int32), device=dev)
m = np.zeros((B, H, W), np.uint8)
for i in range(B):
    h, w = rs.randint(1, 400), rs.randint(1, 500)
    r0, c0 = rs.randint(0, H - h + 1), rs.randint(0, W - w + 1)
    m[i, r0:r0 + h, c0:c0 + w] = 1
masks = torch.as_tensor(m, device=dev)
K = torch.tensor([[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]], dtype=torch.float64, device=dev)
f = InstanceFitter(B, H, W, dev)
f.boxes.fill_(12345.0); f.status.fill_(-1)
b, s, a = f.run(depth, masks, K, image_index=ii)
torch.cuda.synchronize()
assert int((s != 0).sum()) == 0, s
print("SHA", hashlib.sha1(b.cpu().numpy().tobytes() + s.cpu().numpy().tobytes() + a.cpu().numpy().tobytes()).hexdigest())
np.save(sys.argv[1], b.cpu().numpy())
"""


def test_speed_knobs_leave_the_records_alone(la, tmp_path):
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shas, recs = {}, {}
    two = {"LA3D_SEP": "0"}   # the two-pass plain build (round 5: un-grounded calls take the separable single pass by default)
    for name, env in (("default", {}), ("no stagger", {"LA3D_STAGGER_US": "0"}), ("long stagger", {"LA3D_STAGGER_US": "23"}),
                      ("no launch order", {"LA3D_BALANCE": "0"}),
                      # the self-estimating launch (no helper kernel): off, and with every seventh workgroup withholding its key,
                      # so that the waiting workgroups time out and compute the missing keys themselves
                      ("helper kernel", {"LA3D_ORDER_SELF": "0"}), ("self-estimate fallback", {"LA3D_ORDER_SELF": "2"}),
                      ("two-pass", two), ("two-pass, cull everything", dict(two, LA3D_CULL_MIN="1")),
                      ("two-pass, cull nothing", dict(two, LA3D_CULL_MIN="100000")), ("two-pass, no stagger", dict(two, LA3D_STAGGER_US="0")),
                      ("no-cull build", {"LA3D_BUILD": "nocull"})):
        e = dict(os.environ, LA3D_ENGINE="instance", **env)
        out = str(tmp_path / (name.replace(" ", "_").replace(",", "") + ".npy"))
        r = subprocess.run([sys.executable, "-c", _KNOB_SCRIPT % root, out], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (name, r.stderr[-2000:])
        shas[name] = [ln for ln in r.stdout.splitlines() if ln.startswith("SHA")][0]
        recs[name] = np.load(out)
    # the single pass groups its sums alike whatever the knobs; so does the two-pass plain build, and the no-cull build groups them
    # like the two-pass plain one as long as the active tiles fit the plain build's list (rectangles below 400 x 500 px do)
    single = {k: v for k, v in shas.items() if not k.startswith("two-pass") and k != "no-cull build"}
    double = {k: v for k, v in shas.items() if k.startswith("two-pass") or k == "no-cull build"}
    assert len(set(single.values())) == 1, single
    assert len(set(double.values())) == 1, double
    np.testing.assert_allclose(recs["default"][:, :15], recs["two-pass"][:, :15], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(recs["default"][:, 15:], recs["two-pass"][:, 15:], rtol=0, atol=2e-2)   # fp16-quantised corners


@pytest.mark.parametrize("B", [1024, 1500, 2600])
def test_self_estimating_launch_alternating_batches(la, monkeypatch, B):
    """The ordered launch of up to one resident set estimates its sort keys inside the fit kernel and hands them over through the
    workspace (round 4).  Alternate two batches whose instance sizes differ slot by slot on ONE workspace, many times: a key that
    arrived from the previous call (or one half of a record) would rank a workgroup differently from its neighbours - a skipped
    and a duplicated instance.  Every record of every call must equal the unordered launch's, and every instance must be written."""
    import torch

    from labelany3d_amd import InstanceFitter

    monkeypatch.setattr(SCHED(), "engine", "instance")
    dev = torch.device("cuda", 0)
    H, W = 96, 128   # (above 1024 instances the workgroups of the first resident set estimate for everybody)
    K = torch.tensor([[100.0, 0, 64], [0, 100.0, 48], [0, 0, 1]], dtype=torch.float64, device=dev)
    sets = []
    for seed in (1, 2):
        rs = np.random.RandomState(seed)
        depth = torch.as_tensor(rs.uniform(0.5, 10, (B, H, W)).astype(np.float32), device=dev)
        m = np.zeros((B, H, W), np.uint8)
        for i in range(B):
            h, w = rs.randint(1, H + 1), rs.randint(1, W + 1)
            r0, c0 = rs.randint(0, H - h + 1), rs.randint(0, W - w + 1)
            m[i, r0:r0 + h, c0:c0 + w] = 1
        sets.append((depth, torch.as_tensor(m, device=dev)))
    f = InstanceFitter(B, H, W, dev)
    f.workspace.zero_()
    ref = []
    monkeypatch.setattr(SCHED(), "launch_order", False)
    for depth, masks in sets:
        b, s, a = f.run(depth, masks, K)
        torch.cuda.synchronize()
        ref.append((b.clone(), s.clone(), a.clone()))
    monkeypatch.setattr(SCHED(), "launch_order", None)
    ref = [(torch.nan_to_num(b, nan=-7.0), s, torch.nan_to_num(a, nan=-7.0)) for b, s, a in ref]
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    for it in range(400 if B == 1024 else 150):
        depth, masks = sets[it & 1]
        if it % 3 == 0:   # (back-to-back calls and calls with a little stream work between them)
            f.boxes.fill_(12345.0); f.status.fill_(-1); f.aux.fill_(12345.0)
        b, s, a = f.run(depth, masks, K)
        rb, rs_, ra = ref[it & 1]   # every call is checked, on the stream: no host synchronisation between the calls
        bad += (s != rs_).sum() + (torch.nan_to_num(b, nan=-7.0) != rb).sum() + (torch.nan_to_num(a, nan=-7.0) != ra).sum()
    assert int(bad) == 0
    # the fallback (a workgroup computing a neighbour's key because it did not arrive in time) must not have been needed: its counter
    # sits behind the B two-word records, which follow the 256-byte aligned key table
    off = ((B * 4 + 255) // 256) * 256 + 16 * B
    assert int(f.workspace[0][off:off + 8].view(torch.int64)[0]) == 0


@pytest.mark.parametrize("engine,B", [("instance", 1024), ("band", 64), ("rows", 24)])
def test_graph_replay_with_new_masks_in_the_same_buffers(la, monkeypatch, engine, B):
    """An ordered call captured into a HIP graph and replayed over NEW masks in the same buffers: the captured call must not carry the
    self-estimating launch's per-call nonce (a replay would find the previous replay's records complete): it keeps the helper kernel.
    Every replay must equal the eager, unordered launch on the same data."""
    import torch

    from labelany3d_amd import InstanceFitter

    # (the band engine's arrival words carry a per-call tag too: a captured call clears them with a memset node instead; the row
    # engine's two launches carry nothing per call)
    monkeypatch.setattr(SCHED(), "engine", engine)
    dev = torch.device("cuda", 0)
    H, W = 96, 128
    K = torch.tensor([[100.0, 0, 64], [0, 100.0, 48], [0, 0, 1]], dtype=torch.float64, device=dev)
    rs = np.random.RandomState(5)
    depth = torch.as_tensor(rs.uniform(0.5, 10, (B, H, W)).astype(np.float32), device=dev)

    def new_masks(seed):
        r = np.random.RandomState(seed)
        m = np.zeros((B, H, W), np.uint8)
        for i in range(B):
            h, w = r.randint(1, H + 1), r.randint(1, W + 1)
            r0, c0 = r.randint(0, H - h + 1), r.randint(0, W - w + 1)
            m[i, r0:r0 + h, c0:c0 + w] = 1
        return torch.as_tensor(m, device=dev)

    masks = new_masks(0)
    f = InstanceFitter(B, H, W, dev)
    side = torch.cuda.Stream(device=dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        f.run(depth, masks, K, stream=side)
        side.synchronize()
        with torch.cuda.graph(g, stream=side):
            f.run(depth, masks, K, stream=torch.cuda.current_stream())
    fr = InstanceFitter(B, H, W, dev)
    for seed in range(1, 9):
        masks.copy_(new_masks(seed))
        f.boxes.fill_(12345.0); f.status.fill_(-1)
        g.replay()
        torch.cuda.synchronize()
        got_b, got_s = f.boxes[0].clone(), f.status[0].clone()
        monkeypatch.setattr(SCHED(), "launch_order", False)
        rb, rs_, _ = fr.run(depth, masks, K)
        monkeypatch.setattr(SCHED(), "launch_order", None)
        torch.cuda.synchronize()
        assert torch.equal(got_s, rs_), seed
        assert torch.equal(torch.nan_to_num(got_b, nan=-7.0), torch.nan_to_num(rb, nan=-7.0)), seed
    if engine == "band":   # and the eager band call after the replays (its own tag, the words as the last replay left them)
        f.boxes.fill_(12345.0)
        b, s, _ = f.run(depth, masks, K)
        torch.cuda.synchronize()
        assert torch.equal(s, rs_) and torch.equal(torch.nan_to_num(b, nan=-7.0), torch.nan_to_num(rb, nan=-7.0))


def test_two_ordered_calls_running_concurrently(la, monkeypatch):
    """Two ORDERED 1024-instance calls on two streams at once, each with its own workspace: neither launch has all its workgroups
    resident, so workgroups wait for keys whose producers are not on the chip yet - the self-estimating launch must fall back to
    computing them itself (slow, ~0.25 ms, but never a hang) and every record must still be right."""
    import torch

    from labelany3d_amd import InstanceFitter

    monkeypatch.setattr(SCHED(), "engine", "instance")
    dev = torch.device("cuda", 0)
    B, H, W = 1024, 480, 640
    K = torch.tensor([[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]], dtype=torch.float64, device=dev)
    data = []
    for seed in (11, 12):
        rs = np.random.RandomState(seed)
        depth = torch.as_tensor(rs.uniform(0.5, 10, (4, H, W)).astype(np.float32), device=dev)
        ii = torch.as_tensor(rs.randint(0, 4, B).astype(np.int32), device=dev)
        m = np.zeros((B, H, W), np.uint8)
        for i in range(B):
            h, w = rs.randint(8, 301), rs.randint(8, 331)
            r0, c0 = rs.randint(0, H - h + 1), rs.randint(0, W - w + 1)
            m[i, r0:r0 + h, c0:c0 + w] = 1
        data.append((depth, torch.as_tensor(m, device=dev), ii))
    fit = [InstanceFitter(B, H, W, dev) for _ in data]
    streams = [torch.cuda.Stream(device=dev) for _ in data]
    ref = []
    monkeypatch.setattr(SCHED(), "launch_order", False)
    for f, (depth, masks, ii) in zip(fit, data):
        b, s, _ = f.run(depth, masks, K, image_index=ii)
        torch.cuda.synchronize()
        ref.append((torch.nan_to_num(b, nan=-7.0).clone(), s.clone()))
    monkeypatch.setattr(SCHED(), "launch_order", None)
    for rep in range(20):
        for f in fit:
            f.boxes.fill_(12345.0); f.status.fill_(-1)
        torch.cuda.synchronize()
        for f, st, (depth, masks, ii) in zip(fit, streams, data):
            f.run(depth, masks, K, image_index=ii, stream=st)
        torch.cuda.synchronize()
        for f, (rb, rs_) in zip(fit, ref):
            assert torch.equal(f.status[0], rs_), rep
            assert torch.equal(torch.nan_to_num(f.boxes[0], nan=-7.0), rb), rep
