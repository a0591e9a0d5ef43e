"""BASELINE configs 4 and 5 on ONE GPU.

config 4 (per-image sharding + one gather of box records): the sharded driver with the REAL fit, two processes on the
same device joined by a gloo group (RCCL refuses two ranks on one GPU; the nccl path itself runs in bench.py --gpus N).
Records must arrive on rank 0 in global instance order and agree with the oracle.

config 5 (mask areas log-uniform 8..100k px, private depth): both engines, batches below and above the three resident
sets the size-balanced launch order covers."""
import os
import socket

import numpy as np
import pytest

from .conftest import SCHED

from oracle import la3d_oracle as O

pytestmark = pytest.mark.gpu

H, W = 480, 640
K640 = np.array([[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene(seed=11, P=9):
    rs = np.random.RandomState(seed)
    depth = rs.uniform(0.5, 10, (P, H, W)).astype(np.float32)
    per = rs.randint(0, 6, P)
    per[3] = 0
    img = np.repeat(np.arange(P), per).astype(np.int32)
    B = len(img)
    masks = np.zeros((B, H, W), bool)
    for i in range(B):
        h, w = rs.randint(4, 200), rs.randint(4, 260)
        r0, c0 = rs.randint(0, H - h + 1), rs.randint(0, W - w + 1)
        masks[i, r0:r0 + h, c0:c0 + w] = True
    Ks = np.repeat(K640[None], P, 0) * (1 + 0.01 * np.arange(P))[:, None, None]
    Ks[:, 2, 2] = 1
    ground = np.array([[0.03, -0.98, 0.08, 1.4]] * B) + 0.02 * rs.randn(B, 4)
    return depth, masks, Ks, img, ground


def _worker(rank, world, port, q):
    import traceback

    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from labelany3d_amd.shard import fit_instances_sharded, plan_shards

        torch.cuda.set_device(0)
        depth, masks, Ks, img, ground = _scene()
        areas = masks.reshape(len(img), -1).sum(1)
        shard = plan_shards(img, depth.shape[0], world, areas=areas, frame_pixels=H * W)[rank]

        def load(sh):   # this rank uploads ONLY its own images / instances
            assert sh == shard
            dev = torch.device("cuda", 0)
            return (torch.as_tensor(depth[sh.img_lo:sh.img_hi], device=dev), torch.as_tensor(masks[sh.inst_lo:sh.inst_hi], device=dev),
                    torch.as_tensor(Ks[sh.img_lo:sh.img_hi], device=dev), ground[sh.inst_lo:sh.inst_hi], None)

        out = fit_instances_sharded(depth.shape, None, None, img, areas=areas, load_fn=load)
        # and the global-tensor form (views of the same arrays)
        out2 = fit_instances_sharded(depth, masks, Ks, img, ground=ground, areas=areas)
        if rank == 0:
            b, s, counts = out
            q.put((rank, "ok", b.cpu().numpy(), s.cpu().numpy(), counts, out2[0].cpu().numpy()))
        else:
            assert out is None and out2 is None
            q.put((rank, "ok", shard.inst_hi - shard.inst_lo))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e) + traceback.format_exc()[-800:]))
        raise
    finally:
        dist.destroy_process_group()


def test_config4_two_ranks_one_gpu_real_fit():
    import torch
    import torch.multiprocessing as mp

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(120)
    assert res[0][1] == "ok" and res[1][1] == "ok", res
    _, _, boxes, status, counts, boxes2 = res[0]
    depth, masks, Ks, img, ground = _scene()
    assert sum(counts) == len(img) and counts[1] == res[1][2] and min(counts) > 0
    ref, rst, _, _ = O.fit_instances(depth, masks, Ks, ground=ground, depth_index=img)
    assert status.tolist() == rst.tolist()
    ok = rst == 0
    np.testing.assert_allclose(boxes[ok][:, :15], ref[ok][:, :15], rtol=0, atol=1e-9)
    np.testing.assert_allclose(boxes[ok][:, 15:], ref[ok][:, 15:], rtol=0, atol=2e-2)
    np.testing.assert_array_equal(boxes, boxes2)          # load_fn form == global-tensor form


@pytest.fixture(scope="module")
def la():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import labelany3d_amd

    return labelany3d_amd


@pytest.mark.parametrize("B", [1024, 4096, 16384])
def test_config5_size_distribution(la, B):
    """Areas log-uniform 8..100k px: the instance engine (with and without its launch order; 16384 is past the three
    resident sets the order covers), the split engine on a sub-batch, and the oracle on a sample."""
    import torch

    import bench

    dev = torch.device("cuda", 0)
    depth, masks, K, n_masked, _ = bench.make_config5(B, dev, 77)
    area = masks.reshape(B, -1).sum(1, dtype=torch.int64).cpu().numpy()
    assert area.min() >= 4 and area.max() <= 110000 and np.median(area) < 3000     # log-uniform: mostly small, a few huge
    SCHED().engine = "instance"
    try:
        b1, s1, a1 = la.fit_instances(depth, masks, K)
        SCHED().launch_order = False
        b0, s0, _ = la.fit_instances(depth, masks, K)
    finally:
        SCHED().engine = None
        SCHED().launch_order = None
    torch.cuda.synchronize()
    assert int((s1 != 0).sum()) == 0
    assert torch.equal(b1, b0) and torch.equal(s1, s0)            # the launch order is invisible in the records
    assert torch.equal(a1[:, 2].long().cpu(), torch.as_tensor(area))  # n_masked
    # split engine on the first 256 (its batch range): same records to rounding
    SCHED().engine = "split"
    try:
        bs, ss, _ = la.fit_instances(depth[:256], masks[:256], K)
    finally:
        SCHED().engine = None
    assert int((ss != 0).sum()) == 0
    np.testing.assert_allclose(bs[:, :15].cpu().numpy(), b1[:256, :15].cpu().numpy(), rtol=0, atol=1e-9)
    # oracle on the smallest, the largest and a spread of instances
    order = np.argsort(area)
    pick = np.unique(np.concatenate([order[:6], order[-4:], order[:: max(1, B // 22)]]))
    d, m = depth[pick].cpu().numpy(), masks[pick].cpu().numpy().astype(bool)
    ref, rst, _, _ = O.fit_instances(d, m, np.repeat(K640[None], len(pick), 0))
    got = b1[pick].cpu().numpy()
    assert (rst == 0).all()
    np.testing.assert_allclose(got[:, :15], ref[:, :15], rtol=0, atol=1e-9)
    np.testing.assert_allclose(got[:, 15:], ref[:, 15:], rtol=0, atol=2e-2)


def test_pipelined_batches_and_nocull_build_give_the_same_records(la, monkeypatch):
    """Scheduling never shows in the records: (a) batches issued through fit_batches (two streams, launch order off) and
    (b) the no-cull build (opt_build = LA3D_BUILD_NOCULL: two passes, every active tile walked in pass B) are bit-identical
    to plain serial calls."""
    import torch

    import bench

    dev = torch.device("cuda", 0)
    monkeypatch.setattr(SCHED(), "engine", "instance")
    batches, want = [], []
    for k in range(5):
        depth, masks, K, _, _ = bench.make_inputs(700 + 100 * k, dev, 40 + k)
        batches.append((depth, masks, K))
        want.append(tuple(t.clone() for t in la.fit_instances(depth, masks, K)))
    got = list(la.fit_batches(batches))          # results are clones by default: keeping all of them is safe
    assert len(got) == 5
    for g, w in zip(got, want):
        assert torch.equal(g[0], w[0]) and torch.equal(g[1], w[1]) and torch.equal(g[2], w[2])

    def lazily():   # batches PRODUCED while the generator runs (uploads on the current stream, freed right after the yield)
        for k in range(5):
            d, m, K_ = batches[k]
            yield d.clone(), m.clone(), K_.clone()
    with la.scheduling(launch_order=True):   # the caller's own (thread-local) choice; fit_batches switches the order off per call
        views = []
        for k, r in enumerate(la.fit_batches(lazily(), copy=False)):   # zero-copy views: valid until two more are requested
            assert torch.equal(r[0], want[k][0]) and torch.equal(r[1], want[k][1])
            views.append(r)
            if k >= 1:
                assert torch.equal(views[k - 1][0], want[k - 1][0])
        assert SCHED().launch_order is True      # fit_batches did not touch the caller's setting
    # the no-cull build = the two-pass plain build bit for bit; the default build takes the separable single pass for these
    # un-grounded calls (round 5): equal to rounding
    for (depth, masks, K), w in zip(batches[:2], want[:2]):
        monkeypatch.setattr(SCHED(), "build", "plain")
        w2 = tuple(t.clone() for t in la.fit_instances(depth, masks, K))
        monkeypatch.setattr(SCHED(), "build", "nocull")
        b, s, a = la.fit_instances(depth, masks, K)
        assert torch.equal(b, w2[0]) and torch.equal(s, w2[1]) and torch.equal(a, w2[2])
        assert torch.equal(s, w[1])
        torch.testing.assert_close(b[:, :15], w[0][:, :15], rtol=1e-11, atol=1e-11)
    # a batch with masks of more than 160 active tiles and a non-finite depth (checked re-run)
    depth, masks, K, _, _ = bench.make_config5(600, dev, 9)
    depth[3, 200, 300] = float("inf")
    masks[3, 190:260, 280:400] = 1
    monkeypatch.setattr(SCHED(), "build", "plain")
    w = la.fit_instances(depth, masks, K)
    monkeypatch.setattr(SCHED(), "build", "nocull")
    g = la.fit_instances(depth, masks, K)
    assert torch.equal(g[0], w[0]) and torch.equal(g[1], w[1])
    monkeypatch.setattr(SCHED(), "build", None)       # default: single pass; instance 3 (inf under the mask) re-runs the two-pass path
    d = la.fit_instances(depth, masks, K)
    assert torch.equal(d[1], w[1]) and torch.equal(d[0][3], w[0][3])
    torch.testing.assert_close(d[0][:, :15], w[0][:, :15], rtol=1e-11, atol=1e-11, equal_nan=True)


def test_subsample_mode_on_config5_mix(la):
    """Reference-subsample mode on the config-5 size mix: masks of <= 500 px are not sampled and walk their active tiles
    (bit-identical to full-mask mode), larger ones pick their 500 drawn pixels through the block prefix of the bit image.
    Also: a scattered mask with more active tiles than the list holds (dense walk), a non-finite depth under a small mask
    (checked re-run), and a frame width that is not a multiple of 32 (the untiled instantiation)."""
    import torch

    import bench

    dev = torch.device("cuda", 0)
    B = 1024
    depth, masks, K, _, _ = bench.make_config5(B, dev, 5)
    masks[7] = 0
    masks[7, ::9, ::41] = 1                      # 54 x 16 = 864 px scattered over 54 x 16 tiles -> sampled
    masks[8] = 0
    masks[8, 3::40, 5::64] = 1                   # 12 x 10 = 120 px, one per tile: not sampled, 120 active tiles
    masks[9] = 0
    masks[9, 200:210, 300:330] = 1               # 300 px with a NaN depth among them
    depth[9, 205, 310] = float("nan")
    counts = la.mask_counts(masks).cpu().numpy()
    assert (counts <= 500).sum() > 200 and (counts > 500).sum() > 200
    idx = la.draw_sample_idx(counts, np.random.RandomState(3))
    SCHED().engine = "instance"
    try:
        bs, ss, as_ = la.fit_instances(depth, masks, K, sample_idx=idx)
        SCHED().build = "plain"     # the two-pass walk the subsample build uses for its small masks: bit for bit
        bf, sf, af = la.fit_instances(depth, masks, K)
        SCHED().build = None        # the default full-mask call takes the separable single pass (round 5): equal to rounding
        bd, sd, _ = la.fit_instances(depth, masks, K)
    finally:
        SCHED().engine = None
        SCHED().build = None
    small = torch.as_tensor(counts <= 500, device=dev)
    assert torch.equal(bs[small], bf[small]) and torch.equal(ss[small], sf[small]) and torch.equal(as_[small], af[small])
    wide = small & (as_[:, 3] > 1e-4)      # (a few of the smallest masks are nearly isotropic: their axis is conditioned like 1 / gap)
    assert torch.equal(sd, sf)
    torch.testing.assert_close(bd[wide][:, :15], bf[wide][:, :15], rtol=1e-9, atol=1e-9)
    assert int((ss != 0).sum()) == 0
    order = np.argsort(counts)
    pick = np.unique(np.concatenate([[7, 8, 9], order[:4], order[-4:], order[:: B // 24]]))
    d, m = depth[pick].cpu().numpy(), masks[pick].cpu().numpy().astype(bool)
    ref, rst, _, rn = O.fit_instances(d, m, np.repeat(K640[None], len(pick), 0), sample_idx=idx[pick])
    got = bs[pick].cpu().numpy()
    assert (rst == 0).all()
    np.testing.assert_allclose(got[:, :15], ref[:, :15], rtol=0, atol=1e-9)
    np.testing.assert_allclose(got[:, 15:], ref[:, 15:], rtol=0, atol=2e-2)
    np.testing.assert_array_equal(as_[pick, 1].cpu().numpy(), rn)
    # frame width 600 (not a multiple of 32): untiled sample-mode instantiation, same answer as the oracle
    rs = np.random.RandomState(8)
    Hs, Ws = 120, 600
    d2 = rs.uniform(0.5, 10, (6, Hs, Ws)).astype(np.float32)
    m2 = np.zeros((6, Hs, Ws), bool)
    for i in range(6):
        h, w = rs.randint(4, 100), rs.randint(4, 500)
        r0, c0 = rs.randint(0, Hs - h + 1), rs.randint(0, Ws - w + 1)
        m2[i, r0:r0 + h, c0:c0 + w] = True
    m2[0] = False
    m2[0, 10:20, 10:30] = True
    K2 = np.array([[400.0, 0, 300], [0, 400.0, 60], [0, 0, 1]])
    idx2 = la.draw_sample_idx(m2.reshape(6, -1).sum(1), np.random.RandomState(4))
    b2, s2, _ = la.fit_instances(d2, m2, K2, sample_idx=idx2)
    ref2, rst2, _, _ = O.fit_instances(d2, m2, K2, sample_idx=idx2)
    assert (s2.cpu().numpy() == rst2).all()
    np.testing.assert_allclose(b2.cpu().numpy()[:, :15], ref2[:, :15], rtol=0, atol=1e-9)


def test_bench_multi_rank_path_two_ranks_one_gpu():
    """bench.py's N > 1 path (barriers, max-over-ranks timing, the final gather of records, n_gpus / value aggregation), run as
    the driver runs it (torch.distributed.run, one process per rank) with two ranks on this box's single GPU through gloo
    (LA3D_BENCH_BACKEND=gloo; RCCL refuses two ranks on one device).  A functional check, not a measurement."""
    import json
    import subprocess
    import sys

    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LA3D_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--batch", "512", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak" and d["config"]["instances_per_gpu"] == 512
    assert abs(d["value"] - 2 * 6 * 512 / (d["ms_per_step"] * 1e-3 * 6)) < 1e-6 * d["value"]     # whole-job aggregate over both ranks
    assert "DRY RUN" in d["config"]["sharding"] and "cpu_baseline" not in d
    assert 0 < d["roofline"]["frac"] < 1


def test_bench_config4_mode_two_ranks_one_gpu_equals_one_process(tmp_path):
    """bench.py --config4: ONE global metadata list -> plan_shards -> every rank materialises and fits only its image range ->
    one gather (the north_star partitioning, reference --start_index / --end_index).  Two gloo ranks sharing this box's GPU
    must deliver, in global instance order, exactly the records of the one-process job, which in turn match a direct
    fit_instances call on the whole materialised job; the JSON line carries per-rank fit times and the gather's time."""
    import json
    import subprocess
    import sys

    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for world in (1, 2):
        dump = str(tmp_path / f"rec{world}.npy")
        env = dict(os.environ, LA3D_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", LA3D_ENGINE="instance")
        base = [os.path.join(root, "bench.py"), "--gpus", str(world), "--config4", "60", "--jobs", "1", "--warmup-jobs", "1", "--dump", dump]
        cmd = ([sys.executable] + base if world == 1 else
               [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port())] + base)
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
        d = json.loads(lines[0])
        assert d["n_gpus"] == world and d["scaling"] == "strong" and len(d["per_rank_fit_ms"]) == world
        assert sum(d["config"]["instances_per_rank"]) == d["config"]["instances"] and sum(d["config"]["images_per_rank"]) == 60
        assert d["gather_ms"] >= 0 and d["fit_ms_max"] >= d["fit_ms_min"] > 0 and d["imbalance_max_over_mean"] >= 1.0
        assert abs(d["value"] - d["config"]["instances"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
        outs[world] = np.load(dump)
    # a shard boundary changes neither the instance engine's records (one workgroup per instance: no cross-instance arithmetic)
    np.testing.assert_array_equal(outs[1], outs[2])
    import bench
    import labelany3d_amd as la
    from .conftest import SCHED

    meta = bench.config4_metadata(60, 1234)
    dev = torch.device("cuda", 0)
    depth, masks, K, _, _ = bench.config4_materialize(meta, (0, 60, 0, meta["B"]), dev)
    SCHED().engine = "instance"
    try:
        b, s, _ = la.fit_instances(depth, masks, K, image_index=meta["img"].astype(np.int32))
    finally:
        SCHED().engine = None
    np.testing.assert_array_equal(b.cpu().numpy(), outs[1])


@pytest.mark.parametrize("fmt", ["--poly", "--rle"])
def test_bench_config4_annotation_formats_two_ranks_one_gpu_equals_one_process(tmp_path, fmt):
    """Round 5: the same partitioning on the reference's ANNOTATION formats (bench.py --config4 N --poly | --rle ->
    shard.fit_annotations_sharded): every rank holds the annotation list, packs and fits only the segmentations of its own image
    range (no u8 plane anywhere) and joins the one gather.  Two gloo ranks on this box's GPU = the one-process job bit for bit = a
    direct fit_instances_poly / _rle call over all annotations; against the oracle through the decoded planes."""
    import json
    import subprocess
    import sys

    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for world in (1, 2):
        dump = str(tmp_path / f"rec{world}.npy")
        env = dict(os.environ, LA3D_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
        base = [os.path.join(root, "bench.py"), "--gpus", str(world), "--config4", "40", fmt, "--jobs", "1", "--warmup-jobs", "1", "--dump", dump]
        cmd = ([sys.executable] + base if world == 1 else
               [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port())] + base)
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
        d = json.loads(lines[0])
        assert d["n_gpus"] == world and len(d["per_rank_fit_ms"]) == world and "u8" not in d["config"]["mask_input"]
        assert sum(d["config"]["instances_per_rank"]) == d["config"]["instances"] and sum(d["config"]["images_per_rank"]) == 40
        outs[world] = np.load(dump)
    np.testing.assert_array_equal(outs[1], outs[2])
    import bench
    import labelany3d_amd as la

    meta = bench.config4_metadata(40, 1234)
    dev = torch.device("cuda", 0)
    depth, _, K, _, _ = bench.config4_materialize(meta, (0, 40, 0, meta["B"]), dev, with_masks=False)
    anns = bench.config4_annotations(meta, "rle" if fmt == "--rle" else "poly")
    segs = [a["segmentation"] for a in anns]
    img = meta["img"].astype(np.int32)
    if fmt == "--rle":
        b, s, _ = la.fit_instances_rle(depth, segs, K, image_index=img)
        planes = la.rle_decode(segs).cpu().numpy().astype(bool)
    else:
        b, s, _ = la.fit_instances_poly(depth, la.pack_polygons(segs, bench.H, bench.W), K, image_index=img)
        planes = la.poly_decode(la.pack_polygons(segs, bench.H, bench.W)).cpu().numpy().astype(bool)
    np.testing.assert_array_equal(b.cpu().numpy(), outs[1])
    pick = np.arange(0, meta["B"], max(1, meta["B"] // 12))
    ref, rst, _, _ = O.fit_instances(depth.cpu().numpy(), planes[pick], np.broadcast_to(K.cpu().numpy(), (40, 3, 3)), depth_index=img[pick])
    assert (s.cpu().numpy()[pick] == rst).all()
    np.testing.assert_allclose(outs[1][pick][:, :15], ref[:, :15], rtol=0, atol=1e-8)


def test_config4_metadata_and_plan_cpu_side():
    """the metadata every rank derives and the plan over it: contiguous, complete, balanced by the cost model (no GPU work)"""
    import bench
    from labelany3d_amd.shard import plan_shards

    meta = bench.config4_metadata(500, 7)
    assert (np.diff(meta["img"]) >= 0).all() and meta["img"].max() == 499 and len(meta["area"]) == meta["B"]
    for world in (1, 2, 8):
        plan = plan_shards(meta["img"], 500, world, areas=meta["area"], frame_pixels=480 * 640)
        assert plan[0].img_lo == 0 and plan[-1].img_hi == 500 and plan[-1].inst_hi == meta["B"]
        for a, b in zip(plan, plan[1:]):
            assert a.img_hi == b.img_lo and a.inst_hi == b.inst_lo
        cost = [float((480 * 640 + 8 * meta["area"][p.inst_lo:p.inst_hi]).sum()) for p in plan]
        assert max(cost) / (sum(cost) / world) < 1.1


_NCCL_WORKER = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import bench
import labelany3d_amd as la
from labelany3d_amd.shard import fit_instances_sharded, gather_boxes

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
depth, masks, K, _, image_index = bench.make_config3(12, dev, 77)
B = masks.shape[0]
want = la.fit_instances(depth, masks, K, image_index=image_index)
got = fit_instances_sharded(depth, masks, K, image_index)          # plan -> local fit -> RCCL gather of DEVICE tensors
assert got is not None and got[0].is_cuda and got[1].is_cuda and got[2] == [B]
assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
# the gather on its own: ragged payloads are what the other ranks would send; with one rank it must be the identity, on the device
gb = gather_boxes(want[0][:7].contiguous(), want[1][:7].contiguous(), dst=0)
assert gb[0].is_cuda and torch.equal(gb[0], want[0][:7]) and gb[2] == [7]
empty = gather_boxes(want[0][:0].contiguous(), want[1][:0].contiguous(), dst=0)
assert empty[0].shape == (0, 39) and empty[2] == [0]
t = torch.ones(1, device=dev)
dist.all_reduce(t)
dist.barrier()
dist.destroy_process_group()
print("NCCL_OK", B)
"""


def test_rccl_branch_world_size_one(tmp_path):
    """The RCCL code path itself (backend "nccl" IS RCCL on ROCm): a one-rank process group on the one GPU runs
    fit_instances_sharded + gather_boxes with DEVICE tensors (no host staging, dist.gather / all_gather on RCCL) - the branch the
    gloo tests never touch.  No scaling claim; it proves the calls, dtypes and shapes are accepted by the real backend."""
    import subprocess
    import sys

    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "nccl_worker.py"
    script.write_text(_NCCL_WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, str(script), root, str(_free_port())], env=env, capture_output=True, text=True, timeout=600,
                       cwd=root)
    assert r.returncode == 0 and "NCCL_OK" in r.stdout, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])


def test_bench_rccl_branch_one_rank():
    """bench.py through its `dist is not None` branch on RCCL with WORLD_SIZE=1 (LA3D_BENCH_FORCE_DIST=1): communicator
    warm-up, barriers, the gather of all records inside the timed region and the MAX all-reduces run on the device."""
    import json
    import subprocess
    import sys

    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LA3D_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0",
               LOCAL_RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("LA3D_BENCH_BACKEND", None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--batch", "512",
           "--no-cpu-baseline", "--no-pipelined"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 5 and "RCCL gather" in d["config"]["sharding"]
    assert 0 < d["roofline"]["frac"] < 1
