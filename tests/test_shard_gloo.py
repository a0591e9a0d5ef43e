"""world_size-2 tests of the multi-GPU layer on CPU (gloo): partitioning, the ragged gather and the
sharded driver (with an injected CPU fit function — the GPU fit itself is covered by -m gpu tests)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from labelany3d_amd.shard import Shard, fit_annotations_sharded, fit_instances_sharded, gather_boxes, partition_contiguous, plan_shards


def test_partition_contiguous_balances_cost():
    rs = np.random.RandomState(0)
    cost = rs.uniform(1, 100, 1000)
    for world in (1, 2, 3, 8):
        parts = partition_contiguous(cost, world)
        assert parts[0][0] == 0 and parts[-1][1] == 1000
        assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
        loads = np.array([cost[a:b].sum() for a, b in parts])
        assert loads.max() <= cost.sum() / world + cost.max()
    assert partition_contiguous([], 2) == [(0, 0), (0, 0)]
    assert partition_contiguous([5.0], 3) in ([(0, 0), (0, 0), (0, 1)], [(0, 1), (1, 1), (1, 1)], [(0, 0), (0, 1), (1, 1)])


def test_plan_shards_uses_metadata_only():
    rs = np.random.RandomState(1)
    P = 50
    per = rs.randint(0, 9, P)
    per[10:14] = 0                                   # images without instances
    img = np.repeat(np.arange(P), per)
    areas = rs.randint(8, 100000, len(img)).astype(np.float64)
    for world in (1, 2, 3, 8):
        sh = plan_shards(img, P, world, areas=areas)
        assert len(sh) == world and sh[0].img_lo == 0 and sh[-1].img_hi == P and sh[0].inst_lo == 0 and sh[-1].inst_hi == len(img)
        for a, b in zip(sh, sh[1:]):
            assert a.img_hi == b.img_lo and a.inst_hi == b.inst_lo
        for s in sh:                                  # an image's instances never straddle two ranks
            assert ((img[s.inst_lo:s.inst_hi] >= s.img_lo) & (img[s.inst_lo:s.inst_hi] < s.img_hi)).all()
        cost = np.array([(307200 + 8 * areas[s.inst_lo:s.inst_hi]).sum() for s in sh])
        assert cost.max() <= cost.sum() / world + (307200 + 8 * 100000) * 9 + 4 * 307200
    # count-based when no areas are known; torch inputs
    sh = plan_shards(torch.as_tensor(img), P, 2)
    assert abs((sh[0].inst_hi - sh[0].inst_lo) - len(img) / 2) < 20
    with pytest.raises(ValueError):
        plan_shards(img[::-1], P, 2)
    with pytest.raises(ValueError):
        plan_shards(img, P, 2, areas=areas[:-1])
    assert plan_shards(np.zeros(0, np.int64), 0, 2) == [Shard(0, 0, 0, 0), Shard(0, 0, 0, 0)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_fit(depth, masks, K, ground=None, sample_idx=None, image_index=None):
    """CPU stand-in with the product function's contract (depth / K / image_index are the rank's LOCAL slice): record n
    encodes (global image read off the depth plane, mask area)."""
    B = masks.shape[0]
    rec = torch.zeros((B, 39), dtype=torch.float64)
    assert depth.shape[0] > int(np.asarray(image_index).max(initial=-1))     # local indices address the local planes
    rec[:, 0] = depth[torch.as_tensor(np.asarray(image_index), dtype=torch.long), 1, 1].double()
    rec[:, 1] = masks.reshape(B, -1).ne(0).sum(1).double()
    rec[:, 2] = depth[torch.as_tensor(np.asarray(image_index), dtype=torch.long), 0, 0].double()
    return rec, torch.zeros(B, dtype=torch.int32), None


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # ragged gather: rank r contributes r+2 rows
        n = rank + 2
        boxes = torch.full((n, 39), float(rank), dtype=torch.float64) + torch.arange(n, dtype=torch.float64)[:, None] / 10
        status = torch.full((n,), rank, dtype=torch.int32)
        out = gather_boxes(boxes, status, dst=0)
        if rank == 0:
            b, s, counts = out
            assert counts == [2, 3] and b.shape == (5, 39)
            assert s.tolist() == [0, 0, 1, 1, 1]
            assert torch.equal(b[:, 0], torch.tensor([0.0, 0.1, 1.0, 1.1, 1.2], dtype=torch.float64))
        else:
            assert out is None
        # empty shard on one rank
        out = gather_boxes(boxes[: (0 if rank == 1 else 2)], status[: (0 if rank == 1 else 2)], dst=0)
        if rank == 0:
            assert out[2] == [2, 0] and out[0].shape == (2, 39)
        # sharded driver: identical global description on every rank
        rs = np.random.RandomState(5)
        P, H, W = 7, 8, 10
        depth = torch.arange(P, dtype=torch.float32).view(P, 1, 1).expand(P, H, W).contiguous()
        per = rs.randint(0, 5, P)
        img = np.repeat(np.arange(P), per).astype(np.int32)
        Bt = len(img)
        masks = torch.as_tensor(rs.rand(Bt, H, W) < 0.3)
        out = fit_instances_sharded(depth, masks, np.eye(3), img, fit_fn=_fake_fit)
        if rank == 0:
            b, s, counts = out
            assert sum(counts) == Bt and b.shape == (Bt, 39)
            assert b[:, 0].tolist() == img.astype(float).tolist()           # global instance order kept
            assert b[:, 1].tolist() == masks.reshape(Bt, -1).sum(1).double().tolist()
            assert b[:, 2].tolist() == img.astype(float).tolist()           # each instance saw its own depth plane
            assert all(c > 0 for c in counts)
        # the same through load_fn: a rank materialises ONLY its own slice; areas drive the balance
        areas = masks.reshape(Bt, -1).sum(1).numpy()
        seen = {}

        def load(sh):
            seen["shard"] = sh
            return depth[sh.img_lo:sh.img_hi].clone(), masks[sh.inst_lo:sh.inst_hi].clone(), np.eye(3), None, None

        out2 = fit_instances_sharded((P, H, W), None, None, img, areas=areas, fit_fn=_fake_fit, load_fn=load)
        assert seen["shard"] == plan_shards(img, P, world, areas=areas, frame_pixels=H * W)[rank]
        if rank == 0:
            assert torch.equal(out2[0], b) or out2[2] != counts     # same records whenever the cut is the same ...
            assert out2[0][:, 0].tolist() == img.astype(float).tolist()   # ... and always in global order
            assert out2[0][:, 1].tolist() == areas.astype(float).tolist()
        # every instance on rank 0 (all images but the first are empty): rank 1 joins the gather with nothing
        img1 = np.zeros(3, np.int32)
        out3 = fit_instances_sharded(depth, masks[:3], np.eye(3), img1, fit_fn=_fake_fit)
        if rank == 0:
            assert sorted(out3[2]) == [0, 3] and out3[0].shape == (3, 39)
        # round 5: the same partitioning on the reference's ANNOTATION formats - every rank holds the annotation list (metadata) and
        # loads only its own depth planes; the plan balances by the annotations' "area" fields
        anns = []
        for n, im in enumerate(img):
            seg = ([[float(n), 1.0, n + 3.0, 1.0, n + 3.0, 5.0]] if n % 3 else {"size": [H, W], "counts": [n % 7, 5, H * W - 5 - n % 7]})
            anns.append({"id": 100 + n, "image_id": int(im), "category_id": 1 + n % 5, "bbox": [0, 0, 3, 3], "area": float(10 + 7 * n),
                         "iscrowd": int(n == 4), "segmentation": seg})
        loaded = {}

        def depth_loader(sh):
            loaded["shard"] = sh
            return depth[sh.img_lo:sh.img_hi].clone(), np.eye(3)

        def fake_ann_fit(annotations, image_size, d, K, ground=None, image_index=None, filter=None):
            assert image_size == (W, H) and d.shape[0] > int(np.asarray(image_index).max(initial=-1))
            rec = torch.zeros((len(annotations), 39), dtype=torch.float64)
            rec[:, 0] = d[torch.as_tensor(np.asarray(image_index), dtype=torch.long), 2, 2].double()   # the plane each annotation saw
            rec[:, 1] = torch.tensor([a["id"] for a in annotations], dtype=torch.float64)
            st = torch.tensor([6 if a["iscrowd"] else 0 for a in annotations], dtype=torch.int32)
            return rec, st

        out4 = fit_annotations_sharded(anns, (W, H), img, P, depth_loader, fit_fn=fake_ann_fit)
        want_plan = plan_shards(img, P, world, areas=np.array([a["area"] for a in anns]), frame_pixels=H * W)
        assert loaded["shard"] == want_plan[rank]                         # areas came from the annotations themselves
        if rank == 0:
            b4, s4, c4 = out4
            assert c4 == [p.inst_hi - p.inst_lo for p in want_plan] and b4.shape == (Bt, 39)
            assert b4[:, 0].tolist() == img.astype(float).tolist()        # every annotation saw its own image's depth plane
            assert b4[:, 1].tolist() == [100.0 + n for n in range(Bt)]    # global annotation order
            assert s4.tolist() == [6 if n == 4 else 0 for n in range(Bt)]
        with pytest.raises(ValueError):
            fit_annotations_sharded(anns[:-1], (W, H), img, P, depth_loader, fit_fn=fake_ann_fit)
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, repr(e) + " " + traceback.format_exc()[-600:]))
        raise
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    res = sorted(q.get(timeout=5) for _ in range(2))
    assert res == [(0, "ok"), (1, "ok")], res
    assert all(p.exitcode == 0 for p in procs)
