"""world_size-2 tests of the multi-GPU layer on CPU (gloo): partitioning, the ragged gather and the
sharded driver (with an injected CPU fit function — the GPU fit itself is covered by -m gpu tests)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from labelany3d_amd.shard import fit_instances_sharded, gather_boxes, partition_contiguous


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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_fit(depth, masks, K, ground=None, sample_idx=None, image_index=None):
    """CPU stand-in with the product function's contract: record n encodes (image, mask area)."""
    B = masks.shape[0]
    rec = torch.zeros((B, 39), dtype=torch.float64)
    rec[:, 0] = torch.as_tensor(np.asarray(image_index), dtype=torch.float64)
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
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))
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
