"""Multi-GPU layer: one process per GPU, instances sharded across ranks, ONE result gather.

The reference scales out only by running separate processes over contiguous index ranges
(``--start_index/--end_index/--gpu_idx``, reference src/batch_scripts/whole.py:25-27,42); instances
are independent, so the data path needs no collective.  The only exchange is the final gather of
the (n_i, 39) box tensors (+ status) to one rank — RCCL over xGMI on GPUs (torch.distributed backend
"nccl"), gloo on CPU in the tests.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def partition_contiguous(costs: Sequence[float], world: int) -> List[Tuple[int, int]]:
    """Split items 0..n-1 into ``world`` contiguous [start, end) ranges with balanced total cost
    (cost of an image = sum of its instances' mask areas + one depth plane).  Contiguous ranges keep
    every image's instances — and its depth plane — on one GPU, like the reference's index ranges."""
    c = np.asarray(costs, dtype=np.float64)
    n = len(c)
    if world <= 0:
        raise ValueError("world must be positive")
    cum = np.concatenate([[0.0], np.cumsum(c)])
    total = cum[-1]
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        k = int(np.searchsorted(cum, target, side="left"))
        # choose the boundary whose prefix cost is closest to the target
        if k > 0 and (k > n or abs(cum[k - 1] - target) <= abs(cum[min(k, n)] - target)):
            k -= 1
        cuts.append(min(max(k, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[i], cuts[i + 1]) for i in range(world)]


def gather_boxes(boxes: torch.Tensor, status: torch.Tensor, dst: int = 0, group=None):
    """Gather every rank's (n_i, 39) float64 records and (n_i,) int32 status to ``dst`` in rank order.

    Ranks may hold different n_i: counts are exchanged first (one tiny all_gather), payloads are padded
    to the maximum and gathered with a single ``dist.gather`` each.  Returns ``(boxes, status, counts)``
    on ``dst`` and ``None`` elsewhere.  ~312 B per box: 860k boxes over 8 GPUs is 33 MB per rank —
    irrelevant next to the compute, so the simplest correct collective is used.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = torch.tensor([boxes.shape[0]], dtype=torch.int64, device=boxes.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c) for c in counts]
    nmax = max(counts)
    pb = boxes if boxes.shape[0] == nmax else torch.cat(
        [boxes, boxes.new_zeros((nmax - boxes.shape[0], boxes.shape[1]))])
    ps = status if status.shape[0] == nmax else torch.cat([status, status.new_zeros(nmax - status.shape[0])])
    pb, ps = pb.contiguous(), ps.contiguous()
    try:
        if rank == dst:
            gb = [torch.empty_like(pb) for _ in range(world)]
            gs = [torch.empty_like(ps) for _ in range(world)]
        else:
            gb = gs = None
        dist.gather(pb, gb, dst=dst, group=group)
        dist.gather(ps, gs, dst=dst, group=group)
    except (RuntimeError, NotImplementedError):  # a backend without gather: every rank collects, dst keeps
        gb = [torch.empty_like(pb) for _ in range(world)]
        gs = [torch.empty_like(ps) for _ in range(world)]
        dist.all_gather(gb, pb, group=group)
        dist.all_gather(gs, ps, group=group)
    if rank != dst:
        return None
    return (torch.cat([g[:c] for g, c in zip(gb, counts)]), torch.cat([g[:c] for g, c in zip(gs, counts)]), counts)


def fit_instances_sharded(depth, masks, K, image_index, ground=None, sample_idx=None, dst: int = 0, group=None,
                          fit_fn=None):
    """Every rank passes the SAME global description (or at least its own slice of it); images are
    split into contiguous, cost-balanced ranges, each rank fits the instances of its images on its own
    GPU and the records are gathered on ``dst`` in global instance order.

    depth (P,H,W), K (P,3,3) or (3,3), masks (B,H,W), image_index (B,) non-decreasing.
    ``fit_fn`` defaults to labelany3d_amd.fit_instances (injectable so the sharding logic is testable
    on CPU with gloo)."""
    if fit_fn is None:
        from .batched import fit_instances as fit_fn
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    img = np.asarray(image_index.cpu() if isinstance(image_index, torch.Tensor) else image_index).astype(np.int64)
    if (np.diff(img) < 0).any():
        raise ValueError("image_index must be non-decreasing (instances grouped by image)")
    P = depth.shape[0]
    m = masks if isinstance(masks, torch.Tensor) else torch.as_tensor(np.asarray(masks))
    area = m.reshape(m.shape[0], -1).ne(0).sum(1).cpu().numpy().astype(np.float64)
    hw = float(m.shape[1] * m.shape[2])
    cost = np.bincount(img, weights=area + hw, minlength=P) + 4.0 * hw * (np.bincount(img, minlength=P) > 0)
    lo_img, hi_img = partition_contiguous(cost, world)[rank]
    sel = np.nonzero((img >= lo_img) & (img < hi_img))[0]
    lo, hi = (int(sel[0]), int(sel[-1]) + 1) if len(sel) else (0, 0)
    sl = slice(lo, hi)
    boxes, status, _ = fit_fn(depth, masks[sl], K, ground=None if ground is None else ground[sl],
                              sample_idx=None if sample_idx is None else sample_idx[sl],
                              image_index=image_index[sl])
    return gather_boxes(boxes, status, dst=dst, group=group)
