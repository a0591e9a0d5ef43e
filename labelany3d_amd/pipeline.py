"""Pipelined issue of independent batches: batch k+1 is enqueued on a second HIP stream while batch k
runs, the size-balanced launch order is switched off for these calls (it assumes an idle chip), and the results come back
in order.  On an MI355X this turns 110 us per 1024-instance call into 81-92 us.

    for boxes, status, aux in fit_batches(batches):        # batches: iterable of dicts / tuples for InstanceFitter.run
        ...

The reference processes one image after the other in a Python loop (src/batch_scripts/whole.py:42, :72); this is the
counterpart for a stream of per-image (or per-chunk) batches whose tensors are already on the GPU.
"""
from __future__ import annotations

from typing import Iterable, Iterator, Tuple

import torch

from .batched import InstanceFitter


def fit_batches(batches: Iterable, streams: int = 2, copy: bool = True) -> Iterator[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
    """``batches`` yields ``(depth, masks, K)`` tuples or dicts with the keyword arguments of ``InstanceFitter.run``
    (``depth, masks, K, ground, sample_idx, image_index``); all tensors on the GPU with the ABI's dtypes (f32 / u8|bool / f64 /
    f64 / i32 / i32).  Yields ``(boxes (B,39), status (B,), aux (B,4))`` per batch, in order; every result is complete (its
    stream has been waited for) when it is handed out.

    Ordering and lifetime: batches are pulled lazily, so before batch k is issued its pool stream waits for the stream that is
    current at that moment (where the caller produced the tensors), and every input tensor is recorded on the pool stream so
    that the caching allocator cannot hand its memory to the producer of batch k+1 while batch k still reads it.
    ``copy=True`` (default) hands out clones: results stay valid for as long as the caller keeps them (``list(fit_batches(..))``
    is safe).  ``copy=False`` hands out views of internal buffers (``streams + 1`` rotating sets per batch shape): a result is
    overwritten once the generator has been advanced TWICE more - use it before asking for the result after next.
    The size-balanced launch order is switched off PER CALL (``la3d_fit_args::opt_launch_order``): no process state is touched."""
    dev = None
    pool, fitters = [], {}
    pending = []          # (event, result) in issue order
    nbuf = max(1, streams) + 1
    for k, b in enumerate(batches):
        kw = dict(b) if isinstance(b, dict) else dict(zip(("depth", "masks", "K"), b))
        masks = kw["masks"]
        if masks.dtype == torch.bool:
            kw["masks"] = masks.view(torch.uint8)
        if dev is None:
            dev = masks.device
            pool = [torch.cuda.Stream(device=dev) for _ in range(max(1, streams))]
        st = pool[k % len(pool)]
        st.wait_stream(torch.cuda.current_stream(dev))   # this batch's tensors were produced on the stream current NOW
        for t in kw.values():
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(st)
        B, H, W = kw["masks"].shape
        key = (B, H, W, k % nbuf)
        if key not in fitters:
            fitters[key] = InstanceFitter(B, H, W, dev)
        while len(pending) >= len(pool):    # keep at most `streams` batches in flight: hand out the oldest result
            ev, res = pending.pop(0)
            ev.synchronize()
            yield tuple(t.clone() for t in res) if copy else res
        res = fitters[key].run(stream=st, launch_order=False, **kw)
        ev = torch.cuda.Event()
        ev.record(st)
        pending.append((ev, res))
    for ev, res in pending:
        ev.synchronize()
        yield tuple(t.clone() for t in res) if copy else res
