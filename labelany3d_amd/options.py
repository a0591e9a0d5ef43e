"""Per-call scheduling options of the batched fit (C-ABI ``la3d_fit_args::opt_engine / opt_launch_order / opt_build``).

The library keeps no mutable process state: which engine fits a batch, whether the size-balanced launch order runs and which
build of the instance kernel is used are decided per call.  These options pin a decision for the calls made inside a
``with scheduling(...)`` block (thread-local, so two threads can run differently scheduled calls side by side) or for one
call through the keyword arguments of ``InstanceFitter.run``.  They steer SPEED only: records never depend on them (the split
engine groups its fp64 partial sums by tile range, so split vs instance agree to rounding, not bit for bit - INTEGRATION.md).

    with scheduling(engine="split"):            # or "instance", "band", "rows", "rows2"
        boxes, status, aux = fit_instances(depth, masks, K)
    with scheduling(launch_order=False):         # a caller pipelining independent batches on several streams
        ...
    with scheduling(build="plain"):              # or "nocull": two passes, every active tile walked in pass B
        ...
"""
from __future__ import annotations

import contextlib
import threading

ENGINE = {None: 0, "instance": 1, "split": 2, "band": 3, "rows": 4, "rows2": 5}
ORDER = {None: 0, False: 1, True: 2}
BUILD = {None: 0, "plain": 1, "nocull": 2, "retaining": 2}   # ("retaining": the deleted 128-VGPR build; its value now means "nocull")


class _Sched(threading.local):
    engine = None          # None | "instance" | "split" | "band" | "rows" | "rows2"
    launch_order = None    # None | False | True
    build = None           # None | "plain" | "nocull"


sched = _Sched()
_UNSET = object()


@contextlib.contextmanager
def scheduling(engine=_UNSET, launch_order=_UNSET, build=_UNSET):
    """Pin scheduling decisions for the fit calls of this thread inside the block (``None`` = the library's choice)."""
    prev = (sched.engine, sched.launch_order, sched.build)
    try:
        if engine is not _UNSET:
            codes(engine=engine)
            sched.engine = engine
        if launch_order is not _UNSET:
            sched.launch_order = None if launch_order is None else bool(launch_order)
        if build is not _UNSET:
            codes(build=build)
            sched.build = build
        yield sched
    finally:
        sched.engine, sched.launch_order, sched.build = prev


def codes(engine=None, launch_order=None, build=None):
    """(opt_engine, opt_launch_order, opt_build) for one call: explicit arguments win over the thread's ``scheduling`` block."""
    e = engine if engine is not None else sched.engine
    o = launch_order if launch_order is not None else sched.launch_order
    b = build if build is not None else sched.build
    if e not in ENGINE:
        raise ValueError(f"engine must be one of {sorted(k for k in ENGINE if k)} or None, not {e!r}")
    if b not in BUILD:
        raise ValueError(f"build must be one of {sorted(k for k in BUILD if k)} or None, not {b!r}")
    return ENGINE[e], ORDER[None if o is None else bool(o)], BUILD[b]
