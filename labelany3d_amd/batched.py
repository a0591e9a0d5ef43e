"""Batched, device-resident entry points over the C-ABI (include/la3d.h).

PyTorch is used only for device memory and streams; every computation is a HIP kernel in
libla3d.so.  Inputs may be torch tensors on the GPU (zero-copy) or NumPy arrays (uploaded).

The composition these functions implement is defined in SURVEY.md §3.3:
    boxes[n] = estimate_bbox(depth_to_points(depth[img(n)][None], K[img(n)])[masks[n]], None, ground[n], 'pca')
with depth_to_points = reference src/util.py:52-75 and estimate_bbox = reference
src/util_3dbox.py:106-178.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib, options
from ._lib import AUX, NSAMPLE, REC, FitArgs, check, lib


def _dev(device=None) -> torch.device:
    if device is not None:
        return torch.device(device)
    if not torch.cuda.is_available():
        raise _lib.La3dError("no GPU visible: labelany3d_amd has no CPU path (libla3d.so is a HIP library)")
    return torch.device("cuda", torch.cuda.current_device())


_SMALL_UPLOADS: dict = {}   # (bytes, dtype, shape, device) -> device tensor: small host arrays that callers pass again and again (K)


def _as_dev(x, dtype, device, cache: bool = False) -> torch.Tensor:
    if isinstance(x, torch.Tensor):
        t = x
        if t.dtype == torch.bool and dtype == torch.uint8:
            t = t.view(torch.uint8) if t.is_contiguous() else t.contiguous().view(torch.uint8)
        elif t.dtype != dtype:
            t = t.to(dtype)
        if t.device != device:
            t = t.to(device)
        return t.contiguous()
    a = np.asarray(x)
    if dtype == torch.uint8 and a.dtype == np.bool_:
        a = a.view(np.uint8)
    if cache and a.nbytes <= 1024:   # opt-in (the 3x3 intrinsics): one upload per distinct value instead of one per call
        key = (a.tobytes(), str(a.dtype), a.shape, dtype, str(device))
        hit = _SMALL_UPLOADS.get(key)
        if hit is None:
            if len(_SMALL_UPLOADS) > 256:
                _SMALL_UPLOADS.clear()
            t = torch.as_tensor(np.ascontiguousarray(a), device=device).to(dtype).contiguous()
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(device))       # the upload / conversion runs on the stream current NOW
            hit = _SMALL_UPLOADS[key] = (t, ev)
        t, ev = hit
        torch.cuda.current_stream(device).wait_event(ev)          # a later user on another stream is ordered behind the upload
        return t                                                  # read-only by contract: shared between callers
    return torch.as_tensor(np.ascontiguousarray(a), device=device).to(dtype).contiguous()


def _upload_many(arrays, device, pinned: Optional[torch.Tensor] = None):
    """Several small host arrays -> device tensors with ONE host-to-device copy (each separate upload costs ~17 us of launch
    overhead on this stack; the per-image wrappers have six to eight of them).  ``arrays``: list of (ndarray | None, torch dtype);
    None stays None.  The views share one device buffer (16-byte aligned pieces).  ``pinned``: a pinned uint8 staging buffer of at
    least the packed size - the copy is then asynchronous on the current stream (the caller keeps the buffer untouched until that
    stream has passed the copy)."""
    metas, total = [], 0
    for a, dt in arrays:
        if a is None:
            metas.append(None)
            continue
        h = np.ascontiguousarray(np.asarray(a))
        want = torch.empty(0, dtype=dt).numpy().dtype
        if h.dtype != want:
            h = h.astype(want)
        metas.append((h, total))
        total += (h.nbytes + 15) & ~15
    if total == 0:
        return [None if m is None else torch.empty(m[0].shape, dtype=arrays[i][1], device=device) for i, m in enumerate(metas)]
    if pinned is not None and pinned.numel() >= total:
        buf = pinned.numpy()[:total]
    else:
        pinned = None
        buf = np.empty(total, np.uint8)
    for m in metas:
        if m is not None and m[0].nbytes:
            buf[m[1]:m[1] + m[0].nbytes] = m[0].reshape(-1).view(np.uint8)
    if pinned is not None:
        dbuf = torch.empty(total, dtype=torch.uint8, device=device)
        dbuf.copy_(pinned[:total], non_blocking=True)
    else:
        dbuf = torch.as_tensor(buf, device=device)
    out = []
    for (a, dt), m in zip(arrays, metas):
        if m is None:
            out.append(None)
        else:
            h, off = m
            out.append(dbuf[off:off + h.nbytes].view(dt).view(h.shape))
    return out


def _bulk(device, *pairs):
    """(value, torch dtype) pairs -> the values with every HOST array among them uploaded in one copy (_upload_many); tensors and
    None pass through untouched (the later _as_dev calls convert / move tensors as before)."""
    host = [i for i, (v, _) in enumerate(pairs) if v is not None and not isinstance(v, torch.Tensor)]
    out = [v for v, _ in pairs]
    if len(host) >= 2:
        up = _upload_many([pairs[i] for i in host], device)
        for i, t in zip(host, up):
            out[i] = t
    return out


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _order(stream) -> None:
    """A caller-supplied stream that is not the current one is ordered behind the current stream: the convenience wrappers
    upload / convert their arguments (and reuse the cached small uploads of _as_dev) on the CURRENT stream."""
    if stream is not None:
        cur = torch.cuda.current_stream()
        if stream != cur:
            stream.wait_stream(cur)


def _stream(stream=None, raw: bool = False):
    """The launch stream as a C handle.  raw=True (InstanceFitter.run: a pure enqueue whose inputs the caller has made ready on
    `stream`, e.g. batches pipelined on several streams) skips the ordering of _order()."""
    if not raw:
        _order(stream)
    s = torch.cuda.current_stream() if stream is None else stream
    return C.c_void_p(s.cuda_stream)


def _record(stream, *tensors):
    """Temporaries allocated on the current stream but consumed by kernels on ``stream``: tell the caching allocator, so the
    blocks are not handed out again while those kernels still read them."""
    if stream is None or stream == torch.cuda.current_stream():
        return
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            t.record_stream(stream)


def unpack_boxes(rec):
    """(B,39) record -> dict of center_cam (B,3), dimensions (B,3) = [dz,dy,dx], R_cam (B,3,3),
    bbox3D_cam (B,8,3): the four return values of the reference's estimate_bbox and the keys of its
    3dbbox.json (reference src/util_3dbox.py:283-290)."""
    return dict(center_cam=rec[..., 0:3], dimensions=rec[..., 3:6],
                R_cam=rec[..., 6:15].reshape(*rec.shape[:-1], 3, 3),
                bbox3D_cam=rec[..., 15:39].reshape(*rec.shape[:-1], 8, 3))


def mask_counts(masks, stream=None) -> torch.Tensor:
    """True pixels per mask plane (what the reference sees as in_pc.shape[0], :123)."""
    dev = masks.device if isinstance(masks, torch.Tensor) and masks.is_cuda else _dev()
    m = _as_dev(masks, torch.uint8, dev)
    B, H, W = m.shape
    out = torch.empty(B, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(lib.la3d_mask_counts(_ptr(m), B, H, W, _ptr(out), _stream(stream)), "la3d_mask_counts")
    return out


def draw_sample_idx(counts, rng=None) -> np.ndarray:
    """The indices the reference would draw, in instance order, from the global NumPy stream:
    ``np.random.randint(0, N, 500)`` for every instance with N > 500 (reference
    src/util_3dbox.py:123-125); instances with N <= 500 consume nothing (row left 0)."""
    counts = np.asarray(counts.cpu() if isinstance(counts, torch.Tensor) else counts)
    rng = np.random if rng is None else rng
    idx = np.zeros((len(counts), NSAMPLE), np.int32)
    for n, c in enumerate(counts):
        if c > NSAMPLE:
            idx[n] = rng.randint(0, int(c), NSAMPLE)
    return idx


class InstanceFitter:
    """Reusable launch state for fit_instances on fixed (B,H,W): owns the output / status / aux /
    workspace buffers so the steady-state call allocates nothing and is a pure enqueue."""

    def __init__(self, B: int, H: int, W: int, device=None, slots: int = 1, ws_slots: int = 1):
        self.B, self.H, self.W = int(B), int(H), int(W)
        self.device = _dev(device)
        self.slots = slots
        nbytes = int(lib.la3d_workspace_bytes(self.B, self.H, self.W))
        wsz = max((nbytes + 255) // 256 * 256, 256)
        # ONE device allocation carved into the four buffers (a call through the convenience wrappers allocates once)
        up = lambda v: (v + 255) // 256 * 256  # noqa: E731  (every region starts 256-byte aligned, like a fresh allocation)
        nb, na, ns = slots * B * REC * 8, slots * B * AUX * 8, slots * B * 4
        o_aux, o_st, o_ws = up(nb), up(nb) + up(na), up(nb) + up(na) + up(ns)
        self._arena = torch.empty(o_ws + ws_slots * wsz, dtype=torch.uint8, device=self.device)
        self.boxes = self._arena[:nb].view(torch.float64).view(slots, B, REC)
        self.aux = self._arena[o_aux:o_aux + na].view(torch.float64).view(slots, B, AUX)
        self.status = self._arena[o_st:o_st + ns].view(torch.int32).view(slots, B)
        # one workspace per concurrently running call (calls on different streams must not share it)
        self.workspace = self._arena[o_ws:].view(ws_slots, wsz)

    def run(self, depth: torch.Tensor, masks: torch.Tensor, K: torch.Tensor, ground=None, sample_idx=None,
            image_index=None, slot: int = 0, stream=None, ws_slot: int = 0, engine=None, launch_order=None, build=None,
            area_hint=None):
        """All arguments already on the device with the ABI's dtypes (f32 / u8 / f64 / f64 / i32 / i32).
        ``engine`` / ``launch_order`` / ``build``: scheduling of THIS call (labelany3d_amd.options; speed only)."""
        B, H, W = self.B, self.H, self.W
        planes = depth.shape[0] if depth.dim() == 3 else 1
        dstride = H * W if planes > 1 else 0
        kstride = 9 if (K.dim() == 3 and K.shape[0] > 1) else 0
        oe, oo, ob = options.codes(engine, launch_order, build)
        if oe or oo or ob or area_hint is not None:   # per-call options travel in the argument block
            a = FitArgs()
            a.struct_size = C.sizeof(FitArgs)
            a.B, a.H, a.W = B, H, W
            a.depth, a.depth_plane_stride, a.image_index = _ptr(depth), dstride, _ptr(image_index)
            a.mask, a.K, a.k_stride = _ptr(masks), _ptr(K), kstride
            a.ground, a.sample_idx, a.area_hint = _ptr(ground), _ptr(sample_idx), _ptr(area_hint)
            a.out, a.status, a.aux = _ptr(self.boxes[slot]), _ptr(self.status[slot]), _ptr(self.aux[slot])
            a.workspace, a.stream = _ptr(self.workspace[ws_slot]), _stream(stream, raw=True)
            a.opt_engine, a.opt_launch_order, a.opt_build = oe, oo, ob
            check(lib.la3d_fit_instances_ex(C.byref(a)), "la3d_fit_instances_ex")
            return self.boxes[slot], self.status[slot], self.aux[slot]
        rc = lib.la3d_fit_instances(_ptr(depth), dstride, _ptr(image_index), _ptr(masks), _ptr(K), kstride,
                                    _ptr(ground), _ptr(sample_idx), B, H, W, _ptr(self.boxes[slot]),
                                    _ptr(self.status[slot]), _ptr(self.aux[slot]), _ptr(self.workspace[ws_slot]),
                                    _stream(stream, raw=True))
        check(rc, "la3d_fit_instances")
        return self.boxes[slot], self.status[slot], self.aux[slot]


def _check_image_index(given, ii, B, P):
    """image_index must lie in [0, P): an index outside makes the kernel read another allocation (the C-ABI takes no plane count).
    Host arrays are checked on the host, a device tensor on the device - on EVERY call: a tensor refilled through its data pointer
    (this library's kernels, another C-ABI user, a DLPack alias) keeps its object identity and its version counter, so no cache
    of "already checked" tensors is sound.  ~40 us (two reductions + the read-back) for a device tensor; `InstanceFitter.run` is
    the entry for callers that have validated their index once and loop."""
    if B == 0:
        return
    if not (isinstance(given, torch.Tensor) and given.is_cuda):
        h = np.asarray(given.numpy() if isinstance(given, torch.Tensor) else given)
        if h.min() < 0 or h.max() >= P:
            raise ValueError("image_index out of range")
        return
    lo, hi = torch.aminmax(ii)
    if int(lo) < 0 or int(hi) >= P:
        raise ValueError("image_index out of range")


def pad_rows_f32(d: torch.Tensor, Wp: int) -> torch.Tensor:
    """(..., H, W) float32 on the device -> (..., H, Wp) with zeros on the right (C-ABI ``la3d_pad_rows``, current stream)."""
    W = int(d.shape[-1])
    d = d.contiguous()
    with torch.cuda.device(d.device):
        out = torch.empty(d.shape[:-1] + (Wp,), dtype=torch.float32, device=d.device)
        check(lib.la3d_pad_rows(_ptr(d), d.numel() // W, W, Wp, _ptr(out), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "la3d_pad_rows")
    return out


def fit_instances(depth, masks, K, ground=None, sample_idx=None, image_index=None, stream=None, device=None):
    """Batched composed hot path on the GPU.

    depth        (P,H,W) or (H,W) float32 — P planes; one plane = shared by all instances
    masks        (B,H,W) bool / uint8 (non-zero = True), reference layout src/util.py:367,382
    K            (P,3,3) or (3,3) float64 pixel intrinsics
    ground       (B,4) float64 or None; rows whose first entry is NaN mean "no ground"
    sample_idx   None = full-mask mode; (B,500) int = reference-subsample mode (see draw_sample_idx)
    image_index  (B,) int — depth plane / K of each instance (default: instance n -> plane n, or 0)
    Returns (boxes (B,39) f64, status (B,) i32, aux (B,4) f64 = yaw, n_valid, n_masked, eigen-gap), on the GPU.
    """
    if device is None and isinstance(masks, torch.Tensor) and masks.is_cuda:
        device = masks.device
    dev = _dev(device)
    m = _as_dev(masks, torch.uint8, dev)
    if m.dim() != 3:
        raise ValueError("masks must be (B,H,W)")
    B, H, W = m.shape
    d = _as_dev(depth, torch.float32, dev)
    if d.dim() == 2:
        d = d[None]
    if d.shape[1:] != (H, W):
        raise ValueError(f"depth planes {tuple(d.shape[1:])} do not match masks {(H, W)}")
    k = _as_dev(K, torch.float64, dev, cache=True)
    if k.dim() == 2:
        k = k[None]
    P = d.shape[0]
    if k.shape[0] not in (1, P) or k.shape[1:] != (3, 3):
        raise ValueError("K must be (3,3) or (P,3,3)")
    ii = None
    if image_index is not None:
        ii = _as_dev(image_index, torch.int32, dev)
        if ii.shape != (B,):
            raise ValueError("image_index must be (B,)")
        _check_image_index(image_index, ii, B, P)
    elif P not in (1, B):
        raise ValueError("without image_index, depth must have 1 or B planes")
    g = None
    if ground is not None:
        g = _as_dev(ground, torch.float64, dev)
        if g.shape != (B, 4):
            raise ValueError("ground must be (B,4)")
    si = None
    if sample_idx is not None:
        si = _as_dev(sample_idx, torch.int32, dev)
        if si.shape != (B, NSAMPLE):
            raise ValueError("sample_idx must be (B,500)")
    with torch.cuda.device(dev):
        if W % 32 != 0 and 2 <= B <= 256 and si is None:
            # a small batch on a frame of odd width (COCO: 427, 500, 375, 333 ...): the tiled forms - and the row engine small batches
            # take - need word-aligned rows; padding the B mask planes and the depth rows with zeros costs less than the row-linear
            # form they would otherwise run (profiles/r05/r05_odd_width_u8.txt: 8 / 64 / 256 masks of 640x427: 172 / 198 / 202 us ->
            # 111 / 102 / 152 us per call).  Above 256 planes the copy costs what it saves: pad once yourself, or hand over annotations.
            Wp = (W + 31) // 32 * 32
            m = torch.nn.functional.pad(m, (0, Wp - W))
            if ii is not None and P > B:
                # a large resident stack of planes of which this call references at most B: pad the referenced planes only
                # (P = 1000 planes of 640x427 would otherwise be copied - 1 GB - on every call)
                sel = ii.long()
                d = d.index_select(0, sel)
                if k.shape[0] > 1:
                    k = k.index_select(0, sel)
                ii, P = None, B
            d = pad_rows_f32(d, Wp)
            W = Wp
        f = InstanceFitter(B, H, W, dev)
        if B == 0:
            return f.boxes[0], f.status[0], f.aux[0]
        if k.shape[0] == 1 and P > 1:
            k = k.expand(P, 3, 3).contiguous()
        _order(stream)   # the arguments were uploaded / converted on the current stream
        out = f.run(d if P > 1 else d[0], m, k, g, si, ii, stream=stream)
        _record(stream, d, m, k, g, si, ii, f.workspace, f.boxes, f.status, f.aux)
        return out


def fit_points(clouds, ground=None, sample_idx=None, method: str = "pca", stream=None, device=None, small_clouds=None, _packed=False,
               hull_512=None):
    """estimate_bbox for a list of (N_i,3) clouds in one launch (reference src/util_3dbox.py:106-178).

    clouds: list of arrays/tensors, or a tuple (points (T,3) f64, offsets (B+1,) i64).
    small_clouds: True promises that no cloud has more than a few thousand rows to visit (LA3D_HINT_SMALL_CLOUDS: one wave per
    cloud); None = decided here when the cloud sizes are known on the host (a list of clouds, or sample_idx given).
    hull_512 (method="convex_hull"): True promises that no cloud holds more than 512 valid rows (LA3D_HINT_HULL_512: the kernel's
    small-LDS form, what the reference's 500-point clouds want); None = decided here from the cloud sizes when they are known on the
    host.  A cloud that breaks the promise comes back with status 5; without the promise the kernel holds 2048 rows per cloud.
    Returns (boxes (B,39), status (B,), aux (B,4)) on the GPU.
    """
    dev = _dev(device)
    hull512 = bool(hull_512)
    if isinstance(clouds, tuple):
        pts = _as_dev(clouds[0], torch.float64, dev)
        off = _as_dev(clouds[1], torch.int64, dev)
        if small_clouds is None:
            small_clouds = sample_idx is not None
    else:
        lens = [int(len(c)) for c in clouds]
        if hull_512 is None:
            hull512 = max(lens, default=0) <= 512      # (convex hull: the small-LDS form of the kernel, LA3D_HINT_HULL_512)
        if small_clouds is None:
            small_clouds = sample_idx is not None or max(lens, default=0) <= 4096
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        if not any(isinstance(c, torch.Tensor) for c in clouds):
            # host clouds (the reference's calling pattern): points, offsets and the other small host arguments in ONE copy
            hp = (np.concatenate([np.asarray(c, dtype=np.float64).reshape(-1, 3) for c in clouds if len(c)]) if sum(lens)
                  else np.zeros((1, 3)))
            pts, off, ground, sample_idx = _bulk(dev, (hp, torch.float64), (offs, torch.int64), (ground, torch.float64), (sample_idx, torch.int32))
        else:
            off = torch.as_tensor(offs, device=dev)
            if sum(lens):
                pts = torch.cat([_as_dev(c, torch.float64, dev).reshape(-1, 3) for c in clouds if len(c)])
            else:
                pts = torch.zeros((1, 3), dtype=torch.float64, device=dev)
    B = off.numel() - 1
    meth = {"pca": _lib.METHOD_PCA, "convex_hull": _lib.METHOD_CONVEX_HULL}.get(method)
    if meth is None:
        raise ValueError(f"Unknown method: {method}. Use 'pca' or 'convex_hull'")  # reference :151
    g = None if ground is None else _as_dev(ground, torch.float64, dev)
    si = None if sample_idx is None else _as_dev(sample_idx, torch.int32, dev)
    # one output buffer (records | aux | status): a caller that wants everything on the host reads it back in one copy
    packed = torch.empty(B * (REC + AUX) + (B + 1) // 2, dtype=torch.float64, device=dev)
    boxes = packed[:B * REC].view(B, REC)
    aux = packed[B * REC:B * (REC + AUX)].view(B, AUX)
    status = packed[B * (REC + AUX):].view(torch.int32)[:B]
    with torch.cuda.device(dev):
        check(lib.la3d_fit_points(_ptr(pts), _ptr(off), _ptr(g), _ptr(si), meth | (_lib.HINT_SMALL_CLOUDS if small_clouds else 0) | (_lib.HINT_HULL_512 if hull512 else 0), B, _ptr(boxes), _ptr(status),
                                  _ptr(aux), _stream(stream)), "la3d_fit_points")
    _record(stream, pts, off, g, si, packed)
    if _packed:
        return boxes, status, aux, packed
    return boxes, status, aux


def unproject(depth, K, R=None, t=None, out_dtype=torch.float64, stream=None, device=None) -> torch.Tensor:
    """depth (H,W) float32 -> (H,W,3) points on the GPU (reference src/util.py:52-75).  depth (P,H,W) with K (3,3) or (P,3,3):
    every frame in ONE launch -> (P,H,W,3) (``la3d_unproject_batch``; the reference's stage loops over the images)."""
    if device is None and isinstance(depth, torch.Tensor) and depth.is_cuda:
        device = depth.device
    dev = _dev(device)
    d = _as_dev(depth, torch.float32, dev)
    if d.dim() == 3:
        P, H, W = d.shape
        k = _as_dev(K, torch.float64, dev, cache=True).reshape(-1, 9)
        if k.shape[0] not in (1, P):
            raise ValueError("K must be (3,3) or (P,3,3)")
        Rt = None
        if R is not None or t is not None:
            Rt = (C.c_double * 12)(*(np.eye(3) if R is None else np.asarray(R, dtype=np.float64)).ravel(),
                                   *(np.zeros(3) if t is None else np.asarray(t, dtype=np.float64)).ravel())
        out = torch.empty((P, H, W, 3), dtype=out_dtype, device=dev)
        with torch.cuda.device(dev):
            check(lib.la3d_unproject_batch(_ptr(d), _ptr(k), 9 if k.shape[0] > 1 else 0, Rt, P, H, W, _ptr(out),
                                           int(out_dtype == torch.float64), _stream(stream)), "la3d_unproject_batch")
        _record(stream, d, k, out)
        return out
    H, W = d.shape
    K9 = (C.c_double * 9)(*np.asarray(K.cpu() if isinstance(K, torch.Tensor) else K, dtype=np.float64).ravel())
    Rt = None
    if R is not None or t is not None:
        Rm = np.eye(3) if R is None else np.asarray(R, dtype=np.float64)
        tv = np.zeros(3) if t is None else np.asarray(t, dtype=np.float64)
        Rt = (C.c_double * 12)(*Rm.ravel(), *tv.ravel())
    out = torch.empty((H, W, 3), dtype=out_dtype, device=dev)
    with torch.cuda.device(dev):
        check(lib.la3d_unproject(_ptr(d), K9, Rt, H, W, _ptr(out), int(out_dtype == torch.float64), _stream(stream)),
              "la3d_unproject")
    return out
