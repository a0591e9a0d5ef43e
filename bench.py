#!/usr/bin/env python3
"""bench.py — fitted 3D boxes/sec @640x480 on MI355X (BASELINE.json metric, config 2).

One "step" = one pass of the hot path (la3d_fit_instances: unproject + fit, fused) over one batch of
1024 synthetic instances per GPU — each instance a private 480x640 float32 depth plane ~U(0.5,10) and
a u8 mask plane holding one axis-aligned rectangle (h~U{8..300}, w~U{8..330}) — resident in HBM before
the timed region.  Full-mask mode, ground=None, K=[[500,0,320],[0,500,240],[0,0,1]].

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU, instances sharded across ranks (weak scaling: 1024 per rank, no data-path
collective); the only communication is ONE gather of every rank's (steps*B, 39) box tensor + status to
rank 0 over RCCL after the K fit steps, inside the timed region (gather_ms / value_fit_only split it).  Rank 0 prints one JSON line.

Other modes (never the headline): --config4 IMAGES = the north_star partitioning (one global metadata list -> plan_shards -> every
rank materialises and fits only its image range -> one gather; strong scaling, per-rank fit times); --end-to-end IMAGES = host-resident
scenes -> records through labelany3d_amd.fit_scenes with the pack / H2D / fit / D2H split; --rle / --poly / --subsample / --area-hint /
--config3 / --config5 / --streams as their help texts say.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W = 480, 640
ALG_BYTES_PER_BOX = H * W * (4 + 1) + 39 * 8  # private-depth layout, counted once
HBM_PEAK_GBPS = 8000.0                        # MI355X_MICROARCH.md: 8.0 TB/s spec
K640 = [[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]]


def make_inputs(B, device, seed):
    """Config-2 inputs generated on the device (depth) / from RandomState(seed) (rectangles)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    depth = torch.empty((B, H, W), dtype=torch.float32, device=device)
    depth.uniform_(0.5, 10.0, generator=g)
    rs = np.random.RandomState(seed)
    hh = rs.randint(8, 301, B)
    ww = rs.randint(8, 331, B)
    r0 = (rs.rand(B) * (H - hh + 1)).astype(np.int64)
    c0 = (rs.rand(B) * (W - ww + 1)).astype(np.int64)
    t = lambda a: torch.as_tensor(a, device=device).view(B, 1, 1)  # noqa: E731
    rows = torch.arange(H, device=device).view(1, H, 1)
    cols = torch.arange(W, device=device).view(1, 1, W)
    masks = ((rows >= t(r0)) & (rows < t(r0 + hh)) & (cols >= t(c0)) & (cols < t(c0 + ww))).to(torch.uint8).contiguous()
    K = torch.tensor(K640, dtype=torch.float64, device=device)
    return depth, masks, K, int((hh * ww).sum()), (r0, c0, hh, ww)


def make_config3(P, device, seed):
    """COCO-like scene mix (SURVEY §8d config 3 stand-in): shared depth plane + K per image, instance count
    ~Poisson(7), elliptical masks with log-uniform area 400..100k px."""
    rs = np.random.RandomState(seed)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    depth = torch.empty((P, H, W), dtype=torch.float32, device=device).uniform_(0.5, 10.0, generator=g)
    per = np.maximum(1, rs.poisson(7, P))
    img = np.repeat(np.arange(P), per).astype(np.int32)
    B = len(img)
    area = np.exp(rs.uniform(np.log(400), np.log(100000), B))
    asp = np.exp(rs.uniform(-0.7, 0.7, B))
    hh = np.clip(np.sqrt(area * asp), 8, H).astype(np.int64)
    ww = np.clip(area / hh, 8, W).astype(np.int64)
    r0 = (rs.rand(B) * (H - hh + 1)).astype(np.int64)
    c0 = (rs.rand(B) * (W - ww + 1)).astype(np.int64)
    masks = torch.empty((B, H, W), dtype=torch.uint8, device=device)
    rows = torch.arange(H, device=device, dtype=torch.float32).view(1, H, 1)
    cols = torch.arange(W, device=device, dtype=torch.float32).view(1, 1, W)
    for a in range(0, B, 2048):   # chunked: the comparison temporaries are (chunk,H,W)
        sl = slice(a, min(B, a + 2048))
        t = lambda v: torch.as_tensor(v[sl], device=device, dtype=torch.float32).view(-1, 1, 1)  # noqa: E731
        masks[sl] = ((((rows - t(r0) - t(hh) / 2) / (t(hh) / 2)) ** 2 + ((cols - t(c0) - t(ww) / 2) / (t(ww) / 2)) ** 2) < 1.0).to(torch.uint8)
    K = torch.tensor(K640, dtype=torch.float64, device=device).expand(P, 3, 3).contiguous()
    n_masked = int(masks.sum(dtype=torch.int64))
    return depth, masks, K, n_masked, torch.as_tensor(img, device=device)


def make_config5(B, device, seed):
    """BASELINE config 5 workload: private depth planes, mask areas log-uniform 8..100k px (rectangles below 400 px,
    ellipses above; aspect ratio log-uniform in e^+-0.7), position uniform in the frame."""
    rs = np.random.RandomState(seed)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    depth = torch.empty((B, H, W), dtype=torch.float32, device=device).uniform_(0.5, 10.0, generator=g)
    area = np.exp(rs.uniform(np.log(8), np.log(100000), B))
    asp = np.exp(rs.uniform(-0.7, 0.7, B))
    ell = area >= 400
    box_area = np.where(ell, area * 4 / np.pi, area)          # bounding box of the ellipse
    hh = np.clip(np.round(np.sqrt(box_area * asp)), 2, H).astype(np.int64)
    ww = np.clip(np.round(box_area / hh), 2, W).astype(np.int64)
    r0 = (rs.rand(B) * (H - hh + 1)).astype(np.int64)
    c0 = (rs.rand(B) * (W - ww + 1)).astype(np.int64)
    masks = torch.empty((B, H, W), dtype=torch.uint8, device=device)
    rows = torch.arange(H, device=device, dtype=torch.float32).view(1, H, 1)
    cols = torch.arange(W, device=device, dtype=torch.float32).view(1, 1, W)
    for a in range(0, B, 1024):   # chunked: the comparison temporaries are (chunk,H,W)
        sl = slice(a, min(B, a + 1024))
        t = lambda v: torch.as_tensor(v[sl], device=device, dtype=torch.float32).view(-1, 1, 1)  # noqa: E731
        inside = (rows >= t(r0)) & (rows < t(r0 + hh)) & (cols >= t(c0)) & (cols < t(c0 + ww))
        e = (((rows + 0.5 - t(r0) - t(hh) / 2) / (t(hh) / 2)) ** 2 + ((cols + 0.5 - t(c0) - t(ww) / 2) / (t(ww) / 2)) ** 2) <= 1.0
        masks[sl] = (inside & (e | ~torch.as_tensor(ell[sl], device=device).view(-1, 1, 1))).to(torch.uint8)
    K = torch.tensor(K640, dtype=torch.float64, device=device)
    return depth, masks, K, int(masks.sum(dtype=torch.int64)), (r0, c0, hh, ww)


def required_bytes(masks, image_index=None, num_images=0):
    """HBM bytes the path cannot avoid for these inputs: every u8 mask plane once, the 128-byte depth lines of the 32 px x
    8 row tiles that hold a mask pixel once (a depth line with no mask pixel need not be read; with a shared depth plane
    per image, a tile used by several instances of the image counts once), one 312-byte record per instance.  Computed
    from the masks themselves, so it is checkable from the inputs alone.  Returns (bytes, active tiles summed over
    instances)."""
    B, h, w = masks.shape
    tiles, union = 0, None
    hp, wp = (h + 7) // 8 * 8, (w + 31) // 32 * 32
    nt = (hp // 8) * (wp // 32)
    if image_index is not None:
        union = torch.zeros((num_images, nt), dtype=torch.float32, device=masks.device)
    for a in range(0, B, 512):
        m = masks[a:a + 512]
        if (hp, wp) != (h, w):
            m = torch.nn.functional.pad(m, (0, wp - w, 0, hp - h))
        tm = m.view(m.shape[0], hp // 8, 8, wp // 32, 32).amax(dim=(2, 4)).ne(0).reshape(m.shape[0], nt)
        tiles += int(tm.sum())
        if union is not None:
            union.index_add_(0, image_index[a:a + 512].long(), tm.float())
    depth_tiles = tiles if union is None else int((union > 0).sum())
    return B * h * w + depth_tiles * 1024 + B * 39 * 8, tiles


def measured_stream_ceiling(mask_sets, min_ms=30.0):
    """Read-only stream over the SAME mask planes the fit reads (la3d_mask_counts: 16-byte loads, nothing else), timed with HIP
    events on the launch stream: the bandwidth this box delivers to a pure reader in this run ('also report against
    a measured read-only stream ceiling').  Like the timed steps it rotates through the resident batches (launch k reads the mask
    planes of batch k % R: ~0.9 GB in rotation at R = 3, against 256 MB of Infinity Cache - reading ONE batch's 315 MB over and
    over, as rounds 1-5 did, is served partly from that cache).  Runs before the warm-up steps, for at least min_ms of GPU time."""
    import ctypes as C

    from labelany3d_amd._lib import check, lib
    if isinstance(mask_sets, torch.Tensor):
        mask_sets = [mask_sets]
    B, Hh, Ww = mask_sets[0].shape
    counts = torch.empty(B, dtype=torch.int32, device=mask_sets[0].device)
    st = torch.cuda.current_stream()
    R = len(mask_sets)
    ptrs = [C.c_void_p(m.data_ptr()) for m in mask_sets]

    def run(k):
        check(lib.la3d_mask_counts(ptrs[k % R], B, Hh, Ww, C.c_void_p(counts.data_ptr()), C.c_void_p(st.cuda_stream)),
              "la3d_mask_counts")
    for k in range(5):
        run(k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    nbytes = float(mask_sets[0].numel())
    # ~min_ms at 5 TB/s; at most 600 launches (a small --batch would otherwise issue thousands of short launches: under a rocprofv3
    # counter pass, which serialises every launch, that alone took 20 minutes at --batch 64)
    iters = min(600, max(20, int(min_ms * 1e-3 * 5.0e12 / nbytes)))
    best, total_ms, reps = 0.0, 0.0, 0
    while total_ms < min_ms and reps < 8:
        torch.cuda.synchronize()
        e0.record(st)
        for k in range(iters):
            run(k)
        e1.record(st)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        total_ms += ms
        reps += 1
        best = max(best, nbytes * iters / (ms * 1e-3) / 1e9)
    return best, int(counts.sum()), 5 + reps * iters


def traffic_mode_key(args, B):
    """Key of this run's workload in profiles/traffic_per_launch.json (PMC-measured HBM bytes per step, profiles/r03_run.sh)."""
    if args.config3:
        key = f"config3_{args.config3}"
    elif args.config5:
        key = "config5" if B == 1024 else f"config5_B{B}"
    else:
        key = "config2" if B == 1024 else f"config2_B{B}"
    for flag in ("rle", "poly", "subsample", "area_hint", "ground"):
        if getattr(args, flag):
            key += "_" + flag
    return key


def kernel_source_sha256():
    """Hash of the sources the FIT kernels are compiled from (engines, walks, stages, shared device code, dispatcher): stamps the PMC
    traffic / VALU figures so that a stale constant is detectable.  The other kernels of the library (point clouds, mask decode /
    statistics, consumers) are not part of any bench step and do not enter."""
    import hashlib
    h = hashlib.sha256()
    for f in ("la3d.hip", "la3d_instance.hip", "la3d_band.hip", "la3d_rows.hip", "la3d_split.hip", "la3d_walks.hpp", "la3d_stages.hpp",
              "la3d_engines.hpp", "la3d_device.hpp", "la3d_poly.hpp"):
        h.update(open(os.path.join(ROOT, "labelany3d_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def build_identity():
    """Which binary produced this line: the hash of the kernel sources and of the compile command embedded in the loaded libla3d.so
    at build time (la3d_build_info) next to the hash of the sources in the tree."""
    try:
        from labelany3d_amd import _build
        from labelany3d_amd._lib import LIB, lib
        info = lib.la3d_build_info().decode()
        _, src, cmd = info.split(":")
        tree = _build.source_sha256()
        return {"lib": os.path.relpath(LIB, ROOT), "lib_sources_sha256": src, "lib_compile_cmd_sha256": cmd, "tree_sources_sha256": tree,
                "lib_built_from_tree": src == tree, "kernel_source_sha256": kernel_source_sha256()}
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


def rect_rle(rects):
    """COCO run lengths (column-major, zeros first) of the same rectangles: the --rle input format."""
    r0, c0, hh, ww = rects
    counts, offs = [], [0]
    for a, b, h, w in zip(r0, c0, hh, ww):
        c = [int(b * H + a)] + [int(h), int(H - h)] * (int(w) - 1) + [int(h), int((W - b - w) * H + (H - a - h))]
        counts += c
        offs.append(len(counts))
    return np.asarray(counts, np.int32), np.asarray(offs, np.int64)


def cpu_baseline_all_cores(d, m, per_worker=12, timeout_s=90.0):
    """The same NumPy path on every host core at once: one single-threaded worker process per core (plain
    subprocesses of oracle/cpu_worker.py, started before the clock and released together), each fitting `per_worker`
    instances of the sample.  Returns (boxes/s, workers, seconds) or None if anything goes wrong."""
    import shutil
    import subprocess
    import tempfile

    nproc = max(1, min(os.cpu_count() or 1, 128))
    tmp = tempfile.mkdtemp(prefix="la3d_cpu_")
    procs = []
    try:
        np.save(os.path.join(tmp, "d.npy"), d)
        np.save(os.path.join(tmp, "m.npy"), m.astype(np.uint8))
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1", PYTHONPATH=ROOT)
        for w in range(nproc):
            procs.append(subprocess.Popen([sys.executable, "-m", "oracle.cpu_worker", os.path.join(tmp, "d.npy"),
                                           os.path.join(tmp, "m.npy"), str(w * per_worker), str(per_worker)],
                                          stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                          env=env, cwd=ROOT, text=True))
        deadline = time.time() + timeout_s
        for p in procs:
            line = p.stdout.readline()
            if not line.startswith("ready") or time.time() > deadline:
                return None
        t0 = time.perf_counter()
        for p in procs:
            p.stdin.write("go\n")
            p.stdin.flush()
        for p in procs:
            line = p.stdout.readline()
            if not line.startswith("done") or time.time() > deadline:
                return None
        dt = time.perf_counter() - t0
        return nproc * per_worker / dt, nproc, dt
    except Exception:  # noqa: BLE001 - a baseline leg must never take the bench down
        return None
    finally:
        for p in procs:
            try:
                p.kill()
            except Exception:  # noqa: BLE001
                pass
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_baseline(depth, masks, budget_s=8.0, max_inst=256):
    """Reference-equivalent NumPy path (oracle/la3d_oracle.py, verified against the reference's own
    outputs in tests/) timed on this box's host cores on a bounded sample of the same workload: per
    instance depth_to_points of its private plane (reference array ops, src/util.py:52-75) + pts[mask] +
    estimate_bbox (src/util_3dbox.py:106-178, full-mask mode).  Two legs: one thread, and one worker process per
    host core (up to 128); `value` is the all-cores rate, `cores` the workers that produced it."""
    from oracle import la3d_oracle as O

    try:
        torch.set_num_threads(1)
    except Exception:  # noqa: BLE001
        pass
    K = np.array(K640)
    n_take = min(max_inst, depth.shape[0])
    d = depth[:n_take].cpu().numpy()
    m = masks[:n_take].cpu().numpy().astype(bool)
    O.fit_instance(d[0], m[0], K, refstyle=True)  # warm caches / first-touch
    t0 = time.perf_counter()
    done = 0
    for i in range(n_take):
        O.fit_instance(d[i], m[i], K, refstyle=True)
        done += 1
        if time.perf_counter() - t0 > budget_s and done >= 32:
            break
    dt = time.perf_counter() - t0
    single = done / dt
    out = {"value": single, "unit": "boxes/s", "cores": 1, "kind": "port",
           "single_thread_value": single,
           "sample": f"first {done} instances of the same config-2 batch (private 480x640 planes), "
                     f"NumPy oracle single-thread, {dt:.1f} s; host has {os.cpu_count()} logical cores"}
    multi = cpu_baseline_all_cores(d, m)
    if multi is not None:
        rate, nproc, mdt = multi
        out.update({"value": rate, "cores": nproc,
                    "sample": out["sample"] + f"; all-cores leg: {nproc} single-threaded worker processes x 12 instances of "
                                              f"the same sample, released together, {mdt:.1f} s"})
    return out


def build_inputs(args, B, device, seed, aux_seed):
    """One resident input batch of the workload the flags name: depth, K, the masks as u8 planes (`all_masks`: always built - the
    byte model and the stream ceiling read them) and, per flag, the run lengths / polygon parts the fit is fed instead (`masks` is
    then None), the drawn sample indices, the area hints, the ground planes."""
    x = {"image_index": None, "rle": None, "poly": None, "sample_idx": None, "area_hint": None, "ground": None}
    if args.config3:
        depth, masks, K, n_masked, x["image_index"] = make_config3(args.config3, device, seed)
        rects = None
    elif args.config5:
        depth, masks, K, n_masked, rects = make_config5(B, device, seed)
    else:
        depth, masks, K, n_masked, rects = make_inputs(B, device, seed)
    B = masks.shape[0]
    x.update(depth=depth, K=K, all_masks=masks, masks=masks, n_masked=n_masked)
    if args.subsample:
        from labelany3d_amd import draw_sample_idx
        counts = masks.reshape(B, -1).sum(1, dtype=torch.int64)
        x["sample_idx"] = torch.as_tensor(draw_sample_idx(counts, np.random.RandomState(99 + aux_seed)), device=device)
    if args.rle:
        if rects is None:
            raise SystemExit("--rle needs the rectangle masks of the default workload (not --config3)")
        rc_np, ro_np = rect_rle(rects)
        x["rle"] = (torch.as_tensor(rc_np, device=device), torch.as_tensor(ro_np, device=device))
        x["masks"] = None
    if args.poly:
        if rects is None:
            raise SystemExit("--poly needs the rectangle masks of the default workload (not --config3 / --config5)")
        from labelany3d_amd import pack_polygons
        r0_, c0_, hh_, ww_ = rects
        segs = [[[int(b), int(a), int(b + w - 1), int(a), int(b + w - 1), int(a + h - 1), int(b), int(a + h - 1)]]
                for a, b, h, w in zip(r0_, c0_, hh_, ww_)]
        pxy, pro, pir, _, _ = pack_polygons(segs, H, W)
        x["poly"] = tuple(torch.as_tensor(t, device=device) for t in (pxy, pro, pir))
        x["masks"] = None
    if args.area_hint:
        if args.config3 or args.subsample:
            raise SystemExit("--area-hint: private depth planes, full-mask mode only")
        x["area_hint"] = masks.reshape(B, -1).sum(1, dtype=torch.int32)     # what the annotation's "area" field holds
    if args.ground:     # near-level cameras: the plane normal within a few degrees of -y, as the reference's canonical upright vectors are
        gr = np.random.RandomState(77 + aux_seed)
        x["ground"] = torch.as_tensor(np.array([[0.02, -0.97, 0.1, 1.2]] * B) + 0.03 * gr.randn(B, 4), device=device)
    return x


def batch_required_bytes(args, x):
    """(required bytes, active tiles) of one input batch: bytes this input cannot be fitted without - every mask plane once + the
    128-B depth lines of the 32x8 tiles that hold a mask pixel once + the records (computed from the masks; with run-length /
    polygon input the mask term is their arrays; reference-subsample mode: one 64-B sector per drawn point)."""
    masks = x["all_masks"]
    B = masks.shape[0]
    req, tiles = required_bytes(masks, x.get("image_index"), args.config3)
    if x.get("rle") is not None:
        req += int(x["rle"][0].numel()) * 4 - B * H * W
    if x.get("poly") is not None:
        pxy, pro, pir = x["poly"]
        req += int(pxy.numel()) * 4 + int(pro.numel() + pir.numel()) * 8 - B * H * W
    if args.subsample:   # mask planes + one 64-B sector per drawn point (masks of <= 500 px: their tiles, as above) + records
        cnt = masks.reshape(B, -1).sum(1, dtype=torch.int64)
        big = cnt > 500
        tiles_small = required_bytes(masks[~big], None, 0)[1] if int((~big).sum()) else 0
        req = B * H * W + int(big.sum()) * 500 * 64 + tiles_small * 1024 + B * 39 * 8
    return req, tiles


class StepRunner:
    """One step = ONE la3d_fit_instances_ex call on a pre-built argument block (the steady-state call is a pure enqueue: no
    allocation, no Python work beyond the ctypes call).  `inputs` is a list of resident input batches (dicts: depth, K and exactly
    one of masks / rle / poly, optional image_index / sample_idx / area_hint / ground); step k reads batch k % len(inputs) - with
    more than one batch no two consecutive steps read the same bytes (--rotate).  The scheduling of a call travels in the block
    (opt_launch_order), never in process state."""

    def __init__(self, fitter, inputs, one_slot=False):
        self.f, self.inputs, self.one_slot = fitter, list(inputs), one_slot
        self.blocks = {}
        import ctypes as C

        from labelany3d_amd._lib import check, lib
        self._fit, self._byref, self._check = lib.la3d_fit_instances_ex, C.byref, check

    def __call__(self, slot=0, stream=None, ws_slot=0, launch_order=None, batch=None):
        if self.one_slot:
            slot = 0
        if batch is None:
            batch = slot % len(self.inputs)
        key = (slot, ws_slot, stream.cuda_stream, launch_order, batch)
        a = self.blocks.get(key)
        if a is not None and not self.dry:      # the steady-state call: one dictionary lookup, one foreign call
            rc = self._fit(self._byref(a))
            if rc:
                self._check(rc, "la3d_fit_instances_ex")
            return
        import ctypes as C

        from labelany3d_amd import options
        from labelany3d_amd._lib import FitArgs, check, lib
        if a is None:
            f, x = self.f, self.inputs[batch]
            d, k = x["depth"], x["K"]
            ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
            a = FitArgs()
            a.struct_size = C.sizeof(FitArgs)
            a.B, a.H, a.W = f.B, f.H, f.W
            a.depth, a.depth_plane_stride = d.data_ptr(), (f.H * f.W if (d.dim() == 3 and d.shape[0] > 1) else 0)
            a.image_index = ptr(x.get("image_index"))
            if x.get("rle") is not None:
                a.rle_counts, a.rle_offsets = x["rle"][0].data_ptr(), x["rle"][1].data_ptr()
            elif x.get("poly") is not None:
                a.poly_xy, a.ring_offsets, a.inst_rings = (t.data_ptr() for t in x["poly"])
            else:
                a.mask = x["masks"].data_ptr()
            a.K, a.k_stride = k.data_ptr(), (9 if (k.dim() == 3 and k.shape[0] > 1) else 0)
            a.filter_boundary = -1
            a.ground = ptr(x.get("ground"))
            a.sample_idx = ptr(x.get("sample_idx"))
            a.area_hint = ptr(x.get("area_hint"))
            a.out, a.status, a.aux = f.boxes[slot].data_ptr(), f.status[slot].data_ptr(), f.aux[slot].data_ptr()
            a.workspace, a.stream = f.workspace[ws_slot].data_ptr(), stream.cuda_stream
            a.opt_launch_order = options.ORDER[launch_order]
            self.blocks[key] = a
        if self.dry:
            return
        check(lib.la3d_fit_instances_ex(C.byref(a)), "la3d_fit_instances_ex")

    dry = False

    def prepare(self, calls):
        """Build the argument blocks of the given calls ahead of time (same keyword arguments as __call__): the timed loop is then
        nothing but ctypes calls.  (A step of K = 1000 writes into its own output slot, hence its own block: built inside the loop,
        ~30 us of Python per step, the blocks made every kernel shorter than ~60 us - run lengths, polygons, B = 256 - host-bound.)"""
        self.dry = True
        try:
            for kw in calls:
                self(**kw)
        finally:
            self.dry = False


# ------------------------------------------------------------------------------------------------------------------------
# --config4 IMAGES: the north_star partitioning - ONE global metadata list, per-image shards, one gather
# ------------------------------------------------------------------------------------------------------------------------
def config4_metadata(P, seed):
    """What every rank knows about the whole job WITHOUT holding any of its tensors (BASELINE config 4: COCO-train sharded per
    image): per image ~Poisson(7) instances, per instance an ellipse (centre, half axes) of log-uniform area 400..100k px.
    Host NumPy, identical on every rank."""
    rs = np.random.RandomState(seed)
    per = np.maximum(1, rs.poisson(7, P))
    img = np.repeat(np.arange(P), per).astype(np.int64)
    B = len(img)
    area = np.exp(rs.uniform(np.log(400), np.log(100000), B))
    asp = np.exp(rs.uniform(-0.7, 0.7, B))
    hh = np.clip(np.sqrt(area * asp), 8, H)
    ww = np.clip(area / hh, 8, W)
    r0 = rs.rand(B) * (H - hh)
    c0 = rs.rand(B) * (W - ww)
    return {"img": img, "hh": hh, "ww": ww, "r0": r0, "c0": c0, "area": np.pi / 4 * hh * ww, "P": P, "B": B, "seed": seed}


def config4_annotations(meta, kind):
    """The metadata's ellipses as COCO-style annotation dicts, identical on every rank: kind "poly" = one 24-vertex outline per
    instance (what the reference's COCONut converter writes, src/download_coconut.py:178-199), "rle" = uncompressed column-major run
    lengths of the same ellipse (its :167-176).  ``area`` is the annotation's own field: what plan_shards balances by."""
    anns = []
    ang = np.linspace(0, 2 * np.pi, 24, endpoint=False)
    ca, sa = np.cos(ang), np.sin(ang)
    for n in range(meta["B"]):
        hh, ww, r0, c0 = meta["hh"][n], meta["ww"][n], meta["r0"][n], meta["c0"][n]
        cy, cx = r0 + hh / 2, c0 + ww / 2
        if kind == "poly":
            xs, ys = cx + ww / 2 * ca, cy + hh / 2 * sa
            seg = [np.stack([xs, ys], 1).reshape(-1).tolist()]
        else:
            cols = np.arange(W)
            t = 1.0 - ((cols - cx) / (ww / 2)) ** 2
            half = np.sqrt(np.clip(t, 0, None)) * hh / 2
            lo = np.clip(np.ceil(cy - half), 0, H).astype(np.int64)
            hi = np.clip(np.floor(cy + half) + 1, 0, H).astype(np.int64)
            on = np.nonzero((t > 0) & (hi > lo))[0]
            st, en = on * H + lo[on], on * H + hi[on]            # column-major pixel ranges of the ones-runs, one per column
            cnt = np.empty(2 * len(on) + 1, np.int64)
            cnt[0:-1:2] = st - np.concatenate([[0], en[:-1]])      # zeros before every ones-run
            cnt[1::2] = en - st
            cnt[-1] = H * W - (en[-1] if len(on) else 0)
            seg = {"size": [H, W], "counts": cnt.tolist()}
        anns.append({"id": n, "image_id": int(meta["img"][n]), "category_id": 1, "iscrowd": 0, "bbox": [float(c0), float(r0), float(ww), float(hh)],
                     "area": float(meta["area"][n]), "segmentation": seg})
    return anns


def config4_materialize(meta, shard, device, with_masks=True):
    """ONLY this rank's tensors: the depth planes of images [img_lo, img_hi) - plane i is a function of (seed, i), so any rank
    would build the same plane - and the u8 masks of instances [inst_lo, inst_hi).  Returns (depth, masks, K, None, None), the
    load_fn contract of fit_instances_sharded."""
    ilo, ihi, nlo, nhi = shard
    depth = torch.empty((max(ihi - ilo, 1), H, W), dtype=torch.float32, device=device)
    g = torch.Generator(device=device)
    for i in range(ilo, ihi):
        g.manual_seed(meta["seed"] * 1000003 + i)
        depth[i - ilo].uniform_(0.5, 10.0, generator=g)
    n = nhi - nlo
    K = torch.tensor(K640, dtype=torch.float64, device=device)
    if not with_masks:   # annotation formats (--poly / --rle): no u8 plane exists on any rank
        return depth[:ihi - ilo], None, K, None, None
    masks = torch.empty((max(n, 1), H, W), dtype=torch.uint8, device=device)
    rows = torch.arange(H, device=device, dtype=torch.float32).view(1, H, 1)
    cols = torch.arange(W, device=device, dtype=torch.float32).view(1, 1, W)
    for a in range(0, n, 1024):
        sl = slice(nlo + a, min(nhi, nlo + a + 1024))
        t = lambda key: torch.as_tensor(meta[key][sl], device=device, dtype=torch.float32).view(-1, 1, 1)  # noqa: E731
        masks[a:a + (sl.stop - sl.start)] = ((((rows - t("r0") - t("hh") / 2) / (t("hh") / 2)) ** 2 +
                                              ((cols - t("c0") - t("ww") / 2) / (t("ww") / 2)) ** 2) < 1.0).to(torch.uint8)
    K = torch.tensor(K640, dtype=torch.float64, device=device)
    return depth[:ihi - ilo], masks[:n], K, None, None


def run_config4(args, dist, rank, world, device, red_dev):
    """bench.py --config4 IMAGES [--gpus N]: every rank derives the SAME plan from the metadata (plan_shards: contiguous image
    ranges balanced by what the fit moves per image), materialises only its own range, fits it in one call
    (fit_instances_sharded) and joins the ONE gather of (n_i, 39) records + status.  One step = the whole job; `value` =
    all instances / max over ranks of (fit + gather) time; per-rank fit times and the gather's time stand beside it."""
    from labelany3d_amd.shard import fit_instances_sharded, plan_shards

    meta = config4_metadata(args.config4, 1234)
    P, B = meta["P"], meta["B"]
    dist_ = dist
    plan = plan_shards(meta["img"], P, world, areas=meta["area"], frame_pixels=H * W)
    loaded = {}
    ann_mode = args.poly or args.rle

    def load_fn(sh):   # materialised once, outside the timed steps (inputs resident in HBM, like the headline)
        if "t" not in loaded:
            loaded["t"] = config4_materialize(meta, sh, device, with_masks=not ann_mode)
        return loaded["t"]

    load_fn(plan[rank])
    if ann_mode:
        # round 5: the same job on the reference's ANNOTATION formats (src/util.py:336-383): the global metadata list becomes a list of
        # COCO-style annotation dicts (24-vertex polygon outlines of the ellipses, or run lengths), every rank packs and uploads ONLY
        # the segmentations of its own image range (outside the timed jobs, like the depth planes) and fits them with
        # fit_instances_poly / fit_instances_rle: no u8 plane exists on any rank
        from labelany3d_amd import fit_instances_ex, pack_polygons, pack_rle
        from labelany3d_amd.shard import fit_annotations_sharded
        anns = config4_annotations(meta, "rle" if args.rle else "poly")
        packed = {}

        def depth_loader(sh):
            d, _, k, _, _ = load_fn(sh)
            return d, k

        def ann_fit(annotations, image_size, d, K, ground=None, image_index=None, filter=None):
            if "p" not in packed:   # first (warm-up) job: pack + upload this rank's segmentations once
                segs = [a["segmentation"] for a in annotations]
                if args.rle:
                    c, o, hh, ww = pack_rle(segs)
                    packed["p"] = dict(rles=(torch.as_tensor(c, device=device), torch.as_tensor(o, device=device), hh, ww))
                else:
                    xy, ro, ir, hh, ww = pack_polygons(segs, H, W)
                    packed["p"] = dict(polys=tuple(torch.as_tensor(x, device=device) for x in (xy, ro, ir)) + (hh, ww))
                packed["ii"] = torch.as_tensor(image_index, device=device)
                packed["hint"] = torch.as_tensor(np.asarray([a["area"] for a in annotations]).astype(np.int32), device=device)
            r = fit_instances_ex(d, K, image_index=packed["ii"], area_hint=packed["hint"], device=device, **packed["p"])
            return r["boxes"], r["status"]
    best = None
    for it in range(args.warmup_jobs + args.jobs):
        torch.cuda.synchronize()
        tm = {}
        if ann_mode:
            if dist_ is not None:
                dist_.barrier()
            t0 = time.perf_counter()
            if dist_ is None:
                sh = plan[0]
                d, k = depth_loader(sh)
                b, st = ann_fit(anns[sh.inst_lo:sh.inst_hi], (W, H), d, k, image_index=(meta["img"][sh.inst_lo:sh.inst_hi] - sh.img_lo).astype(np.int32))
                torch.cuda.synchronize()
                tm.update(fit_s=time.perf_counter() - t0, gather_s=0.0)
                out = (b, st, [B])
            else:
                out = fit_annotations_sharded(anns, (W, H), meta["img"], P, depth_loader, areas=meta["area"], fit_fn=ann_fit, timings=tm)
                torch.cuda.synchronize()
                dist_.barrier()
        elif dist_ is None:   # one process: the same plan / load / fit, no collective to run
            from labelany3d_amd import fit_instances
            sh = plan[0]
            d, m, k, _, _ = load_fn(sh)
            t0 = time.perf_counter()
            b, st, _ = fit_instances(d, m, k, image_index=(meta["img"][sh.inst_lo:sh.inst_hi] - sh.img_lo).astype(np.int32))
            torch.cuda.synchronize()
            tm.update(fit_s=time.perf_counter() - t0, gather_s=0.0)
            out = (b, st, [B])
        else:
            dist_.barrier()
            t0 = time.perf_counter()
            out = fit_instances_sharded((P, H, W), None, None, meta["img"], areas=meta["area"], load_fn=load_fn, timings=tm)
            torch.cuda.synchronize()
            dist_.barrier()
        tm["job_s"] = time.perf_counter() - t0
        if it >= args.warmup_jobs and (best is None or tm["job_s"] < best["job_s"]):
            best = tm
            best_out = out
    vals = torch.tensor([best["fit_s"], best["gather_s"], best["job_s"]], dtype=torch.float64, device=red_dev)
    allv = [torch.zeros_like(vals) for _ in range(world)]
    if dist_ is None:
        allv = [vals]
    else:
        dist_.all_gather(allv, vals)
    if rank == 0:
        boxes, status, counts = best_out
        assert boxes.shape == (B, 39) and int((status == 0).sum()) == B, (boxes.shape, int((status != 0).sum()))
        mask_input = "u8 planes" if not ann_mode else ("COCO run lengths" if args.rle else "polygon parts (24-vertex outlines)")
        if args.dump:   # tests: the gathered records in global instance order
            np.save(args.dump, boxes.cpu().numpy())
        fit = [float(v[0]) for v in allv]
        gat = [float(v[1]) for v in allv]
        job = max(float(v[2]) for v in allv)
        print(json.dumps({
            "metric": "fitted 3D boxes/sec @640x480", "value": B / job, "unit": "boxes/s", "n_gpus": world, "ranks": getattr(args, "ranks", None), "steps": args.jobs,
            "warmup": args.warmup_jobs, "ms_per_step": job * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE config 4 partitioning: {P} images with a shared 480x640 depth plane each, {B} instances "
                                   f"(~Poisson(7) per image, elliptical masks as {mask_input}, log-uniform area 400..100k px), ONE global metadata list, "
                                   "plan_shards -> every rank materialises and fits only its contiguous image range -> one gather of (n_i,39) records",
                       "mask_input": mask_input, "images": P, "instances": B, "instances_per_rank": [p[3] - p[2] for p in plan], "images_per_rank": [p[1] - p[0] for p in plan],
                       "sharding": "per image, cost-balanced contiguous ranges (reference --start_index/--end_index, whole.py:25-27,42)"},
            "per_rank_fit_ms": [f * 1e3 for f in fit], "fit_ms_max": max(fit) * 1e3, "fit_ms_min": min(fit) * 1e3,
            "imbalance_max_over_mean": max(fit) / (sum(fit) / len(fit)), "gather_ms": max(gat) * 1e3,
            "value_fit_only": B / max(fit),
            "note": "one step = the whole job (best of --jobs); value = instances / (max over ranks of the wall time between two barriers "
                    "around fit + gather); inputs resident in HBM before the clock starts; per_rank_fit_ms shows the balance of the plan",
        }), flush=True)


# ------------------------------------------------------------------------------------------------------------------------
# --end-to-end: host-resident scenes -> records on the host (labelany3d_amd.fit_scenes.ScenePipeline)
# ------------------------------------------------------------------------------------------------------------------------
def pinned_h2d_GBps(device, nbytes=256 << 20):
    src = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
    dst = torch.empty(nbytes, dtype=torch.uint8, device=device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(4):
        dst.copy_(src, non_blocking=True)
    e1.record()
    torch.cuda.synchronize()
    return 4 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9


def run_end_to_end(args, device):
    """bench.py --end-to-end IMAGES: what a caller holding HOST data gets (SURVEY section 7 'report end-to-end separately'): scenes
    as in-memory dicts (pageable float32 depth planes, COCO-style polygon / run-length annotations) through the real-data
    pipeline - copy into pinned memory, pack the segmentations, upload on a copy stream, decode + filter + fit in one launch per
    segmentation kind and batch, download the records - to Python-side record lists.  Never the headline: the depth planes
    alone (1.23 MB per image) bound it by the host link."""
    from labelany3d_amd.fit_scenes import ScenePipeline, synthetic_scenes

    n = args.end_to_end
    scenes, data = synthetic_scenes(n, seed=3)
    link = pinned_h2d_GBps(device)
    warm = {}
    list(ScenePipeline(device=device, batch_images=args.batch_images, write=False, timings=warm).run(scenes[:min(n, 2 * args.batch_images)]))
    best = None
    for _ in range(3):
        tm = {}
        pipe = ScenePipeline(device=device, batch_images=args.batch_images, write=False, timings=tm)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nb = sum(len(recs) for _, recs in pipe.run(scenes))
        torch.cuda.synchronize()
        tm["total_s"] = time.perf_counter() - t0
        tm["boxes"] = nb
        if best is None or tm["total_s"] < best["total_s"]:
            best = tm
    t = best
    bytes_per_image = t["h2d_bytes"] / t["images"]
    ceiling_images = link * 1e9 / bytes_per_image
    print(json.dumps({
        "metric": "fitted 3D boxes/sec @640x480, host-resident scenes -> records on the host", "value": t["boxes"] / t["total_s"],
        "unit": "boxes/s", "n_gpus": 1, "higher_is_better": True, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{n} scenes (one 480x640 f32 depth plane each, ~7 annotations per image: polygons (24-vertex outlines, some in "
                               "two parts) and uncompressed COCO run lengths; crowd / tiny / border-touching ones dropped by the reference's "
                               "keep rule inside the fit launch), batches of {0} images".format(args.batch_images),
                   "images": int(t["images"]), "annotations_submitted": int(t["instances"]), "boxes_kept": int(t["boxes"]),
                   "batches": int(t["batches"])},
        "images_per_s": t["images"] / t["total_s"], "annotations_per_s": t["instances"] / t["total_s"],
        "split_s": {"copy_to_pinned": t["load_s"], "pack_segmentations": t["pack_s"], "h2d_on_copy_stream": t["h2d_s"],
                    "fit_and_d2h_on_compute_stream": t["fit_s"], "issue_fit_calls": t.get("fit_issue_s", 0.0), "records_to_python": t["write_s"],
                    "wall": t["total_s"]},
        "h2d_bytes": t["h2d_bytes"], "h2d_GBps_achieved_while_copying": t["h2d_bytes"] / t["h2d_s"] / 1e9 if t["h2d_s"] else None,
        "host_link": {"pinned_h2d_GBps": link, "bytes_per_image": bytes_per_image, "ceiling_images_per_s": ceiling_images,
                      "ceiling_boxes_per_s": ceiling_images * t["boxes"] / t["images"],
                      "frac_of_ceiling": (t["images"] / t["total_s"]) / ceiling_images},
        "note": "stages overlap (loader thread + copy stream run one batch ahead of the fit), so split_s adds up to more than wall; "
                "copy_to_pinned and pack run on host threads (Python / NumPy); the ceiling is the measured pinned host-to-device rate "
                "divided by the bytes one image needs on the device",
    }), flush=True)


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-run this command line as N ranks of one node under torch.distributed.run
    (one process per GPU, LOCAL_RANK = the rank's device; the rendezvous on 127.0.0.1 and a free port) and return its exit code.
    The per-process index ranges of the reference (`src/batch_scripts/whole.py:25-27,42`) are what the ranks stand for.  Fails
    loudly - no silent one-GPU measurement - when the node has fewer than N GPUs for the RCCL backend."""
    import socket
    import subprocess

    backend = os.environ.get("LA3D_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if backend == "nccl" and have < n:
        print(f"bench.py: --gpus {n} needs {n} GPUs for its {n} RCCL ranks, this node shows {have}; nothing was measured "
              "(LA3D_BENCH_BACKEND=gloo is the functional dry run of several ranks on one GPU)", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LA3D_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def rank_identity(dist, rank, local, device):
    """What proves which ranks and devices a line was measured on: per rank its device index, name, PCI bus id and uuid
    (gathered once, before the timed region)."""
    p = torch.cuda.get_device_properties(device)
    bus = None
    if hasattr(p, "pci_bus_id"):
        bus = "%04x:%02x:%02x" % (int(getattr(p, "pci_domain_id", 0)), int(p.pci_bus_id), int(getattr(p, "pci_device_id", 0)))
    me = {"rank": rank, "local_rank": local, "device": device.index, "name": p.name, "pci_bus_id": bus,
          "uuid": str(getattr(p, "uuid", "")) or None, "pid": os.getpid()}
    if dist is None:
        return [me]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, me)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=1024, help="instances per GPU per step (config 2: 1024)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-steady", action="store_true", help="skip the steady-state secondary figure (counter passes: profiles/run_profile.sh "
                                                             "divides the counters by the launches of --steps + --warmup)")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the secondary two-stream pipelined measurement")
    ap.add_argument("--no-same-batch", action="store_true", help="skip the secondary loop of K steps that all read batch 0 (counter passes divide "
                                                                 "by the launches of --steps + --warmup)")
    ap.add_argument("--rle", action="store_true",
                    help="feed the masks as COCO run lengths (la3d_fit_instances_rle) instead of u8 planes; NOT the "
                         "BASELINE config-2 input format, reported for the mask-ingestion row only")
    ap.add_argument("--poly", action="store_true",
                    help="feed the masks as polygon parts (la3d_fit_instances_poly: the reference's COCONut annotation format, "
                         "rasterised with cv2.fillPoly's rule inside the fit kernel) instead of u8 planes: the rectangles as 4-vertex "
                         "rings; NOT the BASELINE config-2 input format, reported for the mask-ingestion row only")
    ap.add_argument("--area-hint", action="store_true",
                    help="secondary mode: hand the mask areas to the fit (la3d_fit_args::area_hint - what a caller holding the annotation "
                         "metadata or a preceding filter's statistics can do): the launch order then needs no estimate pass over the masks")
    ap.add_argument("--ground", action="store_true",
                    help="secondary workload: one ground plane per instance (the reference's harness always passes one, "
                         "src/util_3dbox.py:273-278): the fit rotates into the ground frame first - the two-pass form of the kernel")
    ap.add_argument("--subsample", action="store_true",
                    help="secondary mode: the reference's own semantics for masks above 500 px - 500 points drawn with replacement "
                         "(np.random.randint, src/util_3dbox.py:123-125; indices drawn once outside the timed region, as the "
                         "reference's RNG calls are host work) instead of the full mask")
    ap.add_argument("--config3", type=int, default=0, metavar="IMAGES",
                    help="secondary mode: BASELINE config-3 stand-in — IMAGES shared depth planes, ~Poisson(7) instances per "
                         "image with log-uniform mask areas 400..100k px, all instances in ONE call per step")
    ap.add_argument("--config5", action="store_true",
                    help="secondary mode: BASELINE config-5 workload (mask areas log-uniform 8..100k px, private depth)")
    ap.add_argument("--config4", type=int, default=0, metavar="IMAGES",
                    help="the north_star partitioning at N ranks: ONE global metadata list of IMAGES shared depth planes (~7 instances each), "
                         "plan_shards -> every rank materialises and fits only its image range -> one gather (strong scaling; one step = the job)")
    ap.add_argument("--jobs", type=int, default=3, help="--config4: timed repetitions of the whole job (best is reported)")
    ap.add_argument("--warmup-jobs", type=int, default=1)
    ap.add_argument("--dump", default=None, help="--config4: np.save the gathered (B,39) records here (tests)")
    ap.add_argument("--end-to-end", type=int, default=0, metavar="IMAGES",
                    help="host-resident scenes -> records on the host through labelany3d_amd.fit_scenes (never the headline): boxes/s with the "
                         "split pack / H2D / fit / D2H and the host-link ceiling")
    ap.add_argument("--batch-images", type=int, default=256, help="--end-to-end: images per fit launch")
    ap.add_argument("--rotate", type=int, default=3, metavar="R",
                    help="distinct resident input batches the timed steps rotate through (step k reads batch k %% R; default 3 = 5.6 GB at "
                         "B = 1024; 1 = every step reads the same batch, the protocol of rounds 1-5; falls back to 1 when memory is short)")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams the independent steps are issued on round-robin (1 = strictly serial steps)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N` (the shape of the driver's N = 1 command): become the launcher of N ranks
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        # a torchrun line whose --nproc-per-node disagrees with --gpus would record an n_gpus nobody asked for
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                         f"(or plain `python bench.py --gpus {args.gpus}`, which spawns its own ranks)")
    if world > 1 and os.environ.get("LA3D_BENCH_BACKEND", "nccl") == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: {world} RCCL ranks need {world} GPUs, this node shows {torch.cuda.device_count()} "
                         "(LA3D_BENCH_BACKEND=gloo is the functional dry run of several ranks on one GPU)")
    # LA3D_BENCH_FORCE_DIST=1: initialise torch.distributed even for one rank, so that the RCCL branch (device-tensor gather,
    # timing all-reduces) can be executed on a single-GPU box (tests/test_gpu_shard.py)
    if world > 1 or os.environ.get("LA3D_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # LA3D_BENCH_BACKEND=gloo: functional dry run of the multi-rank path with several ranks on ONE GPU (RCCL refuses that);
        # the ranks then share device 0 and the three timing reductions go through the host.  Never a performance figure.
        backend = os.environ.get("LA3D_BENCH_BACKEND", "nccl")
        local = local % max(torch.cuda.device_count(), 1) if backend != "nccl" else local
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    else:
        dist = None
        torch.cuda.set_device(0)
    device = torch.device("cuda", torch.cuda.current_device())
    red_dev = device if (dist is None or dist.get_backend() == "nccl") else torch.device("cpu")   # where the timing scalars are reduced
    ranks = {"backend": (dist.get_backend() if dist is not None else None), "world_size": (dist.get_world_size() if dist is not None else 1),
             "rccl_ranks": (dist.get_world_size() if dist is not None and dist.get_backend() == "nccl" else 0),
             "self_launched": os.environ.get("LA3D_BENCH_SELF_LAUNCHED") == "1", "devices": rank_identity(dist, rank, local, device)}
    args.ranks = ranks

    if args.config4:
        run_config4(args, dist, rank, world, device, red_dev)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.end_to_end:
        if world != 1:
            raise SystemExit("--end-to-end is a single-GPU measurement")
        run_end_to_end(args, device)
        return

    from labelany3d_amd import InstanceFitter
    from labelany3d_amd.shard import gather_boxes

    B, steps, warmup = args.batch, args.steps, args.warmup
    # Input batches resident in HBM before the clock starts.  Batch 0 is the batch of every earlier round (seed 1234 + rank); with
    # --rotate R (default 3; 1 = the old protocol) R - 1 more batches of the same distribution follow and timed step k reads batch
    # k % R: no step finds the bytes of the step before it anywhere on the chip (1.9 GB per batch against 256 MB of Infinity Cache).
    R = 1 if args.config3 else max(1, args.rotate)
    free_b = torch.cuda.mem_get_info(device)[0]
    if R > 1 and R * B * H * W * 5 * 1.15 + steps * B * 400 > 0.8 * free_b:
        R = 1
    inputs = [build_inputs(args, B, device, 1234 + rank + 7919 * r, rank + 31 * r) for r in range(R)]
    x0 = inputs[0]
    depth, masks, image_index = x0["depth"], x0["all_masks"], x0.get("image_index")
    B = masks.shape[0]
    n_masked = sum(x["n_masked"] for x in inputs) / R
    fitter = InstanceFitter(B, H, W, device, slots=(1 if args.config3 else max(steps, 1)), ws_slots=max(args.streams, 2))
    stream = torch.cuda.current_stream()
    streams = [stream] + [torch.cuda.Stream(device=device) for _ in range(max(args.streams, 1) - 1)]
    run = StepRunner(fitter, inputs, one_slot=bool(args.config3))

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # measured read-only stream ceiling of THIS run (also what brings the chip to its working clocks before the warm-up steps:
    # a 20-step timed region entered from an idle chip reads ~5 % slower, profiles/r03/exp_step_ramp.py)
    stream_GBps, _, ceiling_launches = measured_stream_ceiling([x["all_masks"] for x in inputs])
    run.prepare([dict(slot=k, stream=streams[k % len(streams)], ws_slot=(k % len(streams)) if len(streams) > 1 else 0) for k in range(steps)])
    for w_ in range(warmup):
        run(slot=0, stream=stream, batch=w_ % R)
    if dist is not None:  # warm the communicator outside the timed region
        gather_boxes(fitter.boxes[:1].reshape(-1, 39), fitter.status[:1].reshape(-1), dst=0, counts=[B] * world)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record(stream)
    if len(streams) == 1:
        for k in range(steps):
            run(slot=k, stream=stream)
    else:
        for st in streams[1:]:
            st.wait_stream(stream)
        for k in range(steps):
            run(slot=k, stream=streams[k % len(streams)], ws_slot=k % len(streams))
        for st in streams[1:]:
            stream.wait_stream(st)
    ev1.record(stream)
    # The job's one collective - every rank's (steps x B, 39) records + status to rank 0 (north_star: "RCCL ... only for a final
    # result gather") - is INSIDE the timed region again (rounds 1-3 protocol; round 4 had moved it out): `value` is the whole
    # job between the two barriers, `value_fit_only` / `gather_ms` split it.  One rank: no collective, nothing to time.
    gathered, gather_s = None, 0.0
    if dist is not None:
        torch.cuda.synchronize()
        tg = time.perf_counter()
        # (every rank fits steps x B instances: the counts are known, so there is exactly ONE collective)
        gathered = gather_boxes(fitter.boxes.reshape(-1, 39), fitter.status.reshape(-1), dst=0,
                                counts=[fitter.boxes.shape[0] * B] * world)
        torch.cuda.synchronize()
        gather_s = time.perf_counter() - tg
    barrier()
    t1 = time.perf_counter()

    # secondary figure, AFTER the timed region (round 5; round 4 ran it before the warm-up, which conditioned the chip's clocks for
    # the timed steps - 2-4 us per step of the round-4 headline): the SAME serial step in back-to-back loops of 100 (>= 40 ms of
    # GPU time, HIP events on the launch stream) - what a caller that keeps fitting batches sees.
    # secondary, right behind the timed region: the same K serial steps all reading batch 0 (the protocol of rounds 1-5), HIP events
    same_batch_ms = None
    if R > 1 and len(streams) == 1 and not args.no_same_batch:
        sb0, sb1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        sb0.record(stream)
        for k in range(steps):
            run(slot=k, stream=stream, batch=0)
        sb1.record(stream)
        torch.cuda.synchronize()
        same_batch_ms = sb0.elapsed_time(sb1) / steps
    steady = None
    if not args.config3 and len(streams) == 1 and not args.no_steady:
        se0, se1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n_ss, tot_ms, best_ms = 0, 0.0, None
        for _ in range(4):
            torch.cuda.synchronize()
            se0.record(stream)
            for _k in range(100):
                run(slot=0, stream=stream, batch=_k % R)
            se1.record(stream)
            torch.cuda.synchronize()
            ms = se0.elapsed_time(se1) / 100
            n_ss += 100; tot_ms += ms * 100
            best_ms = ms if best_ms is None else min(best_ms, ms)
            if tot_ms >= 40.0:
                break
        steady = (best_ms, n_ss)

    # secondary figure, same run: the same K steps PIPELINED - issued round-robin on two HIP streams (batch k+1 is enqueued
    # while batch k runs), launch order off FOR THESE CALLS (it assumes an idle chip; la3d_fit_args::opt_launch_order).  What a
    # caller streaming many batches gets; the headline above stays the strictly serial form.
    pipelined = None
    if len(streams) == 1 and not args.no_pipelined and not args.config3 and not args.subsample:   # (config 3 is one multi-round call already)
        s2 = [stream, torch.cuda.Stream(device=device)]
        run.prepare([dict(slot=k, stream=s2[k % 2], ws_slot=k % 2, launch_order=False) for k in range(steps)])
        for k in range(max(4, warmup // 4)):
            run(slot=0, stream=s2[k % 2], ws_slot=k % 2, launch_order=False, batch=k % R)
        barrier()
        p0 = time.perf_counter()
        s2[1].wait_stream(stream)
        for k in range(steps):
            run(slot=k, stream=s2[k % 2], ws_slot=k % 2, launch_order=False)
        stream.wait_stream(s2[1])
        barrier()
        pel = torch.tensor([time.perf_counter() - p0], dtype=torch.float64, device=red_dev)
        if dist is not None:
            dist.all_reduce(pel, op=dist.ReduceOp.MAX)
        pipelined = float(pel)

    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=red_dev)
    kern_ms = torch.tensor([ev0.elapsed_time(ev1) / steps], dtype=torch.float64, device=red_dev)
    gather_t = torch.tensor([gather_s], dtype=torch.float64, device=red_dev)
    if dist is not None:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        dist.all_reduce(kern_ms, op=dist.ReduceOp.MAX)
        dist.all_reduce(gather_t, op=dist.ReduceOp.MAX)
    elapsed, kern_ms, gather_s = float(elapsed), float(kern_ms), float(gather_t)

    ok = int((fitter.status == 0).sum())
    if rank == 0:
        assert ok == fitter.status.numel(), f"{fitter.status.numel() - ok} boxes failed"
        if gathered is not None:
            assert gathered[0].shape[1] == 39, gathered[0].shape
        value = world * steps * B / elapsed
        alg_bytes = B * ALG_BYTES_PER_BOX
        if args.config3:   # shared-depth layout: depth plane once per image, mask + record per instance
            alg_bytes = args.config3 * H * W * 4 + B * (H * W + 39 * 8)
        # bytes this input cannot be fitted without: every mask plane once + the 128-B depth lines of the 32x8 tiles that hold
        # a mask pixel once + the records (computed from the masks; with run-length input the mask term is the run lengths)
        per_batch = [batch_required_bytes(args, x) for x in inputs]
        # (the timed steps read batch k % R: the per-launch figure is their average)
        req_bytes = sum(per_batch[k % R][0] for k in range(steps)) / steps
        active_tiles = sum(per_batch[k % R][1] for k in range(steps)) / steps
        step_s = kern_ms * 1e-3
        achieved = req_bytes / step_s / 1e9
        traffic, traffic_src, traffic_stale, traffic_kernel_ns, traffic_kernel_ns_long = None, None, None, None, None
        tp = os.path.join(ROOT, "profiles", "traffic_per_launch.json")
        mode = traffic_mode_key(args, B)
        if os.path.exists(tp):
            tj = json.load(open(tp))
            ent = tj.get("modes", {}).get(mode)
            if ent:
                traffic, traffic_src = ent.get("hbm_bytes_per_step"), ent.get("source")
                traffic_kernel_ns = ent.get("dominant_kernel_avg_ns")
                traffic_kernel_ns_long = ent.get("dominant_kernel_avg_ns_default_length")
                traffic_stale = tj.get("kernel_source_sha256") != kernel_source_sha256()
        # shader-side counters of the same profile set (profiles/make_valu_json.py): how busy the fp64 VALU - the path's second roof - is
        valu = None
        try:
            vj = json.load(open(os.path.join(ROOT, "profiles", "valu_per_launch.json")))
            ve = vj["modes"].get(mode)
            if ve is not None:
                valu = {"busy_us_per_simd_per_step": ve["valu_busy_us_per_simd"], "frac_of_step": ve["valu_busy_us_per_simd"] / (kern_ms * 1e3),
                        "wave_instructions_per_step": ve["wave_instructions_valu"], "utilisation_in_the_profiled_kernel": ve["valu_utilisation"],
                        "stale": vj.get("kernel_source_sha256") != kernel_source_sha256(),
                        "source": f"rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU of the same bench command ({ve['summary']}): "
                                  "SQ_ACTIVE_INST_VALU x 4 cycles / 1024 SIMDs / 2.4 GHz"}
        except Exception:   # the file is optional
            valu = None
        if args.config3:
            workload = (f"BASELINE config-3 stand-in: {args.config3} images with a SHARED 480x640 depth plane each, {B} instances "
                        "(~Poisson(7) per image, elliptical u8 masks, log-uniform area 400..100k px) in one call per step")
        elif args.config5:
            workload = (f"BASELINE config 5 workload: {B} instances per GPU per step, private 480x640 f32 depth, u8 masks with "
                        "log-uniform area 8..100k px (rectangles below 400 px, ellipses above), one call per step")
        else:
            workload = ("BASELINE config 2: 1024 instances per GPU per step, private 480x640 f32 depth ~U(0.5,10) "
                        "+ u8 rectangular mask per instance, K=[[500,0,320],[0,500,240],[0,0,1]], ground=None, "
                        "full-mask mode; inputs resident in HBM")
        if args.area_hint:
            workload += "; mask areas handed to the fit (area_hint): no estimate pass"
        if args.ground:
            workload = workload.replace("ground=None", "ground = one plane per instance (normal within a few degrees of -y) - not the config-2 call")
            if "ground = one" not in workload:
                workload += "; one ground plane per instance"
        if args.subsample:
            workload = workload.replace("full-mask mode", "reference-subsample mode (500 drawn points per mask above 500 px)") \
                if "full-mask mode" in workload else workload + "; reference-subsample mode (500 drawn points per mask above 500 px)"
        out = {
            "metric": "fitted 3D boxes/sec @640x480",
            "value": value,
            "unit": "boxes/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "ranks": ranks,
            "gather_ms": (gather_s * 1e3) if dist is not None else None,
            "value_fit_only": (world * steps * B / (kern_ms * 1e-3 * steps)) if dist is not None else None,
            "untimed_launches_before_timed_region": {"fit_steps": warmup, "other_kernels": ceiling_launches,
                                                     "other_kernels_what": "la3d_mask_counts (the pure-reader stream ceiling, a different kernel)"},
            "methodology": "r06: as r05, and timed step k reads resident input batch k % R (--rotate, default 3; `rotation` holds the same-batch figure). r05: stream-ceiling measurement, W warm-up steps, then the K timed steps between two barriers (N > 1: the one "
                           "final gather inside the region, as in rounds 1-3); steady-state and pipelined loops run AFTER the timed region "
                           "(round 4 ran 100-400 steady-state steps before the warm-up)",
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "instances_per_gpu": B,
                "frame": [H, W],
                "mean_mask_occupancy": n_masked / (B * H * W),
                "active_tiles_per_instance": active_tiles / B,
                "sharding": ("single GPU" if dist is None else "instances sharded per rank, one final RCCL gather of box records"
                             if dist.get_backend() == "nccl" else
                             f"DRY RUN: {world} ranks sharing one GPU through {dist.get_backend()} (functional check of the multi-rank path, not a measurement)"),
                "mask_input": ("COCO run lengths (la3d_fit_instances_rle) — not the config-2 format" if args.rle else
                               "polygon parts (la3d_fit_instances_poly), 4-vertex rings — not the config-2 format" if args.poly else "u8 planes"),
                "streams": len(streams),
                "input_batches": R,
                "default_steps": "1000 timed steps / 50 warm-up (0.11 s timed region); any --steps works",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": ("fit_instances_kernel<VEC,LDSMASK,SAMPLE=1> (instance engine, reference-subsample mode)" if args.subsample else
                           "fit_instances_kernel<VEC,LDSMASK,SAMPLE=0,TILED,SRC,RET=0> (instance engine: 64 VGPRs, four workgroups per CU; round 5: "
                           "un-grounded, skew-free cameras - every BASELINE config - take the SEPARABLE SINGLE PASS: one walk over the depth, moments "
                           "factorised per column, x / z extents from per-column depth ranges in LDS, no pass B; grounded calls keep the two-pass "
                           "form with pass-B tile culling)"),
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "frac_required": achieved / HBM_PEAK_GBPS,
                "frac_traffic": (traffic / step_s / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                "frac_algorithmic_model": alg_bytes / step_s / 1e9 / HBM_PEAK_GBPS,
                "required_bytes_per_launch": req_bytes,
                "byte_model": "B*H*W mask bytes once + 1024 B x (32 px x 8 row tiles holding a mask pixel; shared depth planes: "
                              "union per image) depth once + 312 B x B records; computed from the generated masks in this run",
                "measured_stream_GBps": stream_GBps,
                "frac_of_measured_stream": achieved / stream_GBps if stream_GBps else None,
                "avg_launch_ms": kern_ms,
                "algorithmic_bytes_per_launch": alg_bytes,
                "algorithmic_GBps": alg_bytes / step_s / 1e9,
                "traffic": traffic,
                "traffic_GBps": (traffic / step_s / 1e9) if traffic else None,
                "traffic_frac_of_peak": (traffic / step_s / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                "traffic_source": traffic_src,
                "traffic_stale": traffic_stale,
                "valu": valu,
                "traffic_mode": mode,
                "traffic_kernel_avg_ns_under_rocprof": traffic_kernel_ns,
                # (the counter passes are 55-launch runs from an idle chip with the counter collection on: 2-3 % above a steady step;
                # the same kernel in the profile of the default-length command - 1000 timed steps + the secondary loops:)
                "kernel_avg_ns_under_rocprof_default_length": traffic_kernel_ns_long,
                "note": "frac = frac_required = required bytes (byte_model) / avg_launch_ms / 8 TB/s; frac_traffic = PMC-measured bytes / avg_launch_ms / 8 TB/s; frac_algorithmic_model = SURVEY 8d's H*W*5+312 B/box / avg_launch_ms / 8 TB/s (above 1: not a fraction of anything physical); avg_launch_ms is avg_launch_ms is the HIP-EVENT time per step on "
                        "the launch stream (max over ranks); `value` and ms_per_step are on the WALL clock between the two barriers (a few "
                        "us per step more at K = 20). measured_stream_GBps = la3d_mask_counts (a pure 16-byte-load reader) over the same "
                        "mask planes, HIP events, same run, before the warm-up. "
                        "A step = ONE kernel: ordered launches (256 < B <= 3072) estimate their sort keys in the fit kernel's prologue since the end of "
                        "round 4 (no helper launch; the retaining build, the band engine for B > 256 and graph-captured calls keep the ~5 us estimate kernel). "
                        "algorithmic_GBps is SURVEY 8d's H*W*5+312 B/box figure (the kernel never loads depth lines without a mask "
                        "pixel, so that figure exceeds the physical peak and is NOT a roofline fraction). traffic = PMC-measured "
                        "HBM bytes per launch of the profiled build (profiles/, TCC_EA0_RDREQ x 128 B + WRITE_SIZE); "
                        "traffic_stale = the kernel sources changed since that profile. valu = the second roof: the time one SIMD spends issuing this "
                        "step's VALU instructions (fp64 ~5.2 cycles per wave-instruction), from the shader-side counters of the same profile set.",
            },
        }
        out["rotation"] = {
            "batches": R, "resident_input_bytes": R * int(depth.numel() * 4 + masks.numel()),
            "required_bytes_per_batch": [pb[0] for pb in per_batch],
            "same_batch_ms_per_step": same_batch_ms,
            "same_batch_value": (world * B / (same_batch_ms * 1e-3)) if same_batch_ms else None,
            "note": "timed step k reads resident batch k % batches (same distribution, different seeds: batch 0 is the batch of rounds 1-5), so "
                    "no step re-reads the bytes of its predecessor; same_batch_* = the same K serial steps all on batch 0, HIP events, run "
                    "right after the timed region (rank 0's figure x ranks) - the two agree when no cache level helps the repeated read",
        }
        out["build"] = build_identity()
        if steady is not None:
            out["steady_state"] = {
                "value": world * B / (steady[0] * 1e-3), "unit": "boxes/s", "ms_per_step": steady[0], "steps_run": steady[1],
                "note": "secondary: the same serial step (one batch after the other on one stream) in back-to-back loops of 100, HIP events, "
                        "best loop, run AFTER the timed region (rank 0's figure x ranks); the headline `value` is the K steps the driver asked for",
            }
        if pipelined is not None:
            out["pipelined"] = {
                "value": world * steps * B / pipelined, "unit": "boxes/s", "ms_per_step": pipelined / steps * 1e3, "streams": 2,
                "required_GBps": req_bytes / (pipelined / steps) / 1e9, "frac": req_bytes / (pipelined / steps) / 1e9 / HBM_PEAK_GBPS,
                "note": "secondary: the same K steps issued round-robin on two HIP streams (independent batches overlap their "
                        "memory-bound mask stream and their passes), size-balanced launch order off for these calls (la3d_fit_args::opt_launch_order); "
                        "wall clock over K steps between the same barriers; not the headline",
            }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(depth, masks)
            # the figure not to confuse it with: the REFERENCE's own loop amortises depth_to_points per image (one call per frame,
            # src/batch_scripts/depth.py:154) - the port above unprojects a private plane per instance (34 ms each) as config 2 defines it
            out["cpu_baseline"]["reference_probe"] = {
                "value": 573.0, "unit": "boxes/s/core", "ms_per_instance": 1.75,
                "what": "the reference's own functions (depth_to_points amortised per image + pts[mask] + estimate_bbox) on 128 config-2 "
                        "masks, single thread, measured in the survey container (8 vCPU Xeon 2.10 GHz) - BASELINE.md section 2; "
                        "not measured on this box: /root/reference does not travel"}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
