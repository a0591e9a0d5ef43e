"""ctypes binding of libla3d.so (C-ABI in include/la3d.h).  No CPU fallback: if the HIP library
is missing, importing this module raises — the product path is the GPU path."""
from __future__ import annotations

import ctypes as C
import os

from ._build import HERE, INCLUDE, LIB, ROOT, SOURCES, build  # noqa: F401

REC, AUX, NSAMPLE = 39, 4, 500
BOX_OK, BOX_EMPTY, BOX_BAD_GROUND, BOX_TOO_FEW, BOX_NONFINITE, BOX_UNSUPPORTED, BOX_FILTERED = 0, 1, 2, 3, 4, 5, 6
METHOD_PCA, METHOD_CONVEX_HULL = 0, 1
HINT_SMALL_CLOUDS = 0x100
HINT_HULL_512 = 0x200
ERR_UNSUPPORTED = -2

class FitArgs(C.Structure):
    """``la3d_fit_args`` of include/la3d.h (argument block of la3d_fit_instances_ex); field order is the header's."""
    _fields_ = [("struct_size", C.c_int32), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("depth", C.c_void_p), ("depth_plane_stride", C.c_int64), ("image_index", C.c_void_p),
                ("mask", C.c_void_p),
                ("rle_counts", C.c_void_p), ("rle_offsets", C.c_void_p),
                ("poly_xy", C.c_void_p), ("ring_offsets", C.c_void_p), ("inst_rings", C.c_void_p),
                ("K", C.c_void_p), ("k_stride", C.c_int32),
                ("filter_boundary", C.c_int32), ("filter_min_area", C.c_int32), ("filter_max_edge", C.c_int32),
                ("ground", C.c_void_p), ("sample_idx", C.c_void_p),
                ("stats", C.c_void_p),
                ("proj", C.c_void_p), ("image_width", C.c_double), ("image_height", C.c_double),
                ("out", C.c_void_p), ("status", C.c_void_p), ("aux", C.c_void_p),
                ("workspace", C.c_void_p), ("stream", C.c_void_p),
                ("area_hint", C.c_void_p),
                ("opt_engine", C.c_int32), ("opt_launch_order", C.c_int32), ("opt_build", C.c_int32), ("frame_width", C.c_int32)]


_SIGS = {
    "la3d_fit_instances_ex": (C.c_int, [C.POINTER(FitArgs)]),
    "la3d_version": (C.c_int, []),
    "la3d_last_error": (C.c_char_p, []),
    "la3d_unproject": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.c_int,
                                 C.c_void_p, C.c_int, C.c_void_p]),
    "la3d_unproject_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_int, C.c_void_p]),
    "la3d_mask_counts": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "la3d_pad_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "la3d_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "la3d_fit_instances": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                     C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    "la3d_fit_points": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "la3d_f16_round_host": (C.c_double, [C.c_double]),
    "la3d_fit_instances_rle": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                         C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p]),
    "la3d_fit_instances_poly": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                          C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "la3d_fit_instances_rle_filtered": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                                  C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "la3d_fit_instances_poly_filtered": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_int32, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                   C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "la3d_poly_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "la3d_mask_stats_poly": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                       C.c_void_p]),
    "la3d_rle_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "la3d_mask_stats": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "la3d_mask_stats_rle": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "la3d_masked_ratio_median": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                           C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "la3d_align_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "la3d_align_select": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    "la3d_align_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "la3d_align_select_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "la3d_unproject_matches": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double,
                                         C.c_double, C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                         C.c_void_p, C.c_void_p, C.c_void_p]),
    "la3d_project_boxes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int, C.c_double, C.c_double,
                                     C.c_void_p, C.c_void_p]),
    "la3d_iou_matrix": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "la3d_build_info": (C.c_char_p, []),
    "la3d_rle_from_string_host": (C.c_int, [C.c_char_p, C.c_int64, C.POINTER(C.c_int32), C.c_int]),
    "la3d_estimate_bbox_host": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    "la3d_unproject_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "la3d_host_release": (None, []),
    "la3d_fit_annotations_host": (C.c_int, [C.POINTER(FitArgs)]),
    "la3d_gather_planes_host": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int]),
    "la3d_3dbbox_json_bound": (C.c_int64, [C.c_int64, C.c_int64, C.c_int64]),
    "la3d_format_3dbbox_json": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                            C.c_int64, C.c_void_p]),
}

EXPORTS = tuple(_SIGS)


def _refresh_if_stale() -> None:
    """The library carries the hash of the sources it was compiled from (la3d_build_info).  A library that travelled with a
    snapshot but was built from OTHER sources than the tree's is rebuilt here, once, under a file lock (several test processes
    may import at the same time) - when hipcc is there; otherwise it is used as it is, loudly.  LA3D_NO_AUTOBUILD=1 switches
    the rebuild off (the library is then loaded as it is; bench.py's `build.lib_built_from_tree` tells)."""
    import sys

    from . import _build
    try:
        have = _build.embedded_info(LIB)
        if have is not None and have[0] == _build.source_sha256():
            return
        if os.environ.get("LA3D_NO_AUTOBUILD") == "1" or os.environ.get("LA3D_LIB"):
            return
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        if not os.path.exists(hipcc):
            print(f"labelany3d_amd: {LIB} was NOT built from the sources in this tree and there is no hipcc to rebuild it", file=sys.stderr)
            return
        import fcntl
        with open(LIB + ".lock", "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            have = _build.embedded_info(LIB)                    # (another process may have rebuilt it meanwhile)
            if have is None or have[0] != _build.source_sha256():
                print(f"labelany3d_amd: {os.path.basename(LIB)} is stale (built from other sources): rebuilding for gfx950 ...", file=sys.stderr)
                _build.build(force=True)
    except Exception as e:  # noqa: BLE001 - never in the way of loading what is there
        print(f"labelany3d_amd: could not refresh {LIB}: {e!r}", file=sys.stderr)


def load() -> C.CDLL:
    if not os.path.exists(LIB):
        raise ImportError(
            f"{LIB} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(needs hipcc; cross-compiles gfx950 without a GPU). There is no CPU fallback."
        )
    _refresh_if_stale()
    # PyTorch bundles its own libamdhip64; load it FIRST so that libla3d.so binds to the same HIP runtime instance as the
    # tensors it is handed (loaded the other way round, the process ends up with two runtimes and the library's stream /
    # event calls fail with "no ROCm-capable device is detected")
    import torch  # noqa: F401

    lib = C.CDLL(LIB)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    return lib


lib = load()


class La3dError(RuntimeError):
    pass


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise La3dError(f"{what} failed ({rc}): {lib.la3d_last_error().decode()}")
