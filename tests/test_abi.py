"""CPU-only: the C-ABI library loads, exports every symbol include/la3d.h declares, and the host-side
pieces (fp16 rounding routine shared with the kernels, NumPy helper functions, cam_utils) match the
reference fixtures.  No device compute here."""
import os
import re

import numpy as np
import pytest

from tests.conftest import ROOT


def test_library_exports_every_declared_symbol():
    from labelany3d_amd import _lib

    hdr = open(os.path.join(ROOT, "include", "la3d.h")).read()
    declared = sorted(set(re.findall(r"\b(la3d_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(_lib.lib, name), f"{name} declared in la3d.h but not exported by libla3d.so"
    assert sorted(_lib.EXPORTS) == declared
    assert _lib.lib.la3d_version() == 2   # ABI 2 (round 4): per-call opt_* fields, no process-wide launch-order switch
    assert _lib.lib.la3d_workspace_bytes(1024, 480, 640) >= 1024 * 160  # >= per-instance geometry; + split-engine buffers
    assert _lib.lib.la3d_workspace_bytes(1024, 37, 53) == 1024 * 160  # frames the split engine does not take
    assert _lib.lib.la3d_workspace_bytes(0, 480, 640) == 0
    # the row engine (round 5, at most 512 instances): sixteen bands of one instance leave 20 doubles and 2 W column words each
    assert _lib.lib.la3d_workspace_bytes(1, 480, 640) >= 16 * (20 * 8 + 2 * 640 * 4)
    assert _lib.lib.la3d_workspace_bytes(64, 480, 640) >= 64 * 8 * (20 * 8 + 2 * 640 * 4)


def test_so_is_in_tree_and_gfx950():
    from labelany3d_amd import _lib

    assert _lib.LIB.startswith(ROOT)
    blob = open(_lib.LIB, "rb").read()
    assert b"gfx950" in blob
    for kern in (b"fit_instances_kernel", b"fit_points_kernel", b"unproject_kernel", b"mask_counts_kernel"):
        assert kern in blob


def test_f16_round_matches_numpy_astype():
    from labelany3d_amd import _lib

    rs = np.random.RandomState(0)
    vals = np.concatenate([
        rs.randn(2000) * 10.0 ** rs.uniform(-9, 5, 2000),
        np.float16(rs.randn(500)).astype(np.float64) * (1 + 2.0 ** -11),   # exact ties -> even
        np.float16(rs.randn(500)).astype(np.float64) * (1 - 2.0 ** -12),
        [0.0, -0.0, 65504.0, 65519.999, 65520.0, -65520.0, 1e6, -1e6, 2.0 ** -14, 2.0 ** -24, 2.0 ** -25,
         1.5 * 2.0 ** -25, 2.0 ** -26, 6.1e-5, 5.96e-8, np.inf, -np.inf],
    ])
    with np.errstate(over="ignore"):
        want = vals.astype(np.float16).astype(np.float64)
    got = np.array([_lib.lib.la3d_f16_round_host(float(v)) for v in vals])
    np.testing.assert_array_equal(got, want)
    assert np.isnan(_lib.lib.la3d_f16_round_host(float("nan")))


def test_fails_loudly_without_extension(tmp_path, monkeypatch):
    """The product path has no CPU fallback: a missing .so is an ImportError, not a slow path."""
    import importlib.util

    from labelany3d_amd import _build

    monkeypatch.setattr(_build, "LIB", str(tmp_path / "nope.so"))
    spec = importlib.util.spec_from_file_location("labelany3d_amd._lib_probe", os.path.join(ROOT, "labelany3d_amd", "_lib.py"),
                                                  submodule_search_locations=None)
    mod = importlib.util.module_from_spec(spec)
    mod.__package__ = "labelany3d_amd"
    with pytest.raises(ImportError, match="no CPU fallback"):
        spec.loader.exec_module(mod)


def test_helper_functions_vs_reference(golden):
    from labelany3d_amd import util_3dbox as U

    g = golden("g6_helpers.npz")
    tol = dict(rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(np.array([U.rotate_y(y) for y in g["rot_yaws"]]), g["rot_out"], **tol)
    got = np.array([U.rotation_matrix_from_vectors(a, b) for a, b in zip(g["rm_v1"], g["rm_v2"])])
    np.testing.assert_allclose(got, g["rm_out"], **tol)
    assert np.isnan(U.rotation_matrix_from_vectors([0, -1, 0], [0, -2.0, 0])).all()
    assert np.isnan(U.rotation_matrix_from_vectors([0, -1, 0], [0, 3.0, 0])).all()
    z = np.zeros(3)
    assert U.normalize(z) is z
    np.testing.assert_allclose(U.normalize(g["norm_in"]), g["norm_out"], **tol)
    np.testing.assert_allclose(np.array([U.convert_box_vertices(*r) for r in g["cbv_in"]]), g["cbv_out"], **tol)
    p = g["p2p_in"]
    assert U.point_to_plane_distance(p[:4], *p[4:]) == pytest.approx(float(g["p2p_out"]), rel=1e-14)


def test_cam_utils_vs_reference(golden):
    from labelany3d_amd import cam_utils as CU

    g = golden("g6_helpers.npz")
    for ogl in (True, False):
        got = np.array([[CU.orbit_camera(e, a, radius=2.5, opengl=ogl) for a in g["orbit_azim"]] for e in g["orbit_elev"]])
        assert got.dtype == np.float32 and got.shape[-2:] == (4, 4)
        np.testing.assert_array_equal(got, g[f"orbit_opengl{int(ogl)}"])
    got = CU.orbit_camera(0.3, -1.1, radius=1.7, is_degree=False, target=np.array([0.5, 0.1, -0.2], dtype=np.float32))
    np.testing.assert_array_equal(got, g["orbit_rad_target"])
    np.testing.assert_array_equal(CU.look_at(g["look_campos"], g["look_target"], True), g["look_opengl1"])
    np.testing.assert_array_equal(CU.look_at(g["look_campos"], g["look_target"], False), g["look_opengl0"])
    np.testing.assert_array_equal(CU.length(g["length_in"]), g["length_out"])
    np.testing.assert_array_equal(CU.safe_normalize(g["length_in"]), g["safe_norm_out"])
    import torch

    x = torch.tensor(g["length_in"])
    np.testing.assert_allclose(CU.length(x).numpy(), g["length_out"])  # works here; NameError in the reference


def test_compat_modules_resolve_by_bare_name(monkeypatch):
    import importlib
    import sys

    monkeypatch.syspath_prepend(os.path.join(ROOT, "labelany3d_amd", "compat"))
    for name in ("util", "util_3dbox", "cam_utils"):
        sys.modules.pop(name, None)
    u3 = importlib.import_module("util_3dbox")
    u = importlib.import_module("util")
    cu = importlib.import_module("cam_utils")
    for fn in ("normalize", "rotate_y", "rotation_matrix_from_vectors", "point_to_plane_distance", "convert_box_vertices",
               "estimate_bbox", "_estimate_yaw_pca", "_estimate_yaw_convex_hull", "save_3d_with_ground_alignment_bbox"):
        assert callable(getattr(u3, fn)), fn
    assert callable(u.depth_to_points)
    for fn in ("length", "safe_normalize", "look_at", "orbit_camera"):
        assert callable(getattr(cu, fn))
    for name in ("util", "util_3dbox", "cam_utils"):
        sys.modules.pop(name, None)


def test_compat_shim_falls_back_to_the_reference_module(tmp_path, monkeypatch):
    """The import lines of the reference's stages through the shim: `from util import restore_mask_from_crop,
    align_to_depth_match, draw_cube` (whole.py:15) must still resolve — names the shim does not define come from the
    reference's own util.py further down sys.path — while depth_to_points is the MI355X one.  And install() patches the
    reference's modules in place when './' is first on sys.path (whole.py:10)."""
    import importlib
    import sys

    ref = tmp_path / "src"
    ref.mkdir()
    (ref / "util.py").write_text("def depth_to_points(*a, **k):\n    return 'reference'\n\ndef restore_mask_from_crop():\n"
                                 "    return 'ref-helper'\n\ndef draw_cube():\n    return 'ref-draw'\n")
    (ref / "util_3dbox.py").write_text("def estimate_bbox(*a, **k):\n    return 'reference'\n\ndef normalize(v):\n    return 'ref-normalize'\n"
                                       "def save_3d_with_ground_alignment_bbox(*a, **k):\n    return 'reference'\n"
                                       "def _estimate_yaw_pca(*a):\n    return 0\n\ndef _estimate_yaw_convex_hull(*a):\n    return 0\n")
    compat = os.path.join(ROOT, "labelany3d_amd", "compat")
    import labelany3d_amd.compat as C

    def fresh():
        for name in ("util", "util_3dbox", "cam_utils"):
            sys.modules.pop(name, None)
        C._loaded.clear()

    # (a) shim directory first on the path
    fresh()
    monkeypatch.setattr(sys, "path", [compat, str(ref)] + [p for p in sys.path if p not in (compat, str(ref))])
    ns = {}
    exec("from util import restore_mask_from_crop, draw_cube, depth_to_points", ns)
    assert ns["restore_mask_from_crop"]() == "ref-helper" and ns["draw_cube"]() == "ref-draw"
    import labelany3d_amd.util as impl_util
    assert ns["depth_to_points"] is impl_util.depth_to_points
    with pytest.raises(ImportError):
        exec("from util import no_such_function", {})
    # (b) the reference's own directory first ('./' in the stages): install() patches in place
    fresh()
    monkeypatch.setattr(sys, "path", [str(ref)] + [p for p in sys.path if p not in (compat, str(ref))])
    patched = C.install()
    assert "util.depth_to_points" in patched and "util_3dbox.estimate_bbox" in patched
    ns = {}
    exec("from util import depth_to_points, restore_mask_from_crop\nfrom util_3dbox import save_3d_with_ground_alignment_bbox, normalize", ns)
    import labelany3d_amd.util_3dbox as impl_3d
    assert ns["depth_to_points"] is impl_util.depth_to_points and ns["restore_mask_from_crop"]() == "ref-helper"
    assert ns["save_3d_with_ground_alignment_bbox"] is impl_3d.save_3d_with_ground_alignment_bbox
    assert ns["normalize"](1) == "ref-normalize"           # everything else stays the reference's
    fresh()


def test_draw_sample_idx_consumes_global_stream_like_reference():
    from labelany3d_amd import draw_sample_idx

    counts = [800, 300, 1200, 500, 501]
    np.random.seed(77)
    idx = draw_sample_idx(counts)
    np.random.seed(77)
    a = np.random.randint(0, 800, 500)
    b = np.random.randint(0, 1200, 500)
    c = np.random.randint(0, 501, 500)
    np.testing.assert_array_equal(idx[0], a)
    np.testing.assert_array_equal(idx[2], b)
    np.testing.assert_array_equal(idx[4], c)
    assert (idx[1] == 0).all() and (idx[3] == 0).all()


def test_rle_string_host_helper_matches_oracle(golden):
    """la3d_rle_from_string_host (pycocotools rleFrString restated in C) against the oracle's codec on the
    reference encoder's outputs."""
    from labelany3d_amd import pack_rle, rle_from_string
    from oracle import la3d_oracle as O

    g = golden("g8_masks.npz")
    offs = np.concatenate([[0], np.cumsum(g["lens"])])
    rles = []
    for i in range(len(g["lens"])):
        counts = g["counts"][offs[i]:offs[i + 1]].tolist()
        s = O.rle_to_string(counts)
        np.testing.assert_array_equal(rle_from_string(s), counts)
        rles.append({"size": [48, 64], "counts": s if i % 2 else counts})   # mixed list / string forms
    c, o, H, W = pack_rle(rles)
    assert (H, W) == (48, 64)
    np.testing.assert_array_equal(o, offs)
    np.testing.assert_array_equal(c, g["counts"])
    with pytest.raises(ValueError):
        rle_from_string(b"P")              # 0x50 - 48 has the continuation bit set, then the string ends


def test_omni3d_category_table_is_data():
    """labelany3d_amd/data/omni3d_coco_categories.json: the 80 COCO names with the ids the reference's writer uses
    (src/tools/combine_results.py:17-101); pure data, read without the GPU library."""
    import json
    import os

    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "labelany3d_amd", "data", "omni3d_coco_categories.json")
    cats = json.load(open(p))
    assert len(cats) == 80 and len({c["id"] for c in cats}) == 80 and len({c["name"] for c in cats}) == 80
    by = {c["name"]: c["id"] for c in cats}
    assert by["person"] == 7 and by["car"] == 1 and by["teddy bear"] == 151 and all(set(c) == {"supercategory", "id", "name"} for c in cats)


def test_upload_many_packs_host_arrays_into_one_buffer():
    """batched._upload_many / _bulk (host logic of the per-image wrappers): several small host arrays -> views of ONE buffer (one
    host-to-device copy), dtype conversion, 16-byte aligned pieces, None and empty arrays, tensors passed through untouched."""
    import torch

    from labelany3d_amd.batched import _bulk, _upload_many

    cpu = torch.device("cpu")
    xy = np.arange(14, dtype=np.int64).reshape(7, 2)            # converted to int32
    ro = np.array([0, 3, 7], np.int64)
    g = np.array([[0.1, -0.9, 0.2, 1.0]], np.float32)           # converted to float64
    empty = np.zeros((0, 4))
    out = _upload_many([(xy, torch.int32), (ro, torch.int64), (None, torch.int32), (g, torch.float64), (empty, torch.float64)], cpu)
    assert out[2] is None
    assert out[0].dtype == torch.int32 and out[0].shape == (7, 2) and np.array_equal(out[0].numpy(), xy)
    assert out[1].dtype == torch.int64 and np.array_equal(out[1].numpy(), ro)
    assert out[3].dtype == torch.float64 and np.allclose(out[3].numpy(), g) and out[4].shape == (0, 4)
    base = out[0].untyped_storage().data_ptr()
    for t in (out[0], out[1], out[3]):
        assert t.untyped_storage().data_ptr() == base and (t.data_ptr() - base) % 16 == 0        # one buffer, aligned pieces
    keep = torch.arange(3)
    a, b, c, d = _bulk(cpu, (xy, torch.int32), (keep, torch.int64), (None, torch.float64), (ro, torch.int64))
    assert b is keep and c is None and isinstance(a, torch.Tensor) and isinstance(d, torch.Tensor)
    assert a.untyped_storage().data_ptr() == d.untyped_storage().data_ptr()
    one = _bulk(cpu, (xy, torch.int32), (None, torch.int64))
    assert one[0] is xy and one[1] is None                                                      # a single host array is left to _as_dev


@pytest.mark.gpu
def test_scheduling_is_per_call_not_process_state():
    """SURVEY 8b 're-entrant ... no global state': two threads issue fits with DIFFERENT per-call scheduling (launch order on / off,
    plain / no-cull build of the instance kernel) at the same time, each on its own stream and workspace; both get the records
    a lone two-pass call gets, bit for bit.  (ABI 1 had a process-wide la3d_set_launch_order; ABI 2 carries the choice in
    la3d_fit_args::opt_*.)"""
    import threading

    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import bench
    from labelany3d_amd import InstanceFitter, scheduling

    dev = torch.device("cuda", 0)
    B = 600
    depth, masks, K, _, _ = bench.make_inputs(B, dev, 77)
    ref_f = InstanceFitter(B, bench.H, bench.W, dev)
    with scheduling(engine="instance", build="plain"):   # (the default build takes the separable single pass here: equal to rounding only)
        want = tuple(t.clone() for t in ref_f.run(depth, masks, K))
    torch.cuda.synchronize()
    results, errors = {}, []

    def worker(name, order, build):
        try:
            st = torch.cuda.Stream(device=dev)
            f = InstanceFitter(B, bench.H, bench.W, dev)
            st.wait_stream(torch.cuda.current_stream(dev))
            with scheduling(engine="instance", launch_order=order, build=build):   # thread-local
                for _ in range(20):
                    out = f.run(depth, masks, K, stream=st)
            st.synchronize()
            results[name] = tuple(t.clone() for t in out)
        except Exception as e:  # noqa: BLE001
            errors.append((name, e))

    ths = [threading.Thread(target=worker, args=("a", True, "nocull")), threading.Thread(target=worker, args=("b", False, "plain"))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors, errors
    for name in ("a", "b"):
        for g, w in zip(results[name], want):
            assert torch.equal(torch.nan_to_num(g.double(), nan=-7.0), torch.nan_to_num(w.double(), nan=-7.0)), name


@pytest.mark.gpu
def test_bad_scheduling_options_are_rejected():
    import ctypes as C

    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from labelany3d_amd import _lib, options

    a = _lib.FitArgs()
    a.struct_size = C.sizeof(_lib.FitArgs)
    a.B, a.H, a.W = 1, 8, 32
    a.opt_engine = 7
    assert _lib.lib.la3d_fit_instances_ex(C.byref(a)) != 0
    with pytest.raises(ValueError):
        options.codes(engine="warp")
    with pytest.raises(ValueError):
        with options.scheduling(build="huge"):
            pass


def test_scheduling_options_are_thread_local_and_validated():
    """labelany3d_amd.options: the per-call codes of la3d_fit_args::opt_*; a `scheduling` block pins them for ITS thread only and
    restores the previous state on exit (no GPU needed)."""
    import threading

    from labelany3d_amd import options

    assert options.codes() == (0, 0, 0)
    assert options.codes(engine="band", launch_order=False, build="nocull") == (3, 1, 2)
    assert options.codes(engine="rows") == (4, 0, 0)
    seen = {}

    def other():
        seen["other"] = options.codes()

    with options.scheduling(engine="split", launch_order=True):
        assert options.codes() == (2, 2, 0)
        assert options.codes(engine="instance") == (1, 2, 0)          # an explicit argument wins over the block
        t = threading.Thread(target=other)
        t.start(); t.join()
        with options.scheduling(build="plain", launch_order=None):
            assert options.codes() == (2, 0, 1)
        assert options.codes() == (2, 2, 0)
    assert options.codes() == (0, 0, 0) and seen["other"] == (0, 0, 0)
    with pytest.raises(ValueError):
        options.codes(build="big")


def test_pack_polygons_accepts_every_form_the_reference_accepts():
    """np.array(polygon).reshape(-1, 2).astype(np.int32) (reference src/util.py:398) takes flat lists, nested pairs and arrays;
    the one-loop fast path of pack_polygons is for flat lists of scalars only and must not swallow the other forms."""
    import numpy as np

    from labelany3d_amd.masks import pack_polygons

    flat = [[[1.5, 2.5, 10.2, 2.0, 9.9, 8.1]], [[0, 0, 4, 0, 4, 4, 0, 4], [2, 2, 3, 2, 3, 3]]]
    nested = [[[[1.5, 2.5], [10.2, 2.0], [9.9, 8.1]]], [[[0, 0], [4, 0], [4, 4], [0, 4]], [[2, 2], [3, 2], [3, 3]]]]
    arrays = [[np.array(p) for p in seg] for seg in flat]
    mixed = [flat[0], nested[1]]
    want = pack_polygons(flat, 16, 16)
    for other in (nested, arrays, mixed):
        got = pack_polygons(other, 16, 16)
        for a, b in zip(want[:3], got[:3]):
            np.testing.assert_array_equal(a, b)
    assert want[0].dtype == np.int32 and want[0].tolist()[:3] == [[1, 2], [10, 2], [9, 8]]
    import pytest
    with pytest.raises(ValueError):
        pack_polygons([[[1, 2, 3]]], 16, 16)     # odd count: the reference's reshape error


def test_library_carries_the_identity_of_its_sources():
    """la3d_build_info(): the hash of the sources the loaded library was compiled from = the hash of the tree's sources (the
    build recipe reuses a library only then), readable from the file without loading it."""
    import importlib.util

    from labelany3d_amd import _lib

    spec = importlib.util.spec_from_file_location("_la3d_build_t", os.path.join(ROOT, "labelany3d_amd", "_build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    tag, src, cmd = _lib.lib.la3d_build_info().decode().split(":")
    assert tag == "LA3D_BUILD_INFO" and len(src) == 64 and len(cmd) == 64
    assert b.embedded_info(_lib.LIB) == (src, cmd)
    assert src == b.source_sha256(), "libla3d.so was not built from the sources in the tree: run __graft_entry__.build()"
