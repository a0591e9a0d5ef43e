"""Import the reference's hot-path modules read-only from /root/reference (THIS container only).

Used only by tests/golden/make_golden.py to generate fixtures.  The reference's
``util.py`` / ``util_3dbox.py`` import cv2 / trimesh / rembg / pycocotools at module top
(reference src/util.py:1-10, src/util_3dbox.py:8-13); none of them is touched by the
hot-path functions, so empty stand-in modules are registered in ``sys.modules`` for the
import only.  Nothing from the reference is copied; on a machine without /root/reference
``load()`` returns None.
"""
import os
import sys
import types

REF_SRC = "/root/reference/src"


def available() -> bool:
    return os.path.isdir(REF_SRC)


def load():
    if not available():
        return None
    for name in ("cv2", "trimesh", "rembg", "pycocotools", "pycocotools.mask"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)
    if "pycocotools" in sys.modules and not hasattr(sys.modules["pycocotools"], "mask"):
        sys.modules["pycocotools"].mask = sys.modules["pycocotools.mask"]
    saved = list(sys.path)
    # the names util / util_3dbox / cam_utils are also this repo's drop-in module names:
    # import the reference ones under private aliases so they never shadow ours.
    import importlib.util

    mods = {}
    for name in ("util", "util_3dbox", "cam_utils"):
        spec = importlib.util.spec_from_file_location("_la3d_ref_" + name, os.path.join(REF_SRC, name + ".py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        mods[name] = m
    sys.path[:] = saved
    return types.SimpleNamespace(**mods)
