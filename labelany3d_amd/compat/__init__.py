"""Bare-name shims for the reference's geometry modules and the monkey-patch installer.

The reference resolves ``util``, ``util_3dbox`` and ``cam_utils`` by bare name from cwd ``src/`` — and its stage scripts put
``'./'`` FIRST on ``sys.path`` themselves (reference src/batch_scripts/whole.py:10), so a path inserted earlier is shadowed by
the reference's own files.  ``install()`` therefore does not fight over ``sys.path``: it imports whatever ``util`` /
``util_3dbox`` / ``cam_utils`` resolve to and replaces only the hot-path functions with the MI355X implementations."""
import importlib
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_HOT = {
    "util": ("depth_to_points",),
    "util_3dbox": ("estimate_bbox", "_estimate_yaw_pca", "_estimate_yaw_convex_hull", "save_3d_with_ground_alignment_bbox"),
    "cam_utils": (),
}
_loaded = {}


def _reference_module(name):
    """The reference's own ``<name>.py``: the first one on ``sys.path`` that is not this shim directory."""
    if name in _loaded:
        return _loaded[name]
    for d in sys.path:
        d = d or "."
        f = os.path.join(d, name + ".py")
        if os.path.isfile(f) and os.path.abspath(d) != _HERE:
            spec = importlib.util.spec_from_file_location(f"_la3d_reference_{name}", f)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            _loaded[name] = mod
            return mod
    raise AttributeError(f"no reference {name}.py found on sys.path behind the labelany3d_amd shim")


def install(modules=("util", "util_3dbox")):
    """Import the reference's modules by bare name (as its stage scripts do) and replace their hot-path functions in place.
    Call after the script has set ``sys.path`` and before its ``from util import ...`` lines.  Returns the patched names."""
    patched = []
    for name in modules:
        mod = sys.modules.get(name)
        if mod is None or os.path.dirname(os.path.abspath(getattr(mod, "__file__", ""))) == _HERE:
            mod = importlib.import_module(name)
        if os.path.dirname(os.path.abspath(getattr(mod, "__file__", ""))) == _HERE:
            continue   # the shim itself resolved: nothing to patch
        impl = importlib.import_module(f"labelany3d_amd.{name}")
        for fn in _HOT[name]:
            setattr(mod, fn, getattr(impl, fn))
            patched.append(f"{name}.{fn}")
    return patched
