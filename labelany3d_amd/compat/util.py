"""Drop-in shim for the reference's ``util`` module.

Two ways to use it:
  * ``import labelany3d_amd.compat as c; c.install()`` right after the stage script has set up ``sys.path`` (reference
    src/batch_scripts/whole.py:10) and BEFORE its ``from util import ...`` lines: the reference's own modules are imported
    and only their hot-path functions are replaced, so every other helper keeps working;
  * or put this directory first on ``sys.path``: ``from util import ...`` then resolves here, and every name this module
    does not define is fetched lazily from the reference's own ``util.py`` found further down ``sys.path``
    (module ``__getattr__``, PEP 562), so ``from util import restore_mask_from_crop, depth_to_points`` works.
"""
from labelany3d_amd.util import depth_to_points  # noqa: F401   (the hot-path function only: the overlay / helpers stay the reference's own)
from labelany3d_amd import util as _impl
from labelany3d_amd.compat import _reference_module

globals().update({k: getattr(_impl, k) for k in dir(_impl) if k.startswith("_estimate")})


def __getattr__(attr):
    if attr.startswith("__"):
        raise AttributeError(attr)
    return getattr(_reference_module("util"), attr)
