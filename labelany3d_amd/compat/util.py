"""Drop-in shim: put labelany3d_amd/compat first on sys.path (the reference resolves its modules by
bare name from cwd src/, reference src/batch_scripts/whole.py:10,15-16) and
`from util import ...` resolves to the MI355X implementation."""
from labelany3d_amd.util import *  # noqa: F401,F403
from labelany3d_amd import util as _impl

globals().update({k: getattr(_impl, k) for k in dir(_impl) if k.startswith("_estimate")})
