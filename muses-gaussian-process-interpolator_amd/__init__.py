"""mgpi on MI355X: the dense Gaussian-process hot path of reedessick/muses-gaussian-process-interpolator
(kernel-matrix build -> Cholesky factor/solve of K(x,x)+nugget -> posterior mean / variance) as hand-written
HIP for gfx950 behind the reference's own Python API.

The directory name carries a hyphen (it is the repository's required package name); import it as ``mgpi_amd``
(the shim next to it) - ``import mgpi_amd as mgpi`` is the drop-in spelling.  The namespace is flat like the
reference's (mgpi/__init__.py:7-10).
"""
from .utils import *            # noqa: F401,F403
from .kernel_objects import *   # noqa: F401,F403
from .gp import *               # noqa: F401,F403
from .nngp import *             # noqa: F401,F403
from .config import *           # noqa: F401,F403
from . import _lib              # noqa: F401
from ._lib import Context, DeviceMatrix, MgpiError, NotPositiveDefinite, default_context, set_default_context  # noqa: F401

__version__ = '0.1.0'
