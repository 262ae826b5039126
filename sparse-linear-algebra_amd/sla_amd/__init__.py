"""sla_amd -- MI355X (gfx950) backend for the SpMV / CGS / BiCGSTAB / Arnoldi hot path of
ocramz/sparse-linear-algebra, mirroring the names of Numeric.LinearAlgebra.Sparse.

The compute path is libsla_hip.so (hand-written HIP); importing this package without the built
library raises ImportError -- there is no CPU fallback."""
from . import _lib
from ._lib import (IndexOutOfBounds, IterationException, MatVecSizeMismatchException, NeedsPivoting, SlaError,  # noqa: F401
                   SolveInfo, SolveOpts, build)
from .api import *  # noqa: F401,F403
from .api import (BICGSTAB_, BCG_, CGNE_, CGS_, GMRES_, Context, DeviceVector, LinSolveMethod, SpMatrix,  # noqa: F401
                  SpVector, default_context, set_default_context)
from . import workloads  # noqa: F401

_lib.lib()  # fail loudly at import time if the HIP library is missing
