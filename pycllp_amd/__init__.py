"""pycllp_amd -- MI355X-native batched interior-point LP solver behind the pycllp solver-plugin API.

Only the hot path of jetuk/pycllp is rebuilt here: the batched dense
primal-normal-equations IPM that the reference runs as OpenCL (pycllp/cl/*.cl hosted by
pycllp/solvers/cl.py).  The product is ``csrc/libpycllp_hip.so`` (hand-written HIP for gfx950, plain
C ABI declared in include/pycllp_hip.h) plus this thin Python host mirroring the reference's plugin
interface.  There is no CPU fallback: without the HIP library or a GPU the solvers raise.
"""
from . import lp, problems, solvers, ldl  # noqa: F401
from .solvers import solver_registry, BaseSolver  # noqa: F401

__version__ = "0.1.0"
