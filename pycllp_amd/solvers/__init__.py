"""Solver plugin registry: the drop-in boundary.

API-identical mirror of the reference's ``pycllp/solvers/__init__.py:3-21``: any ``BaseSolver`` subclass
whose ``name`` is not ``None`` is inserted into ``solver_registry`` at class-creation time, and is driven
as ``lp.init(solver); lp.solve(solver)`` with results left in ``solver.x`` / ``solver.status``.  When a
real ``pycllp`` is importable, :func:`register_with_pycllp` also inserts the HIP solvers into
``pycllp.solvers.solver_registry`` so existing callers pick them up by name.
"""

solver_registry = {}


class MetaSolver(type):
    def __new__(mcs, clsname, bases, attrs):
        newclass = super(MetaSolver, mcs).__new__(mcs, clsname, bases, attrs)
        if newclass.name is not None:
            solver_registry[newclass.name] = newclass
        return newclass


class BaseSolver(MetaSolver("_BaseSolver", (object,), {"name": None})):
    name = None

    def init(self, lp, verbose=0):
        raise NotImplementedError()

    def solve(self, lp, verbose=0):
        raise NotImplementedError()


class BaseCSCSolver(BaseSolver):
    """Mirror of ``pycllp/solvers/__init__.py:24-26``: ``init`` caches the compressed-sparse-column arrays of ``lp.A``."""

    def init(self, lp, verbose=0):
        self.A, self.Ai, self.Ak = lp.A.tocsc_arrays()


class BaseGeneralSolver(BaseSolver):
    pass


def register_with_pycllp():
    """Insert this package's solvers into a real ``pycllp.solvers.solver_registry`` if importable."""
    try:
        from pycllp.solvers import solver_registry as theirs  # noqa
    except Exception:
        return False
    for k, v in solver_registry.items():
        theirs.setdefault(k, v)
    return True


from .hip import HipDensePrimalNormalSolver, HipSparsePrimalNormalSolver  # noqa: E402,F401
