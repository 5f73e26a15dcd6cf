"""Solver plugin registry: the drop-in boundary.

API-identical mirror of the reference's ``pycllp/solvers/__init__.py:3-21``: any ``BaseSolver`` subclass
whose ``name`` is not ``None`` is inserted into ``solver_registry`` at class-creation time, and is driven
as ``lp.init(solver); lp.solve(solver)`` with results left in ``solver.x`` / ``solver.status``.  When a
real ``pycllp`` is importable, :func:`register_with_pycllp` also inserts the HIP solvers into
``pycllp.solvers.solver_registry`` so existing callers pick them up by name.
"""

solver_registry = {}


def _announce(cls):
    """A solver class with a ``name`` is reachable as ``solver_registry[name]`` from the moment it exists."""
    key = getattr(cls, "name", None)
    if key is not None:
        solver_registry[key] = cls
    return cls


class MetaSolver(type):
    """The reference's metaclass name (``pycllp/solvers/__init__.py:6-11``); here the registration is done once the class
    object is complete (``__init__``), through :func:`_announce`."""

    def __init__(cls, clsname, bases, namespace):
        type.__init__(cls, clsname, bases, namespace)
        _announce(cls)


class BaseSolver(metaclass=MetaSolver):
    """Plugin contract (``pycllp/solvers/__init__.py:14-21``): ``init(lp)`` once per constraint matrix, ``solve(lp)`` per batch."""
    name = None

    def init(self, lp, verbose=0):
        raise NotImplementedError("%s does not implement init()" % type(self).__name__)

    def solve(self, lp, verbose=0):
        raise NotImplementedError("%s does not implement solve()" % type(self).__name__)


class BaseCSCSolver(BaseSolver):
    """Mirror of ``pycllp/solvers/__init__.py:24-26``: ``init`` caches the compressed-sparse-column arrays of ``lp.A``
    as ``self.A`` (values), ``self.Ai`` (row indices), ``self.Ak`` (column starts)."""

    def init(self, lp, verbose=0):
        values, rows, starts = lp.A.tocsc_arrays()
        self.A, self.Ai, self.Ak = values, rows, starts


class BaseGeneralSolver(BaseSolver):
    """Marker base of solvers that take a ``GeneralLP`` (``pycllp/solvers/__init__.py:29-30``)."""


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
