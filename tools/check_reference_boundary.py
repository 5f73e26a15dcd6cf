"""BUILD-CONTAINER ONLY: the drop-in boundary checked against the reference's own objects.

Imports the reference package from /root/reference (read-only; nothing is copied) with the harness shims SURVEY 8c-3
lists -- numpy aliases removed in numpy >= 1.24, and stand-in modules for what cannot be imported here (pyopencl, the
un-built Cython extensions ``pycllp._ldl`` and ``pycllp.solvers.cython_glpk``) -- and checks

  1. ``pycllp_amd.solvers.register_with_pycllp()`` puts the HIP solvers into ``pycllp.solvers.solver_registry`` under
     their names, next to the reference's five, and they are subclasses compatible with ``BaseSolver``'s contract
     (``init(lp, verbose=0)``, ``solve(lp, verbose=0)``, ``pycllp/solvers/__init__.py:14-21``);
  2. reference-built ``StandardLP(...).to_equality_form()`` / ``EqualityLP`` objects expose exactly what the HIP hosts
     consume (``HipDensePrimalNormalSolver.consume``): nrows, ncols, nproblems, A.todense(), b, c, f -- and that this
     package's own containers (``pycllp_amd/lp.py``), built from the same raw inputs, expose the SAME values;
  3. ``lp.init(solver)`` / ``lp.solve(solver)`` of the REFERENCE LP class dispatch to the solver's methods
     (``pycllp/lp.py:531-535``), with a recording stand-in for the device part (there is no GPU in this container).

It writes tests/golden/reference_lp_surface.npz: the raw inputs and the surface the reference objects showed.  The GPU
tests replay it (tests/test_hip_parity.py::test_reference_lp_surface_replay): same inputs -> same surface -> solve ->
oracle parity.  The reference itself never travels.
"""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)


def import_reference():
    import scipy.sparse, scipy.linalg  # noqa: F401  (before the aliases: SURVEY 8c-3)
    if not hasattr(np, "int"):
        np.int = int
    if not hasattr(np, "float"):
        np.float = float
    if not hasattr(np, "product"):
        np.product = np.prod
    cl = types.ModuleType("pyopencl")
    cl.get_platforms = lambda: []
    sys.modules.setdefault("pyopencl", cl)
    ldl = types.ModuleType("pycllp._ldl")                 # Cython extension, not built here; not on the checked path
    ldl.solve_primal_normal = ldl.factor_primal_normal = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("stub"))
    sys.modules.setdefault("pycllp._ldl", ldl)
    glpk = types.ModuleType("pycllp.solvers.cython_glpk")  # needs libglpk, absent offline
    glpk.glpk_solve = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("stub"))
    sys.modules.setdefault("pycllp.solvers.cython_glpk", glpk)
    sys.path.insert(0, REF)
    import pycllp  # noqa: F401
    return pycllp


def cases():
    """Raw inputs: the reference's own textbook problems (data as in tests/vanderbei_problems.py:5-36,
    tests/test_simple.py:17-42) and a random batch of the SURVEY 8d generator."""
    from pycllp_amd import problems
    out = {}
    A = np.array([[0.0, 2.0, 3.0], [1.0, 1.0, 2.0], [1.0, 2.0, 3.0]])
    out["v29"] = dict(kind="standard", A=A, b=np.array([5.0, 4.0, 7.0]), c=np.array([2.0, 3.0, 4.0]), f=0.0)
    out["v210"] = dict(kind="equality", A=np.ones((1, 4)), b=np.array([1.0]), c=np.array([6.0, 8.0, 5.0, 9.0]), f=0.0)
    A, b, c = problems.parallel_small_problem_arrays(32)
    out["small32"] = dict(kind="standard", A=A, b=b, c=c, f=0.0)
    A, b, c = problems.random_dense_arrays(8, 12, 16, seed=0)
    out["rand8x12"] = dict(kind="standard", A=A, b=b, c=c, f=1.25)
    return out


def build(module_lp, case):
    import scipy.sparse as sp
    SM = module_lp.SparseMatrix
    A = SM(matrix=sp.coo_matrix(case["A"]))
    if case["kind"] == "standard":
        return module_lp.StandardLP(A, case["b"], case["c"], case["f"]).to_equality_form()
    return module_lp.EqualityLP(A, case["b"], case["c"], case["f"])


def main():
    pycllp = import_reference()
    import pycllp.lp as ref_lp
    from pycllp.solvers import solver_registry as theirs, BaseSolver as TheirBase
    import pycllp_amd
    from pycllp_amd import lp as our_lp
    from pycllp_amd.solvers import register_with_pycllp, solver_registry as ours
    from pycllp_amd.solvers.hip import HipDensePrimalNormalSolver

    # 1. registration
    before = set(theirs)
    assert register_with_pycllp() is True
    for name in ("hip_dense_primal_normal", "hip_sparse_primal_normal"):
        assert name in theirs and theirs[name] is ours[name], name
        for meth in ("init", "solve"):
            assert callable(getattr(theirs[name], meth))
    assert before <= set(theirs) and {"cl_dense_primal_normal", "dense_primal_normal"} <= before
    print("registry:", sorted(theirs))

    # 2. attribute surface of reference-built LP objects == what the HIP host consumes == our containers' surface
    fixture = {}
    for key, case in cases().items():
        rlp, olp = build(ref_lp, case), build(our_lp, case)
        rs, os_ = HipDensePrimalNormalSolver.consume(rlp), HipDensePrimalNormalSolver.consume(olp)
        for k in ("m", "n", "nproblems"):
            assert rs[k] == os_[k], (key, k, rs[k], os_[k])
        for k in ("A", "b", "c", "f"):
            assert rs[k].dtype == np.float64 and rs[k].shape == os_[k].shape, (key, k, rs[k].shape, os_[k].shape)
            np.testing.assert_array_equal(rs[k], os_[k])
        for k, v in case.items():
            fixture["%s_in_%s" % (key, k)] = np.asarray(v)
        for k in ("A", "b", "c", "f"):
            fixture["%s_%s" % (key, k)] = rs[k]
        fixture["%s_shape" % key] = np.array([rs["m"], rs["n"], rs["nproblems"]])
        print("surface %-9s m=%d n=%d nproblems=%d  identical to pycllp_amd.lp" % (key, rs["m"], rs["n"], rs["nproblems"]))

    # 3. the reference LP class drives a HIP solver object through lp.init / lp.solve (device part recorded, not run)
    calls = []

    class Recording(theirs["hip_dense_primal_normal"]):
        name = None

        def init(self, lp, verbose=0):
            calls.append(("init", HipDensePrimalNormalSolver.consume(lp)["n"], verbose))

        def solve(self, lp, verbose=0):
            calls.append(("solve", lp.nproblems, verbose))
            return "status"

    assert issubclass(Recording, ours["hip_dense_primal_normal"]) and not issubclass(Recording, TheirBase)
    rlp = build(ref_lp, cases()["small32"])
    rec = Recording()
    rlp.init(rec, verbose=1)
    assert rlp.solve(rec) == "status"
    assert calls == [("init", 5, 1), ("solve", 32, 0)], calls
    print("dispatch through pycllp.lp.EqualityLP.init/solve: ok")

    path = os.path.join(ROOT, "tests", "golden", "reference_lp_surface.npz")
    np.savez_compressed(path, keys=np.array(sorted(cases())), **fixture)
    print("wrote", path)


if __name__ == "__main__":
    main()
