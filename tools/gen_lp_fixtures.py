"""BUILD-CONTAINER ONLY: fixtures for GeneralLP.to_standard_form from the reference's own lp.py.

Imports pycllp.lp from /root/reference (read-only) with the shims of tools/check_reference_boundary.py, builds the cases of
the reference's tests/test_lp.py:225-262 (test_gte_conversion) and a few more of the same kind -- row bounds on either
or both sides, lower bounds l > 0, objective offset; upper bounds u = +inf, the branch of pycllp/lp.py:725-792 that runs
(the finite-u branch indexes a 2-D array with row indices and stacks a scipy matrix under a SparseMatrix: it raises) --
and stores inputs + the StandardLP the reference returns (dense A, b, c, f) in tests/golden/general_lp.npz."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.check_reference_boundary import import_reference  # noqa: E402


def main():
    import_reference()
    from pycllp.lp import GeneralLP
    rs = np.random.RandomState(7)
    cases = {}
    # tests/test_lp.py:236-250
    cases["gte"] = dict(rows=[([0, 1, 2], [[1.0, 1.0, 1.0]], 2.0, np.inf)], obj=[0.0, 0.0, 0.0], l=None)
    cases["two_sided"] = dict(rows=[([0, 1], [[1.0, 2.0]], 1.0, 4.0), ([1, 2], [[3.0, 1.0]], -np.inf, 5.0),
                                    ([0, 2], [[1.0, 1.0]], 0.5, np.inf)], obj=[1.0, 2.0, 0.5], l=None)
    A = rs.rand(5, 4)
    lo = np.array([0.1, -np.inf, 0.3, -np.inf, 0.2]); hi = np.array([3.0, 2.5, np.inf, 4.0, 3.5])
    cases["rand5x4"] = dict(rows=[(list(range(4)), [list(A[i])], lo[i], hi[i]) for i in range(5)], obj=list(rs.rand(4)),
                            l=[0.05, 0.0, 0.2, 0.1])
    out = {}
    for key, cs in cases.items():
        lp = GeneralLP()
        for cols, vals, lb, ub in cs["rows"]:
            lp.add_row(cols, np.array(vals), lb, ub)
        for j, o in enumerate(cs["obj"]):
            lp.set_objective(j, o)
        if cs["l"] is not None:
            # GeneralLP.set_col_bounds raises in the reference (np.neginf, lp.py:684): set the attribute it would set
            lp.l = np.array([cs["l"]], dtype=np.float64)
            lp.u = np.full((1, len(cs["l"])), np.inf)
        slp = lp.to_standard_form()
        nr = len(cs["rows"]); nc = len(cs["obj"])
        Ain = np.zeros((nr, nc))
        for i, (cols, vals, lb, ub) in enumerate(cs["rows"]):
            Ain[i, cols] = np.asarray(vals)[0]
        out[key + "_A"] = Ain
        out[key + "_a"] = np.array([r[2] for r in cs["rows"]]); out[key + "_b"] = np.array([r[3] for r in cs["rows"]])
        out[key + "_c"] = np.array(cs["obj"]); out[key + "_l"] = np.array(cs["l"] if cs["l"] is not None else [0.0] * nc)
        out[key + "_std_A"] = np.asarray(slp.A.todense(), dtype=np.float64)
        out[key + "_std_b"] = np.asarray(slp.b, dtype=np.float64); out[key + "_std_c"] = np.asarray(slp.c, dtype=np.float64)
        out[key + "_std_f"] = np.asarray(slp.f, dtype=np.float64)
        print(key, "->", out[key + "_std_A"].shape, out[key + "_std_b"], out[key + "_std_f"])
    path = os.path.join(ROOT, "tests", "golden", "general_lp.npz")
    np.savez_compressed(path, keys=np.array(sorted(cases)), **out)
    print("wrote", path)


if __name__ == "__main__":
    main()
