"""ctypes binding of oracle/liboracle.so (ipm_dense_ref.c) plus a numpy twin of one Newton step.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Reference semantics restated:
pycllp/cl/primal_normal.cl:201-284 (IPM loop), :122-156 (step), pycllp/cl/ldl.cl:314-378 (modified
LDL'), :505-537 (forward/backward), :602-653 (solve with refinement).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OracleOpts(ctypes.Structure):
    _fields_ = [("eps", ctypes.c_double), ("delta", ctypes.c_double), ("r", ctypes.c_double),
                ("pivot_floor", ctypes.c_double), ("refine_tol", ctypes.c_double),
                ("max_iter", ctypes.c_int), ("max_refine", ctypes.c_int), ("flags", ctypes.c_int)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, os.path.join(_HERE, "liboracle.so")])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.oracle_default_opts.argtypes = [ctypes.POINTER(OracleOpts)]
        _LIB.oracle_dense_solve.restype = ctypes.c_int
        _LIB.oracle_solve_primal_normal.restype = ctypes.c_int
    return _LIB


def default_opts(**kw):
    o = OracleOpts()
    lib().oracle_default_opts(ctypes.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise TypeError("unknown option %r" % k)
        setattr(o, k, v)
    if "max_refine" not in kw and (o.flags & 32):     # PYCLLP_FLAG_HSD: the product's default for that variant
        o.max_refine = 20                             # (pycllp_amd/_native.py HSD_MAX_REFINE, reasons there)
    return o


def _p(a, t=ctypes.c_double):
    return a.ctypes.data_as(ctypes.POINTER(t))


def dense_solve(A, b, c, nthreads=1, x0=None, y0=None, z0=None, **opts):
    """Solve max c'x s.t. Ax=b, x>=0 for every row of (b, c) with the shared dense A.

    Returns dict(x, y, z, pobj, dobj, status, iters, nrefs)."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    b = np.ascontiguousarray(np.atleast_2d(b), dtype=np.float64)
    c = np.ascontiguousarray(np.atleast_2d(c), dtype=np.float64)
    m, N = A.shape
    B = b.shape[0]
    assert b.shape == (B, m) and c.shape == (B, N)
    o = default_opts(**opts)
    x = np.ones((B, N)) if x0 is None else np.array(x0, dtype=np.float64, order="C")
    y = np.ones((B, m)) if y0 is None else np.array(y0, dtype=np.float64, order="C")
    z = np.ones((B, N)) if z0 is None else np.array(z0, dtype=np.float64, order="C")
    pobj = np.empty(B); dobj = np.empty(B)
    status = np.empty(B, dtype=np.int32); iters = np.empty(B, dtype=np.int32)
    nrefs = np.empty(B, dtype=np.int32)
    lib().oracle_dense_solve(ctypes.c_int(m), ctypes.c_int(N), _p(A), ctypes.c_long(B), _p(b), _p(c),
                             _p(x), _p(y), _p(z), _p(pobj), _p(dobj), _p(status, ctypes.c_int),
                             _p(iters, ctypes.c_int), _p(nrefs, ctypes.c_int), ctypes.byref(o),
                             ctypes.c_int(int(nthreads)))
    return dict(x=x, y=y, z=z, pobj=pobj, dobj=dobj, status=status, iters=iters, nrefs=nrefs)


def solve_primal_normal(A, x, z, y, b, c, mu, pivot_floor=1e-6):
    """One Newton step dy (ldl.cl:602-653) through the C restatement."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    m, N = A.shape
    args = [np.ascontiguousarray(v, dtype=np.float64) for v in (x, z, y, b, c)]
    dy = np.empty(m)
    lib().oracle_solve_primal_normal(ctypes.c_int(m), ctypes.c_int(N), _p(A), *[_p(v) for v in args],
                                     ctypes.c_double(mu), ctypes.c_double(pivot_floor), _p(dy))
    return dy


def ldl(A, modified=False, beta=None, delta=1e-6):
    """(Modified) LDL' of an explicit SPD matrix; returns dense unit-lower L and D (ldl.cl:28-107)."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    n = A.shape[0]
    Lp = np.zeros(n * (n + 1) // 2); D = np.zeros(n)
    if modified:
        if beta is None:
            beta = np.sqrt(np.abs(np.diag(A)).max())
        lib().oracle_modified_ldl(ctypes.c_int(n), _p(A), _p(Lp), _p(D), ctypes.c_double(beta),
                                  ctypes.c_double(delta))
    else:
        lib().oracle_ldl(ctypes.c_int(n), _p(A), _p(Lp), _p(D))
    L = np.zeros((n, n))
    L[np.tril_indices(n)] = Lp
    return L, D


# ---------------------------------------------------------------------------------------------
# numpy twin of the Newton step: the KNOWN-ANSWER formula the reference's own test uses
# (tests/test_ldl.py:196-216): dy = solve(A (x/z) A', -(b - A x - A (x/z)(c - A'y + mu/x)))
# ---------------------------------------------------------------------------------------------
def newton_step_known_answer(A, x, z, y, b, c, mu):
    d = x / z
    M = (A * d) @ A.T
    rhs = -(b - A @ x - (A * d) @ (c - A.T @ y + mu / x))
    return np.linalg.solve(M, rhs)
