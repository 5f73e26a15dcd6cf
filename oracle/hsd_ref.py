"""ctypes binding of oracle/_ref/libhsd_ref.so -- the REFERENCE'S OWN CPU solver.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  The library is the reference's Vanderbei
homogeneous-self-dual solver (`pycllp/ipo.py:3` -> `_ipo.hsd_solver` -> `pycllp/ipo/hsd.c:27 solver()`),
built by oracle/Makefile from the sources where they lie under /root/reference.  Call convention as
the reference's own wrapper (`pycllp/cyipo.pyx:7-16`, `pycllp/solvers/cython.py:20-26`): CSC arrays of
the StandardLP  max c'x s.t. Ax <= b, x >= 0 ; `inv_clo()` after every solve because the C code keeps
global static factorisation state (`ipo/ldlt.c:126`) -- NOT thread-safe, one solve at a time per process.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(_HERE, "_ref", "libhsd_ref.so")
_LIB = None
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)


def available():
    return os.path.exists(PATH)


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(PATH)
        _LIB.solver.argtypes = [ctypes.c_int] * 3 + [_ip, _ip, _dp, _dp, _dp, ctypes.c_double,
                                                     _dp, _dp, _dp, _dp, ctypes.c_int]
        _LIB.solver.restype = ctypes.c_int
        _LIB.inv_clo.restype = None
    return _LIB


def dense_to_csc(A):
    """Dense [m,n] -> (values, row indices, column pointers), all entries kept (lp.py:289-299)."""
    A = np.asarray(A, dtype=np.float64)
    m, n = A.shape
    vals = np.ascontiguousarray(A.T).ravel()
    iA = np.tile(np.arange(m, dtype=np.int32), n)
    kA = (np.arange(n + 1) * m).astype(np.int32)
    return vals, iA, kA


def sparse_to_csc(A):
    """scipy sparse [m,n] -> (values, row indices, column pointers) of its structural non-zeros: what the reference's
    ``SparseMatrix.tocsc_arrays`` hands the solver for a sparse A (lp.py:289-299 walks the stored entries only)."""
    import scipy.sparse as sp
    C = sp.csc_matrix(A)
    C.sort_indices()
    return (np.ascontiguousarray(C.data, dtype=np.float64), np.ascontiguousarray(C.indices, dtype=np.int32),
            np.ascontiguousarray(C.indptr, dtype=np.int32))


def solve_standard(A, b, c, f=0.0):
    """Solve the batch of StandardLPs max c_i'x s.t. A x <= b_i, x >= 0 one by one.  ``A``: dense array (every entry is
    handed over, as for a dense reference LP) or a scipy sparse matrix (structural non-zeros only).

    Returns dict(x[B,n], y[B,m], w[B,m], z[B,n], pobj, dobj, status)."""
    L = lib()
    sparse = hasattr(A, "tocsc")
    if not sparse:
        A = np.asarray(A, dtype=np.float64)
    b = np.ascontiguousarray(np.atleast_2d(b), dtype=np.float64)
    c = np.ascontiguousarray(np.atleast_2d(c), dtype=np.float64)
    m, n = A.shape
    B = b.shape[0]
    vals, iA, kA = sparse_to_csc(A) if sparse else dense_to_csc(A)
    x = np.empty((B, n)); y = np.empty((B, m)); w = np.empty((B, m)); z = np.empty((B, n))
    status = np.empty(B, dtype=np.int32)
    P = lambda a: a.ctypes.data_as(_dp)
    for i in range(B):
        bi = b[i].copy(); ci = c[i].copy()
        status[i] = L.solver(m, n, len(vals), iA.ctypes.data_as(_ip), kA.ctypes.data_as(_ip), P(vals),
                             P(bi), P(ci), float(f), P(x[i]), P(y[i]), P(w[i]), P(z[i]), 0)
        L.inv_clo()
    pobj = np.einsum("ij,ij->i", c, x) + f
    dobj = np.einsum("ij,ij->i", b, y) + f
    return dict(x=x, y=y, w=w, z=z, pobj=pobj, dobj=dobj, status=status)
