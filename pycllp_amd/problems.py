"""Synthetic LP batches (generator semantics of the reference's
``examples/random_problem.py:12-27`` and ``tests/helpers.py:35-60``, made scipy-independent so the GPU box
regenerates them bit-identically with numpy alone) and the reference's textbook problems as data."""
import numpy as np

from .lp import SparseMatrix, StandardLP, EqualityLP


def random_dense_arrays(m, n, nproblems, seed=0, shard=0):
    """Shared dense A~U[0,1) [m,n]; b~U[0.5,1.5) [B,m]; c~U[0.5,1.5) [B,n].

    Every such StandardLP (max c'x, Ax<=b, x>=0) is feasible (x=0) and bounded (A>0, c>0).
    ``shard`` 0 is exactly the SURVEY section 8d stream (A, then b, then c from RandomState(seed));
    shard k>0 keeps the same A and draws its own b, c, so ranks hold disjoint slices of one global
    batch without any rank generating all of it."""
    rs = np.random.RandomState(seed)
    A = rs.rand(m, n)
    if shard:
        rs = np.random.RandomState(seed + 1000003 * int(shard))
    b = 0.5 + rs.rand(nproblems, m)
    c = 0.5 + rs.rand(nproblems, n)
    return A, b, c


def equality_arrays(A, b, c):
    """[A | I], [c | 0]: dense arrays of ``StandardLP.to_equality_form()`` (``pycllp/lp.py:551-567``)."""
    m = A.shape[0]
    Ae = np.hstack([A, np.eye(m)])
    ce = np.hstack([c, np.zeros((c.shape[0], m))])
    return Ae, b, ce


def random_standard_lp(m, n, nproblems, seed=0):
    A, b, c = random_dense_arrays(m, n, nproblems, seed)
    return StandardLP(SparseMatrix(matrix=A), b, c, 0.0)


def vanderbei_2_9():
    """Data of the reference's ``tests/vanderbei_problems.py:5-19`` (StandardLP 3x3)."""
    A = np.array([[0.0, 2.0, 3.0], [1.0, 1.0, 2.0], [1.0, 2.0, 3.0]])
    b = np.array([5.0, 4.0, 7.0])
    c = np.array([2.0, 3.0, 4.0])
    return StandardLP(SparseMatrix(matrix=A), b, c, 0.0), np.array([1.5, 2.5, 0.0])


def vanderbei_2_10():
    """Data of the reference's ``tests/vanderbei_problems.py:22-36`` (EqualityLP 1x4)."""
    A = np.ones((1, 4))
    return EqualityLP(SparseMatrix(matrix=A), np.array([1.0]), np.array([6.0, 8.0, 5.0, 9.0]), 0.0), \
        np.array([0.0, 0.0, 0.0, 1.0])


def small_problem_arrays():
    """Data of the reference's ``tests/test_simple.py:17-30`` (2x3, '<=' rows)."""
    A = np.array([[3.0, 2.0, 1.0], [2.0, 5.0, 3.0]])
    c = np.array([1.10685436, 3.67678309, 2.04570983])
    b = np.array([5.187898, 16.76453246])
    return A, b, c


def parallel_small_problem_arrays(nproblems=32):
    """``tests/test_simple.py:33-42``: np.random.seed(0); b,c scaled by U[0.5,1.5)."""
    A, b, c = small_problem_arrays()
    rs = np.random.RandomState(0)
    bb = (0.5 + rs.rand(nproblems, len(b))) * b
    cc = (0.5 + rs.rand(nproblems, len(c))) * c
    return A, bb, cc


def random_sparse_arrays(m, n, nproblems, density=0.025, seed=0):
    """BASELINE config 5 generator: A = scipy.sparse.random(m, n, density) with every row forced
    to >= 3 non-zeros (``tests/helpers.py:47``) and -- so that the LP max c'x, Ax <= b, x >= 0 with c > 0 is bounded --
    every column to >= 1; values U[0,1); b, c ~ U[0.5,1.5).  Returns (A csr [m,n], b, c)."""
    import scipy.sparse as sp
    rs = np.random.RandomState(seed)
    A = sp.random(m, n, density=density, random_state=rs, format="lil")
    for i in range(m):
        need = min(3, n) - len(A.rows[i])
        while need > 0:
            j = int(rs.randint(n))
            if A[i, j] == 0:
                A[i, j] = rs.rand()
                need -= 1
    col_nnz = np.asarray((sp.csc_matrix(A) != 0).sum(axis=0)).ravel()
    for j in np.where(col_nnz == 0)[0]:
        A[int(rs.randint(m)), int(j)] = rs.rand()
    A = sp.csr_matrix(A)
    b = 0.5 + rs.rand(nproblems, m)
    c = 0.5 + rs.rand(nproblems, n)
    return A, b, c


def per_problem_values(A, nproblems, seed=7):
    """Per-problem values on the structure of the sparse matrix ``A`` (``SparseMatrix.data[nproblems, nnz]``,
    ``pycllp/lp.py:16-54``): LP k's value of entry e is the shared one times U[0.75, 1.25), drawn row-major from
    RandomState(seed) -- the first LPs of a batch are the same whatever its size.  Returns (rows, cols, data[B, nnz]) in
    the coordinate order of ``A.tocoo()``."""
    Ac = A.tocoo()
    data = Ac.data[None, :] * (0.75 + 0.5 * np.random.RandomState(seed).rand(nproblems, Ac.nnz))
    return Ac.row, Ac.col, data
