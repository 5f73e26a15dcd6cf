"""Problem containers with the attribute surface the reference's solver plugins consume.

The reference's ``pycllp/lp.py`` "stays" (BASELINE.json north_star) but does not travel to the GPU
box, so this module carries a compact, independently written container exposing exactly what a
solver touches (``pycllp/solvers/cl.py:35-39,99,102``): ``nrows, ncols, nproblems, A.todense(), b, c, f``
plus ``init(solver)`` / ``solve(solver)`` (``pycllp/lp.py:531-535``) and ``StandardLP.to_equality_form()``
(``pycllp/lp.py:551-567``).  Broadcasting rules follow ``pycllp/lp.py:338-352``: a 1-D ``b`` is one
problem, a 1-D ``c`` is repeated for every problem, a scalar ``f`` is repeated, and mismatched problem
counts raise ``ValueError``.  A real ``pycllp`` LP object can be passed to the solvers instead: they
duck-type on the attributes above.
"""
import numpy as np
import scipy.sparse as sp


class SparseMatrix(object):
    """One shared sparsity pattern in coordinate form (``pycllp/lp.py:16-54``).

    ``data`` is ``[nproblems, nnz]``: one shared structure, per-problem values (``pycllp/lp.py:16-24``).  The reference's LP
    classes accept a single set of values only (``pycllp/lp.py:335-336``: "can only have a single problem in the current
    implementation"); here ``nproblems`` sets of values are accepted when there is one per problem of the LP, and the HIP
    solvers read them per LP."""

    def __init__(self, rows=None, cols=None, data=None, matrix=None):
        if matrix is not None:
            coo = sp.coo_matrix(matrix)
            self._rows = np.asarray(coo.row, dtype=np.int64)
            self._cols = np.asarray(coo.col, dtype=np.int64)
            self.data = np.asarray(coo.data, dtype=np.float64).reshape(1, -1)
            self._shape = coo.shape
        elif data is not None:
            if not (len(rows) == len(cols) == np.shape(data)[-1]):
                raise ValueError("Arrays rows, cols and data must be the same length.")
            self._rows = np.asarray(rows, dtype=np.int64)
            self._cols = np.asarray(cols, dtype=np.int64)
            self.data = np.atleast_2d(np.asarray(data, dtype=np.float64))
            self._shape = None
        else:
            self._rows = np.zeros(0, dtype=np.int64)
            self._cols = np.zeros(0, dtype=np.int64)
            self.data = np.zeros((1, 0))
            self._shape = None

    @property
    def nrows(self):
        n = int(self._rows.max()) + 1 if self._rows.size else 0
        return max(n, self._shape[0]) if self._shape else n

    @property
    def ncols(self):
        n = int(self._cols.max()) + 1 if self._cols.size else 0
        return max(n, self._shape[1]) if self._shape else n

    @property
    def nnzeros(self):
        return int(self._rows.size)

    @property
    def nproblems(self):
        return int(self.data.shape[0])

    def add_col(self, rows, values):
        """Append a column with the given row entries; returns its index."""
        col = self.ncols
        rows = np.atleast_1d(np.asarray(rows, dtype=np.int64))
        values = np.atleast_1d(np.asarray(values, dtype=np.float64))
        self._rows = np.concatenate([self._rows, rows])
        self._cols = np.concatenate([self._cols, np.full(rows.shape, col, dtype=np.int64)])
        self.data = np.concatenate([self.data, np.tile(values, (self.data.shape[0], 1))], axis=1)
        if self._shape:
            self._shape = (max(self._shape[0], int(rows.max()) + 1), col + 1)
        return col

    def tocoo(self, problem=0):
        return sp.coo_matrix((self.data[problem], (self._rows, self._cols)), shape=(self.nrows, self.ncols))

    def tocsc(self, problem=0):
        return self.tocoo(problem).tocsc()

    def tocsr(self, problem=0):
        return self.tocoo(problem).tocsr()

    def tocsc_arrays(self):
        """(values [nproblems, nnz], row index, column pointer) as ``pycllp/lp.py:289-299``."""
        csc = self.tocsc()
        csc.sort_indices()
        order = np.lexsort((self._rows, self._cols))
        return (np.ascontiguousarray(self.data[:, order]), csc.indices.astype(np.int32),
                csc.indptr.astype(np.int32))

    def todense(self, problem=0):
        return np.asarray(self.tocoo(problem).todense())


class EqualityLP(object):
    """maximise c'x + f  subject to  A x = b, x >= 0   (``pycllp/lp.py:306-330``)."""

    def __init__(self, A=None, b=None, c=None, f=None):
        if A is None:
            self.A = SparseMatrix()
            self.b = np.zeros((1, 0)); self.c = np.zeros((1, 0)); self.f = np.zeros(1)
            return
        if b is None or c is None or f is None:
            raise ValueError("If A matrix is provided then b, c and f must also be provided.")
        if not isinstance(A, SparseMatrix):
            A = SparseMatrix(matrix=A)
        self.A = A
        self.b = np.array(b, dtype=np.float64)
        if self.b.ndim == 1:
            self.b = self.b.reshape(1, -1)
        nprb = self.b.shape[0]
        if A.nproblems > 1 and A.nproblems != nprb:
            raise ValueError("A matrix holds %d sets of values for %d problems: one shared set or one per problem." % (A.nproblems, nprb))
        self.c = np.array(c, dtype=np.float64)
        if self.c.ndim == 1:
            self.c = np.tile(self.c, (nprb, 1))
        if self.c.shape[0] != nprb:
            raise ValueError("A matrix and c array do not have the same number of problems.")
        self.f = np.full(nprb, float(f)) if np.isscalar(f) else np.array(f, dtype=np.float64)

    nrows = property(lambda self: self.A.nrows)
    ncols = property(lambda self: self.A.ncols)
    nnzeros = property(lambda self: self.A.nnzeros)
    nproblems = property(lambda self: self.b.shape[0])
    m = nrows
    n = ncols

    def add_col(self, rows, values, obj):
        """Append a variable with objective coefficient ``obj`` (scalar or per-problem)."""
        col = self.A.add_col(rows, values)
        o = np.broadcast_to(np.asarray(obj, dtype=np.float64), (self.nproblems,)).reshape(-1, 1)
        self.c = np.concatenate([self.c, o], axis=1)
        return col

    def init(self, solver, verbose=0):
        solver.init(self, verbose=verbose)

    def solve(self, solver, verbose=0):
        return solver.solve(self, verbose=verbose)


class StandardLP(EqualityLP):
    """maximise c'x + f  subject to  A x <= b, x >= 0."""

    def to_equality_form(self):
        """Copy and append one unit slack column (objective 0) per row (``pycllp/lp.py:551-567``)."""
        A = SparseMatrix(self.A._rows.copy(), self.A._cols.copy(), self.A.data.copy())   # per-problem values survive
        A._shape = (self.nrows, self.ncols)
        lp = EqualityLP(A, self.b.copy(), self.c.copy(), self.f.copy())
        for row in range(self.nrows):
            lp.add_col([row], [1.0], 0.0)
        return lp


class GeneralLP(StandardLP):
    """optimise c'x + f  subject to  a <= A x <= b,  l <= x <= u   (``pycllp/lp.py:570-623``; ``a`` defaults to "no lower
    bound on the rows", ``l`` to 0, ``u`` to +inf; 1-D bounds are repeated for every problem)."""

    def __init__(self, A=None, b=None, c=None, a=None, l=None, u=None, f=None):
        super(GeneralLP, self).__init__(A=A, b=b, c=c, f=f)
        if A is None:
            self.a = np.zeros((1, 0)); self.l = np.zeros((1, 0)); self.u = np.zeros((1, 0))
            return
        nprb = self.nproblems

        def rows_of(v, like, default):
            if v is None:
                return np.full(like.shape, default)
            v = np.array(v, dtype=np.float64)
            return np.tile(v, (nprb, 1)) if v.ndim == 1 else v

        self.a = rows_of(a, self.b, -np.inf)      # the reference stores +inf and drops the row again (lp.py:597-599, 511-529)
        self.l = rows_of(l, self.c, 0.0)
        self.u = rows_of(u, self.c, np.inf)

    def to_standard_form(self):
        """The StandardLP  max c'x + f', A' x <= b', x >= 0  with the same optimum (``pycllp/lp.py:725-792``):
        shift x <- x - l, write a <= A x as -A x <= -a, add x <= u - l for the finite upper bounds, and drop every row
        without a finite bound (``remove_unbounded``, ``pycllp/lp.py:511-529``).  A row, or an upper bound, that is finite
        for some problems of the batch and infinite for others raises ``ValueError`` exactly as the reference's
        ``remove_unbounded`` does (``lp.py:518-523``): a stand-in "big" bound would enter the relative tolerances
        (eps * (1 + |b|)) and the autoscale factors of that LP and ruin its solve."""
        if np.isneginf(self.l).any():
            raise ValueError('Lower bounds (l) contains -inf.')
        m, n, B = self.nrows, self.ncols, self.nproblems
        b, a, c, f = self.b.copy(), self.a.copy(), self.c.copy(), self.f.copy()
        l = self.l
        u = self.u - np.where(np.isfinite(self.u), l, 0.0)
        for k in range(B):
            Ak = self.A.tocsr(k if self.A.nproblems > 1 else 0)
            Al = Ak @ l[k]
            b[k] -= Al; a[k] -= Al
            f[k] += c[k] @ l[k]
        keep_lo = np.isfinite(a).any(axis=0)          # rows  -A x <= -a
        keep_hi = np.isfinite(b).any(axis=0)          # rows   A x <=  b
        keep_ub = np.isfinite(u).any(axis=0)          # rows     x <=  u - l
        for what, bound, keep in (("lower row bound a", a, keep_lo), ("row bound b", b, keep_hi), ("upper bound u", u, keep_ub)):
            mixed = keep & ~np.isfinite(bound).all(axis=0)
            if mixed.any():
                raise ValueError("Can not remove unbounded rows. %s of index %d is unbounded for some problems of the batch "
                                 "but not all." % (what, int(np.flatnonzero(mixed)[0])))
        rows, cols, data = [], [], []
        rhs = []
        r0 = 0
        for sign, keep, bound in ((-1.0, keep_lo, -a), (1.0, keep_hi, b)):
            newrow = np.cumsum(keep) - 1 + r0
            sel = keep[self.A._rows]
            rows.append(newrow[self.A._rows[sel]]); cols.append(self.A._cols[sel]); data.append(sign * self.A.data[:, sel])
            rhs.append(bound[:, keep])
            r0 += int(keep.sum())
        ub = np.flatnonzero(keep_ub)
        rows.append(r0 + np.arange(ub.size)); cols.append(ub); data.append(np.ones((self.A.data.shape[0], ub.size)))
        rhs.append(u[:, ub])
        bb = np.concatenate(rhs, axis=1)
        A2 = SparseMatrix(np.concatenate(rows), np.concatenate(cols), np.concatenate(data, axis=1))
        A2._shape = (bb.shape[1], n)
        return StandardLP(A2, bb, c, f)
