"""Diagnostic: the LPs of config 5's share that the HSD wave kernel does not finish as optimal."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pycllp_amd import problems, _native
from pycllp_amd.lp import SparseMatrix, StandardLP
from pycllp_amd.solvers import solver_registry
B = 16384
A, b, c = problems.random_sparse_arrays(128, 256, B, density=0.025, seed=0)
lp = StandardLP(SparseMatrix(matrix=A), b, c, 0.0).to_equality_form()
for fl, name in ((0, "wave"), (_native.FLAG_BLOCK_KERNEL, "block")):
    s = solver_registry["hip_sparse_primal_normal"](hsd=True, flags=fl)
    lp.init(s); st = lp.solve(s).copy()
    bad = np.nonzero(st != 0)[0]
    print(name, "non-optimal:", bad, st[bad], s.iters[bad], "iters of 7557:", s.iters[7557], "max iters", s.iters.max())
