"""Development aid: the 'dense 60x700' HSD case of test_large_lps against the oracle, iteration counts LP by LP (and twice, to see
whether the kernel is deterministic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pycllp_amd import problems
from pycllp_amd.lp import SparseMatrix, StandardLP
from pycllp_amd.solvers import solver_registry
from oracle import port
m, n, B = 60, 700, 12
A, b, c = problems.random_dense_arrays(m, n, B, seed=m)
lp = StandardLP(SparseMatrix(matrix=A), b, c, 0.0).to_equality_form()
for rep in range(3):
    s = solver_registry["hip_dense_primal_normal"](hsd=True)
    lp.init(s)
    st = lp.solve(s)
    print("run", rep, "kernel", s.launch_info()["kernel"], "status", st.tolist(), "iters", s.iters.tolist(), "pobj[0] %.15e" % s.primal_obj[0])
from pycllp_amd.solvers.hip import autoscale_wanted
fl = 32 | (8 if autoscale_wanted(lp.b, lp.c) else 0)
r = port.dense_solve(lp.A.todense(), lp.b, lp.c, nthreads=8, flags=fl)
print("oracle (flags %d) iters" % fl, r["iters"].tolist(), "pobj[0] %.15e" % r["pobj"][0])
