"""PCIe-inclusive timing of the plugin path lp.solve(solver) (host numpy in, host numpy out) -- never the bench value."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pycllp_amd import problems
from pycllp_amd.lp import SparseMatrix, EqualityLP
from pycllp_amd.solvers import solver_registry
A, b, c = problems.random_dense_arrays(32, 64, 65536)
Ae, be, ce = problems.equality_arrays(A, b, c)
lp = EqualityLP(SparseMatrix(matrix=Ae), be, ce, 0.0)
s = solver_registry["hip_dense_primal_normal"]()
lp.init(s); lp.solve(s)
ts = []
for _ in range(5):
    t = time.perf_counter(); lp.solve(s); ts.append(time.perf_counter() - t)
print("lp.solve() host-to-host: median %.1f ms -> %.2f M LPs/s (65536 LPs, 67 MB in / 119 MB out over PCIe, pageable memory)"
      % (1e3 * np.median(ts), 65536 / np.median(ts) / 1e6))
