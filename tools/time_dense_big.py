"""Timing of dense LPs beyond the lane-group kernel (the reference's test_ldl.py shape m = 100, n = 80 -> N = 180, and m = 64, n = 64)
through hip_dense_primal_normal (which hands them to the sparse machinery)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pycllp_amd import problems
from pycllp_amd.lp import SparseMatrix, StandardLP
from pycllp_amd.solvers import solver_registry
for m, n, B in ((100, 80, 4096), (90, 80, 4096), (64, 64, 8192), (48, 100, 8192)):
    A, b, c = problems.random_dense_arrays(m, n, B, seed=0)
    lp = StandardLP(SparseMatrix(matrix=A), b[:1], c[:1], 0.0).to_equality_form()
    s = solver_registry["hip_dense_primal_normal"](hsd=False); lp.init(s)
    be = torch.as_tensor(b, device="cuda"); ce = torch.as_tensor(np.hstack([c, np.zeros((B, m))]), device="cuda")
    buf = s.solve_device(be, ce); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); buf = s.solve_device(be, ce); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print("dense m=%d n=%d B=%d: %.1f ms  %.0f LPs/s  status0 %d  mean iters %.1f  %s" % (m, n, B, ms, B / ms * 1e3, int((buf["status"] == 0).sum()),
          float(buf["iters"].float().mean()), s.launch_info()))
