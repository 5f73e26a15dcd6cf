"""Timing of a mid-size sparse workload (m = 64, n = 128, density 0.05; 32 768 LPs) on the wave kernel's m <= 64 variant."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pycllp_amd import problems, _native
from pycllp_amd.lp import SparseMatrix, StandardLP
from pycllp_amd.solvers import solver_registry
m, n, B = 64, 128, 32768
A, b, c = problems.random_sparse_arrays(m, n, B, density=0.05, seed=0)
lp = StandardLP(SparseMatrix(matrix=A), b[:1], c[:1], 0.0).to_equality_form()
be = torch.as_tensor(b, device="cuda"); ce = torch.as_tensor(np.hstack([c, np.zeros((B, m))]), device="cuda")
for fl, name in ((0, "wave"), (_native.FLAG_BLOCK_KERNEL, "block")):
    s = solver_registry["hip_sparse_primal_normal"](hsd=False, flags=fl); lp.init(s)
    buf = s.solve_device(be, ce); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); buf = s.solve_device(be, ce); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(name, s.launch_info(), "%.2f ms  %.0f LPs/s  status0 %d  mean iters %.2f" % (ms, B / ms * 1e3, int((buf["status"] == 0).sum()), float(buf["iters"].float().mean())))
