"""Timing of the sparse shared-A path on BASELINE config 5's per-GPU share: 16 384 LPs, m=128, n=256, density 0.025."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pycllp_amd import problems
from pycllp_amd.lp import SparseMatrix, StandardLP
from pycllp_amd.solvers import solver_registry
m, n, B = 128, 256, int(os.environ.get("SP_B", 16384))
dens = float(os.environ.get("SP_DENSITY", 0.025))
A, b, c = problems.random_sparse_arrays(m, n, B, density=dens, seed=0)
lp = StandardLP(SparseMatrix(matrix=A), b[:1], c[:1], 0.0).to_equality_form()
s = solver_registry["hip_sparse_primal_normal"](hsd=bool(int(os.environ.get("SP_HSD", "0"))))
lp.init(s)
be = torch.as_tensor(b, device="cuda")
ce = torch.as_tensor(np.hstack([c, np.zeros((B, m))]), device="cuda")
buf = s.solve_device(be, ce); torch.cuda.synchronize()
ts = []
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); buf = s.solve_device(be, ce); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
it = buf["iters"].cpu().numpy(); st = buf["status"].cpu().numpy()
print("sparse (m=%d,n=%d,density=%.3f,nnz=%d) B=%d: %.1f ms -> %.0f LPs/s; status0 %d; iters mean %.1f max %d"
      % (m, n, dens, A.nnz, B, np.median(ts), B / np.median(ts) * 1e3, (st == 0).sum(), it.mean(), it.max()))
gap = (buf["pobj"] - buf["dobj"]).abs() / buf["pobj"].abs().clamp(min=1.0)
print("max rel duality gap %.2e" % gap.max().item())
