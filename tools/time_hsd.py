"""Plain path vs PYCLLP_FLAG_HSD on (a) the BASELINE config-3 batch (all LPs feasible and bounded) and (b) a batch of the same
shape whose LPs are mostly infeasible or unbounded (mixed-sign A, b, c).  Device-resident timing, 5 launches each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pycllp_amd import problems
from pycllp_amd.lp import SparseMatrix, EqualityLP
from pycllp_amd.solvers import solver_registry

def batch(kind, m=32, n=64, B=65536):
    if kind == "config3":
        A, b, c = problems.random_dense_arrays(m, n, B)
    else:
        rs = np.random.RandomState(11)
        A = rs.rand(m, n) * 2 - 0.3
        b = rs.rand(B, m) * 2 - 0.2; c = rs.rand(B, n) * 2 - 0.3
    return problems.equality_arrays(A, b, c)

for kind in ("config3", "mixed-sign"):
    Ae, be, ce = batch(kind)
    lp = EqualityLP(SparseMatrix(matrix=Ae), be, ce, 0.0)
    bd, cd = torch.as_tensor(be, device="cuda"), torch.as_tensor(ce, device="cuda")
    for hsd in (False, True):
        s = solver_registry["hip_dense_primal_normal"](hsd=hsd)
        lp.init(s)
        buf = s.solve_device(bd, cd); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); s.solve_device(bd, cd); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        st = buf["status"].cpu().numpy(); it = buf["iters"].cpu().numpy()
        print("%-10s %-5s %8.2f ms  %6.2f M LPs/s  status counts %s  mean iterations %.1f" % (
            kind, "hsd" if hsd else "plain", np.median(ts), len(st) / np.median(ts) / 1e3,
            dict(zip(*np.unique(st, return_counts=True))), it.mean()))
