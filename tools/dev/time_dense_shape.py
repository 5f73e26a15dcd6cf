"""Device-resident timing of the plain dense solve at a given shape and batch: python tools/dev/time_dense_shape.py m n B"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pycllp_amd import problems
from pycllp_amd.lp import SparseMatrix, EqualityLP
from pycllp_amd.solvers import solver_registry
m, n, B = [int(v) for v in sys.argv[1:4]]
A, b, c = problems.random_dense_arrays(m, n, B)
Ae, be, ce = problems.equality_arrays(A, b, c)
lp = EqualityLP(SparseMatrix(matrix=Ae), be, ce, 0.0)
bd, cd = torch.as_tensor(be, device="cuda"), torch.as_tensor(ce, device="cuda")
s = solver_registry["hip_dense_primal_normal"](hsd=False)
lp.init(s)
buf = s.solve_device(bd, cd); torch.cuda.synchronize()
ts = []
for _ in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); s.solve_device(bd, cd); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
st = buf["status"].cpu().numpy(); it = buf["iters"].cpu().numpy()
print("(%d, %d) x %d: %8.3f ms  %6.2f M LPs/s  optimal %d  mean iterations %.2f  pobj sum %.12e  %s" % (
    m, n, B, np.median(ts), B / np.median(ts) / 1e3, int((st == 0).sum()), it.mean(), float(buf["pobj"].double().sum()), s.launch_info()), flush=True)
