"""Device-resident timing of the dense group kernels on BASELINE config 3 (65 536 x (32, 64)): plain, HSD, predictor-corrector
(r = 0.9), plus HSD on the mixed-sign batch.  Median of 7 launches.  For same-box A/B runs of library variants (PYCLLP_HIP_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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

def run(kind, label, **kw):
    Ae, be, ce = batch(kind)
    lp = EqualityLP(SparseMatrix(matrix=Ae), be, ce, 0.0)
    bd, cd = torch.as_tensor(be, device="cuda"), torch.as_tensor(ce, device="cuda")
    s = solver_registry["hip_dense_primal_normal"](**kw)
    lp.init(s)
    buf = s.solve_device(bd, cd); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); s.solve_device(bd, cd); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    st = buf["status"].cpu().numpy(); it = buf["iters"].cpu().numpy()
    print("%-10s %-9s %8.3f ms  %6.2f M LPs/s  optimal %d  mean iterations %.2f  pobj sum %.12e" % (
        kind, label, np.median(ts), len(st) / np.median(ts) / 1e3, int((st == 0).sum()), it.mean(),
        float(buf["pobj"].double().sum())), flush=True)

which = sys.argv[1:] or ["plain", "hsd", "pc", "hsdmixed"]
if "plain" in which: run("config3", "plain", hsd=False)
if "hsd" in which: run("config3", "hsd", hsd=True)
if "pc" in which: run("config3", "predcorr", hsd=False, predcorr=True)
if "hsdmixed" in which: run("mixed", "hsd", hsd=True)
