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
for _ in range(10):
    t = time.perf_counter(); lp.solve(s); ts.append(time.perf_counter() - t)
print("per call [ms]:", " ".join("%.1f" % (1e3 * t) for t in ts))
print("lp.solve() host-to-host: median %.1f ms -> %.2f M LPs/s (65536 LPs, 67 MB in / 119 MB out over PCIe)"
      % (1e3 * np.median(ts), 65536 / np.median(ts) / 1e6))

# where the time goes: each stage of the pipeline alone
def tm(f, n=5):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return 1e3 * np.median(ts)
st = s.buffers["host"]; buf = s.buffers["set0"]
print("stage b,c into page-locked memory : %.1f ms" % tm(lambda: (np.copyto(st["hb"], lp.b), np.copyto(st["hc"], lp.c))))
print("upload b,c                        : %.1f ms" % tm(lambda: (st["db"].copy_(st["b"], non_blocking=True), st["dc"].copy_(st["c"], non_blocking=True))))
print("solve (one launch)                : %.1f ms" % tm(lambda: s.solve_device(st["db"], st["dc"])))
print("download x,y,z,obj,status,iters   : %.1f ms" % tm(lambda: [st[k].copy_(buf[k], non_blocking=True) for k in ("x", "y", "z", "pobj", "dobj", "status", "iters")]))
for nc in (1, 2, 4, 8, 16):
    s.PIPELINE_CHUNKS = nc
    print("pipeline with %2d chunks           : %.1f ms" % (nc, tm(lambda: lp.solve(s))))

# the sparse shared-A workload (configs[4]'s per-GPU share) through the same host-to-host path
from pycllp_amd.lp import StandardLP
B5 = 16384
A5, b5, c5 = problems.random_sparse_arrays(128, 256, B5, density=0.025, seed=0)
lp5 = StandardLP(SparseMatrix(matrix=A5), b5, c5, 0.0).to_equality_form()
s5 = solver_registry["hip_sparse_primal_normal"](hsd=False)
lp5.init(s5); lp5.solve(s5)
ts = []
for _ in range(6):
    t = time.perf_counter(); lp5.solve(s5); ts.append(time.perf_counter() - t)
print("sparse5 per call [ms]:", " ".join("%.1f" % (1e3 * t) for t in ts))
print("sparse5 lp.solve() host-to-host: median %.1f ms -> %.1f k LPs/s (%d LPs of 128 x 256; device-resident: see bench.py --workload sparse5)"
      % (1e3 * np.median(ts), B5 / np.median(ts) / 1e3, B5))
