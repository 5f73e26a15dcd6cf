import ctypes, sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pycllp_amd import _native, problems
from pycllp_amd.lp import SparseMatrix, StandardLP
from pycllp_amd.solvers import solver_registry
from oracle import port
dev = torch.device("cuda:0")
B = int(sys.argv[1]); mi = int(sys.argv[2]); flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
A, b, c = problems.random_sparse_arrays(128, 256, B, density=0.025, seed=0)
lp = StandardLP(SparseMatrix(matrix=A), b, c, 0.0).to_equality_form()
s = solver_registry["hip_sparse_primal_normal"](device=dev, flags=flags, max_iter=mi)
lp.init(s)
marks = torch.zeros(4096, dtype=torch.int64).pin_memory()
raw = ctypes.CDLL(_native.LIB_PATH)
if hasattr(raw, "pycllp_hip_debug_set_prof"):
    raw.pycllp_hip_debug_set_prof(ctypes.c_void_p(marks.data_ptr()))
print("init done", flush=True)
bd = torch.as_tensor(b, device=dev); cd = torch.as_tensor(np.hstack([c, np.zeros((B, 128))]), device=dev)
r = s.solve_device(bd, cd)
print("launched", flush=True)
ev = torch.cuda.Event(); ev.record()
for k in range(40):
    if ev.query(): break
    time.sleep(0.5)
    if k in (2, 10, 39): print("marks t=%.1f" % (0.5 * (k + 1)), marks.view(-1, 32)[:4, :10].tolist(), flush=True)
print("done" if ev.query() else "STILL RUNNING", marks.view(-1, 32)[:4, :10].tolist(), flush=True)
torch.cuda.synchronize()
print("synced", flush=True)
Ae = np.hstack([A.toarray(), np.eye(128)]); ce = np.hstack([c, np.zeros((B, 128))])
ref = port.dense_solve(Ae, b, ce, nthreads=16, max_iter=mi)
print("status", r["status"].cpu().numpy()[:8], ref["status"][:8])
print("iters", r["iters"].cpu().numpy()[:8], ref["iters"][:8])
print("pobj", r["pobj"].cpu().numpy()[:4], ref["pobj"][:4])
print("x err", np.abs(r["x"].cpu().numpy() - ref["x"]).max(), "y err", np.abs(r["y"].cpu().numpy() - ref["y"]).max())
