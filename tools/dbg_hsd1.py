import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pycllp_amd import problems
from pycllp_amd.lp import SparseMatrix, StandardLP
from pycllp_amd.solvers import solver_registry
from oracle import port
B = 16384
A, b, c = problems.random_sparse_arrays(128, 256, B, density=0.025, seed=0)
lp = StandardLP(SparseMatrix(matrix=A), b[:1], c[:1], 0.0).to_equality_form()
ce = np.hstack([c, np.zeros((B, 128))])
out = {}
for name, fl in (("wreg", 0), ("block", 64)):
    s = solver_registry["hip_sparse_primal_normal"](device="cuda:0", flags=fl, hsd=True); lp.init(s)
    r = s.solve_device(torch.as_tensor(b, device="cuda"), torch.as_tensor(ce, device="cuda")); torch.cuda.synchronize()
    out[name] = {k: r[k].cpu().numpy() for k in ("status", "iters", "pobj", "dobj")}
bad = np.where(out["wreg"]["status"] != 0)[0]
print("non-optimal in wreg:", bad, out["wreg"]["status"][bad], "iters", out["wreg"]["iters"][bad], "block:", out["block"]["status"][bad], out["block"]["iters"][bad])
print("iters differ on", int((out["wreg"]["iters"] != out["block"]["iters"]).sum()), "LPs; max obj diff", np.abs(out["wreg"]["pobj"] - out["block"]["pobj"]).max())
Ae = np.hstack([A.toarray(), np.eye(128)])
for i in bad[:2]:
    ref = port.dense_solve(Ae, b[i:i+1], ce[i:i+1], flags=32)
    print("oracle:", ref["status"], ref["iters"], ref["pobj"], "wreg pobj", out["wreg"]["pobj"][i], "block pobj", out["block"]["pobj"][i])
