import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pycllp_amd import problems
from pycllp_amd.lp import SparseMatrix, StandardLP
from pycllp_amd.solvers import solver_registry
from oracle import port
B = 16384; I = 7557
A, b, c = problems.random_sparse_arrays(128, 256, B, density=0.025, seed=0)
lp = StandardLP(SparseMatrix(matrix=A), b[:1], c[:1], 0.0).to_equality_form()
ce = np.hstack([c, np.zeros((B, 128))]); Ae = np.hstack([A.toarray(), np.eye(128)])
bi, ci = b[I:I+1], ce[I:I+1]
sol = {}
for name, fl in (("wreg", 0), ("block", 64)):
    s = solver_registry["hip_sparse_primal_normal"](device="cuda:0", flags=fl, hsd=True); lp.init(s); sol[name] = s
for k in list(range(1, 40, 3)) + list(range(40, 130, 5)):
    row = []
    ref = port.dense_solve(Ae, bi, ci, flags=32, max_iter=k)
    for name in ("wreg", "block"):
        r = sol[name].solve_device(torch.as_tensor(bi, device="cuda"), torch.as_tensor(ci, device="cuda"), max_iter=k); torch.cuda.synchronize()
        x = r["x"].cpu().numpy()[0]; y = r["y"].cpu().numpy()[0]
        row.append("%s st %d it %d pobj %.10f dx %.1e dy %.1e" % (name, int(r["status"][0]), int(r["iters"][0]), float(r["pobj"][0]),
                   np.abs(x - ref["x"][0]).max(), np.abs(y - ref["y"][0]).max()))
    print("k=%3d oracle st %d it %d pobj %.10f dobj %.10f | %s" % (k, ref["status"][0], ref["iters"][0], ref["pobj"][0], ref["dobj"][0], " | ".join(row)), flush=True)
for mr in (5, 10, 20):
    r = sol["wreg"].solve_device(torch.as_tensor(bi, device="cuda"), torch.as_tensor(ci, device="cuda"), max_refine=mr); torch.cuda.synchronize()
    print("max_refine", mr, "status", int(r["status"][0]), "iters", int(r["iters"][0]), "pobj", float(r["pobj"][0]))
