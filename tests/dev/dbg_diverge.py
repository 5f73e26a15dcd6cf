"""Where do kernel and oracle part ways on the two diverging 1x2 LPs of test_infeasible_and_unbounded_status_codes_match_oracle?
Runs both with max_iter = k for growing k and prints the first k at which x, y or z differ by more than 1e-9 relative."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pycllp_amd.lp import SparseMatrix, EqualityLP
from pycllp_amd.solvers import solver_registry
from oracle import port
cases = [("primal infeasible", np.array([[1.0, 1.0]]), np.array([[-1.0]]), np.array([[1.0, 1.0]])),
         ("unbounded", np.array([[1.0, -1.0]]), np.array([[0.0]]), np.array([[1.0, 0.0]]))]
for name, A, b, c in cases:
    lp = EqualityLP(SparseMatrix(matrix=A), b, c, 0.0)
    s = solver_registry["hip_dense_primal_normal"](); lp.init(s)
    full = port.dense_solve(A, b, c)
    st = lp.solve(s)
    print(name, "kernel status", st[0], "iters", s.iters[0], "| oracle status", full["status"][0], "iters", full["iters"][0])
    for k in range(1, 60):
        r = port.dense_solve(A, b, c, max_iter=k)
        g = s.solve_device(b, c, max_iter=k); torch.cuda.synchronize()
        gx, gy, gz = g["x"].cpu().numpy()[0], g["y"].cpu().numpy()[0], g["z"].cpu().numpy()[0]
        rel = lambda a, ref: np.abs(a - ref).max() / max(1e-300, np.abs(ref).max())
        print("  k=%2d kernel st %d it %d | oracle st %d it %d | rel diff x %.1e y %.1e z %.1e | |x| %.2e |y| %.2e |z| %.2e" % (
            k, int(g["status"][0]), int(g["iters"][0]), r["status"][0], r["iters"][0], rel(gx, r["x"][0]), rel(gy, r["y"][0]), rel(gz, r["z"][0]),
            np.abs(r["x"][0]).max(), np.abs(r["y"][0]).max(), np.abs(r["z"][0]).max()), flush=True)
        if r["status"][0] != 5 and int(g["status"][0]) != 5: break
