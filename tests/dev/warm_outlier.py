#!/usr/bin/env python
"""Follow-up of tests/dev/warm_band.py: the one warm-started LP (seed 6) whose iteration count differs from the oracle's by
36 with the carried-rho build.  Prints, per iteration k (solve with max_iter = k from the same start), the TRUE residuals and the
gap of the kernel's point and of the oracle's, to see which quantity stalls.  GPU box only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from oracle import port  # noqa: E402
from pycllp_amd import problems  # noqa: E402
from pycllp_amd.lp import SparseMatrix, StandardLP  # noqa: E402
from pycllp_amd.solvers import solver_registry  # noqa: E402

print("# library:", os.environ.get("PYCLLP_HIP_LIB", "default"))
m, n, seed = 16, 32, 6
rs = np.random.RandomState(3)
A, b, c = problems.random_dense_arrays(m, n, 300, seed=seed)
lp = StandardLP(SparseMatrix(matrix=A), b, c, 0.0).to_equality_form()
s = solver_registry["hip_dense_primal_normal"](warm_start=True)
lp.init(s); lp.solve(s)
x0, y0, z0 = s.x.copy(), s.y.copy(), s.z.copy()
lp.b[:] = lp.b * (1.0 + 0.01 * rs.rand(*lp.b.shape))
lp.c[:, :n] = lp.c[:, :n] * (1.0 + 0.01 * rs.rand(300, n))
lp.solve(s)
Ae = np.asarray(lp.A.todense())
r = port.dense_solve(Ae, lp.b, lp.c, nthreads=8, x0=x0, y0=y0, z0=z0, flags=1)
diff = np.abs(s.iters.astype(int) - r["iters"])
w = int(np.argmax(diff))
print("LP %d: kernel %d iterations, oracle %d; status %d / %d" % (w, s.iters[w], r["iters"][w], s.status[w], r["status"][w]))
bd, cd = torch.as_tensor(lp.b, device="cuda"), torch.as_tensor(lp.c, device="cuda")
tolr = 1e-10 * (1 + np.linalg.norm(lp.b[w])); tols = 1e-10 * (1 + np.linalg.norm(lp.c[w]))
print("tol_r %.2e tol_s %.2e" % (tolr, tols))
def meas(x, y, z):
    rho = np.linalg.norm(lp.b[w] - Ae @ x); sig = np.linalg.norm(lp.c[w] - Ae.T @ y + z); gam = x @ z
    return rho, sig, gam, lp.c[w] @ x
for k in list(range(1, 16)) + [20, 30, 40, 50]:
    st = s.buffers["set0"]
    st["x"].copy_(torch.as_tensor(x0, device="cuda")); st["z"].copy_(torch.as_tensor(z0, device="cuda")); st["y"].copy_(torch.as_tensor(y0, device="cuda"))
    g = s.solve_device(bd, cd, warm_start=True, max_iter=k); torch.cuda.synchronize()
    rk = port.dense_solve(Ae[:, :], lp.b[w:w + 1], lp.c[w:w + 1], nthreads=1, x0=x0[w:w + 1], y0=y0[w:w + 1], z0=z0[w:w + 1], flags=1, max_iter=k)
    a = meas(g["x"][w].cpu().numpy(), g["y"][w].cpu().numpy(), g["z"][w].cpu().numpy())
    o = meas(rk["x"][0], rk["y"][0], rk["z"][0])
    print("k=%2d kernel it %2d st %d |rho| %.2e |sigma| %.2e gap %.2e min x %.1e min z %.1e   oracle it %2d st %d |rho| %.2e |sigma| %.2e gap %.2e"
          % (k, int(g["iters"][w]), int(g["status"][w]), a[0], a[1], a[2], g["x"][w].min().item(), g["z"][w].min().item(),
             rk["iters"][0], rk["status"][0], o[0], o[1], o[2]))
